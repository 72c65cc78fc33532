#!/usr/bin/env python3
"""bench.py — edges/s of the panGNN link-prediction train step (GCNConv propagate fwd+bwd, edge
decoder, BCE, Adam) on 1..8 MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json: the config the metric is quoted on): `--simulate_dataset 50000 20 0.2 100 20`
(N = 1e6 genes, ~7.4e7 directed similarity edges, 3e6 neighbour edges), node_dim 64, hidden_dim 128,
fp32, the whole graph as one batch.  A step = zero_grad -> forward -> BCEWithLogits(pos_weight) ->
backward -> Adam; inputs are resident in HBM before the timed region; no host sync inside it.
At N > 1 the same graph is destination-partitioned over the ranks (strong scaling: node ranges with equal expected
in-edge counts, rank-local generation) and the boundary-node rows travel by RCCL all-to-all-v — per step only the
decoder's P halo and its gradient (on a side stream, under the decoder passes), two 1-row neighbour-graph halos and
one flat 216 KB gradient all-reduce (pangnn_amd/dist.py; `exchange="allgather"` is the dense-halo alternative).

One JSON line on rank 0: value = supervised similarity edges per second over the whole job.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (genes/genome, genomes, frac_pos, fragments, shuffled, node_dim, hidden_dim)
    "cfg4": (50000, 20, 0.2, 100, 20, 64, 128),
    "cfg2": (1000, 5, 0.3, 10, 2, 64, 64),
    # the reference's own regime on config 2: DataLoader(batch_size=32) over per-group sub-graphs
    # (pangnn.py:152-153); a step = one mini-batch; informational, not the headline line
    "cfg2mb": (1000, 5, 0.3, 10, 2, 64, 64),
    # the same with a FRESH Batch per step, as the reference's DataLoader hands them out (pangnn.py:152-155,180): the
    # collation and the per-batch structure build (both CSR orders, degree normalisation) are inside the timed step
    "cfg2mb_fresh": (1000, 5, 0.3, 10, 2, 64, 64),
    # BASELINE.json config 5: --simulate_dataset 200000 50 0.1 500 50 --skip_connections --categorical_node, bf16 mixed
    # precision, 8 GPUs (N = 1e7 nodes, 4.3e9 similarity edges: only runs partitioned, --gpus 8)
    "cfg5": (200000, 50, 0.1, 500, 50, 64, 128),
    # one GPU's share of config 5 on ONE GPU: 6 of its 50 genomes with config 5's negative-edge law (m = 220 per gene),
    # same flags and precision — 1.2e6 nodes, ~4.4e8 similarity edges, i.e. what each of the 8 ranks holds
    "cfg5slice": (200000, 6, 0.1, 500, 50, 64, 128),
}
CFG5 = {"cfg5", "cfg5slice"}
HBM_PEAK = 8.0e12            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# what limits the S kernel (profiles/*_pmc_sq_decoder16.txt; updated with the kernel)
S_BINDS = ("vector-instruction issue PLUS matrix-pipe time (557 vector incl. 84 MFMA + 89 LDS + 58 scalar instructions per 16 "
           "edges at two waves per SIMD; measured cycles = 4 x non-matrix VALU + matrix-pipe cycles within 4 %: the two add, "
           "neither hides the other — profiles/r04y_pmc_sq_decoder16.txt, r04q_s_kernel_schedule_and_p3_experiments.txt, "
           "r04_decoder_instruction_budget.txt), not the matrix pipe alone (43 % busy) nor HBM (fabric traffic = 0.48 x the "
           "algorithmic bytes)")


def spmm_alg_bytes(e, n, f, s=4):
    """B_spmm(E,N,F) = E*(4 + 4 + F*s) + N*F*s + (N+1)*8"""
    return e * (8 + f * s) + n * f * s + (n + 1) * 8


def step_alg_bytes(n, e, e_nb, d, h, parts_s, parts_t, s_rows=4):
    """Algorithmic HBM bytes of ONE default-topology train step as built (one term per kernel launch,
    every gathered row counted once per edge, no cache credit), fp32 rows.  `parts_s` / `parts_t`: run-part rows the
    S / T decoder kernels write (one per (32-edge tile, key) run).  The SURVEY.md §8d B_step (159 GB) describes the
    layer-by-layer step of the reference; this step has no [E, 2D] tensors, no transposed conv_in propagate and
    propagates on the 64-wide side."""
    lin = lambda k, m, gate=0: n * 4 * (k + m + gate)           # noqa: E731   node-level dense layer: read K, write M
    t = {
        # conv_in(embedding(x)) by linearity (functional._EmbedConvIn): r a^T + s c^T + b_in written once; the similarity-
        # graph propagate (the * kernel) runs once per GRAPH for r = A_hat x, s = A_hat 1, not per step
        # ... and since the rows are generated inside conv_out's dense kernels (functional._EmbedConvInLinear) they are never
        # written or read: every kernel of the first two layers reads (r_i, s_i) = 8 bytes per row instead of 4 h
        "conv_in_out_linear": n * (8 + 4 * d),                               # y = ELU(h) W_out^T, h generated
        "conv_out_propagate": spmm_alg_bytes(e_nb, n, d, s_rows),
        "decoder_pq_linear": lin(d, 2 * d),
        "decoder_S": e * 556 + parts_s * 4 * d,                              # ids 16 + P 256 + Q 256 + y 4 + logit 4 + record 20
        #                                         (the 32-byte record slot also holds 12 bytes of replicated dL/dlogit: not counted)
        "decoder_S_part_sum": parts_s * 4 * d + n * 4 * d + (n + 1) * 8,
        "decoder_T": e * 28 + parts_t * 4 * d,                               # perm 4 + key 4 + record 20
        "decoder_T_part_sum": parts_t * 4 * d + n * 4 * d + (n + 1) * 8,
        "decoder_pq_dgrad": lin(2 * d, d, d),                                # + gate (pre-activation)
        "decoder_pq_wgrad": lin(2 * d, d),
        "conv_out_propagate_T": spmm_alg_bytes(e_nb, n, d, s_rows),
        "conv_out_bias_sum": n * 4 * d,
        "conv_out_wgrad": n * (4 * d + 8),                                   # g^T ELU(h), h regenerated
        "conv_in_dgrad_sums": n * (4 * d + 8),                               # [r s 1]^T ((g W_out) * ELU'(h)): dL/dh never written
    }
    return t


# kernel-name substrings of the three kernels whose counter traffic profiles/traffic.json holds (tools/update_traffic.py)
# (the S kernel by its exact instance — fused loss, run sums, no skip feature, f32 tables: the headline step's — so that launches of
# the literal route's / another workload's instance in the same counter pass are not averaged in)
TRAFFIC_KERNELS = {"decoder_train": "decoder_train16_kernel<true, true, false, false>", "decoder_dgrad": "decoder_dgrad16_kernel<true, true>",
                   "spmm_fwd": "spmm_row_kernel<64, 4, false, 0>"}


# the sources the three counted kernels are compiled from: profiles/traffic.json carries their hash at collection time
TRAFFIC_SOURCES = ("decoder16.hip", "spmm.hip", "common.h")


def kernel_source_sha16():
    import hashlib
    h = hashlib.sha256()
    for name in TRAFFIC_SOURCES:
        with open(os.path.join(ROOT, "pangnn_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def traffic_entry(prof, key, workload, world, e_sim):
    """the profiles/traffic.json entry of kernel `key` IF it was collected on this workload, GPU count, edge count and
    kernel name AND from the kernel sources this tree holds (tools/update_traffic.py stamps their hash) — otherwise None
    (the line then carries "traffic": null instead of a stale figure)"""
    sha = kernel_source_sha16()
    for ent in prof.get("entries", []):
        if ent.get("key") == key and ent.get("workload") == workload and ent.get("n_gpus") == world \
                and ent.get("sim_edges") == e_sim and TRAFFIC_KERNELS[key] in ent.get("kernel", "") \
                and ent.get("kernel_source_sha16") == sha:
            return ent
    return None


def cpu_baseline(args, d, h):
    """The oracle train step (index_select -> mul -> index_add_ GCNConv, literal cat+MLP decoder) on the
    host cores of this box, on a bounded sample of the same workload."""
    from oracle import gcn_oracle as go
    from pangnn_amd import simulate
    genes = args.cpu_genes
    _, G, frac, frags, shuf, _, _ = WORKLOADS[args.workload]
    g = simulate.simulate_graph(genes, G, frac, frags, shuf, seed=1, device="cpu")
    torch.manual_seed(0)
    m = go.AlternateGCNOracle(dims=(d, h))
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    go.train_step(m, opt, g, g.y, g.class_balance)
    t0 = time.perf_counter()
    steps = 0
    while steps < args.cpu_steps:
        go.train_step(m, opt, g, g.y, g.class_balance)
        steps += 1
        if steps >= 3 and time.perf_counter() - t0 > 45.0:
            break
    dt = time.perf_counter() - t0
    e = g.edge_index.shape[1]
    return {"value": e * steps / dt, "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"--simulate_dataset {genes} {G} {frac} {frags} {shuf} (E_sim={e}), {steps} whole-graph "
                      f"train steps of oracle/gcn_oracle.py, {dt / steps * 1e3:.0f} ms/step"}


def minibatch_line(workload, dev, steps, warmup_min, seed=0, genes=None):
    """the reference's own regime (pangnn.py:152-216): DataLoader(batch_size=32) over per-group sub-graphs, one train step
    per mini-batch.  cfg2mb: one captured HIP graph per batch, replayed; cfg2mb_fresh: a fresh Batch every step (collation,
    both CSR orders, degree norms, run-sum plans, the first layer's r / s inside the timed step)."""
    import pangnn_amd
    from pangnn_amd import simulate
    from pangnn_amd.train import make_optimizer, train_step
    g0, G, frac, frags, shuf, d, h = WORKLOADS[workload]
    genes = genes or g0
    ds = simulate.simulate_subgraph_dataset(genes, G, frac, frags, shuf, seed=seed, device=dev)
    n_train = int(len(ds) * 0.7)                                       # split_data((0.7, 0.15, 0.01))
    batches = [ds.batch(i, min(i + 32, n_train)) for i in range(0, n_train, 32)]
    pw = ds.class_balance()
    torch.manual_seed(0)
    model = pangnn_amd.AlternateGCN(dev, None, False, dims=[d, h])
    fresh = workload == "cfg2mb_fresh"
    graphed = os.environ.get("PANGNN_HIPGRAPH", "1") != "0" and not fresh
    # a fresh batch per step as ONE captured HIP graph over fixed-shape padded buffers (train.ReplayedFreshStep);
    # PANGNN_FRESH_REPLAY=0: round 3's eagerly launched fresh step
    replayed = fresh and os.environ.get("PANGNN_FRESH_REPLAY", "1") != "0"
    opt = make_optimizer(model, capturable=graphed or replayed)
    edge_counts = [int(b.edge_index.shape[1]) for b in batches]
    if replayed:
        from pangnn_amd.train import ReplayedFreshStep
        # DataLoader(shuffle=True): a random permutation of the training sub-graphs, 32 consecutive ids per batch
        perm = torch.randperm(n_train, generator=torch.Generator().manual_seed(seed)).tolist()
        id_lists = [perm[i:i + 32] for i in range(0, n_train, 32)]
        rstep = ReplayedFreshStep(model, opt, ds, pw, 32, graphs=range(n_train))
        eo = ds._host().edge
        edge_counts = [sum(eo[i + 1] - eo[i] for i in ids) for ids in id_lists]
        steps_fn = [(lambda ids=ids: rstep(ids)) for ids in id_lists]
    elif graphed:
        from pangnn_amd.train import GraphedTrainStep
        steps_fn = [GraphedTrainStep(model, opt, b, b.y, pw) for b in batches]     # one HIP graph per batch
    elif fresh:
        from pangnn_amd.graph import clear_cache
        spans = [(i, min(i + 32, n_train)) for i in range(0, n_train, 32)]

        def fresh_step(k):
            clear_cache()                                   # nothing survives from an earlier batch
            b = ds.batch(*spans[k])                         # collation: new tensors, new structure, new norms
            return train_step(model, opt, b, b.y, pw)
        steps_fn = [(lambda k=k: fresh_step(k)) for k in range(len(spans))]
    else:
        steps_fn = [(lambda b=b: train_step(model, opt, b, b.y, pw)) for b in batches]
    # every batch has its own tensor sizes: the untimed steps cover one pass over the batches, so that the timed ones see the
    # steady state of the caching allocator (and of the kernels' code objects), as every epoch after the first does
    warmup = max(warmup_min, len(batches))
    for k in range(warmup):
        steps_fn[k % len(batches)]()
    torch.cuda.synchronize()
    # a host-bound step: keep the interpreter's cyclic collector from re-walking the (large, static) heap the data set
    # construction left behind during the timed steps — collect once now and move the survivors out of its generations
    import gc
    gc.collect()
    gc.freeze()
    t0 = time.perf_counter()
    edges = 0
    for k in range(steps):
        loss, _ = steps_fn[k % len(batches)]()
        edges += edge_counts[k % len(batches)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.unfreeze()
    return {"metric": "edges/sec in GNN forward+backward (link-pred train step)", "value": edges / dt,
            "unit": "edges/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"--simulate_dataset {genes} {G} {frac} {frags} {shuf} --train, mini-batches of 32 "
                                   f"per-group sub-graphs ({len(batches)} batches, {n_train} train sub-graphs), "
                                   f"node_dim={d} hidden_dim={h}" + (", one captured HIP graph per batch" if graphed else "") +
                                   (", a fresh Batch per step: collation + structure build (2 CSR orders, degree norms) inside "
                                    "the timed step" if fresh else "") +
                                   (", shuffled sub-graph order, ONE captured HIP graph over fixed-shape padded buffers "
                                    f"(max graphs / nodes / edges / neighbour edges = {rstep.spec}) serving every batch"
                                    if replayed else ""),
                       "mean_edges_per_batch": edges / steps, "final_loss": float(loss.item())}}


def minibatch_bench(args, dev, json_fd):
    line = minibatch_line(args.workload, dev, args.steps, args.warmup, seed=args.seed, genes=args.genes)
    os.write(json_fd, (json.dumps(line) + "\n").encode())


def _mb_sibling(workload, dev, seed):
    ln = minibatch_line(workload, dev, 200, 3, seed=seed)
    return {"what": ln["config"]["workload"], "steps": ln["steps"], "ms_per_step": ln["ms_per_step"], "value": ln["value"],
            "unit": "edges/s", "mean_edges_per_batch": ln["config"]["mean_edges_per_batch"],
            "final_loss": ln["config"]["final_loss"]}


def cfg5slice_sibling(dev, steps, seed=0):
    """one GPU's share of BASELINE config 5 (6 of its 50 genomes with config 5's negative-edge law, --skip_connections
    --categorical_node, bf16 autocast) timed in this process after the headline: the categorical embedding keeps the *
    propagate and its transpose inside every step"""
    import pangnn_amd
    from pangnn_amd import functional as PF
    from pangnn_amd import simulate
    from pangnn_amd.train import make_optimizer, train_step
    genes, G, frac, frags, shuf, d, h = WORKLOADS["cfg5slice"]
    t0 = time.perf_counter()
    g = simulate.simulate_graph(genes, G, frac, frags, shuf, seed=seed, device=dev, mean_neg=220, adjacent_only=True)
    n, e_sim = g.num_nodes, g.edge_index.shape[1]
    g.x = torch.arange(n, device=dev)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    torch.manual_seed(0)
    model = pangnn_amd.AlternateGCN(dev, None, True, dims=[d, h], num_nodes=n, skip_connections=True)
    opt = make_optimizer(model)
    amp = torch.autocast("cuda", dtype=torch.bfloat16)

    def step():
        with amp:
            return train_step(model, opt, g, g.y, g.class_balance)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    old, PF.KERNEL_TIMER = PF.KERNEL_TIMER, {"sim.fwd": [], "sim.bwd": [], "dec.bwd": [], "dec.dgrad": []}
    try:
        t1 = time.perf_counter()
        for _ in range(steps):
            loss, _ = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        tm = PF.KERNEL_TIMER
    finally:
        PF.KERNEL_TIMER = old
    avg = lambda tag: (sum(a.elapsed_time(b) for a, b in tm[tag]) / steps) if tm.get(tag) else None     # noqa: E731  ms per step
    return {"what": "--simulate_dataset 200000 6 0.1 500 50 with m = 220 negatives per gene (config 5's law), "
                    "--skip_connections --categorical_node, bf16 autocast, whole slice as one batch on one GPU",
            "nodes": n, "sim_edges": e_sim, "steps": steps, "ms_per_step": dt / steps * 1e3, "value": e_sim * steps / dt,
            "unit": "edges/s", "dtype": "bf16 rows / f32", "graph_build_s": round(t_gen, 3),
            "propagate_fwd_ms_per_step": avg("sim.fwd"), "propagate_bwd_ms_per_step": avg("sim.bwd"),
            "decoder_S_ms_per_step": avg("dec.bwd"), "decoder_T_ms_per_step": avg("dec.dgrad"),
            "final_loss": float(loss.item())}


def reference_loop_sibling(dev, graph, labels, class_balance, d, h, e_sim, steps, deferred=True):
    """The reference's own loop, call for call, under a real accelerate.Accelerator on the headline graph
    (/root/reference/pangnn.py:25 Accelerator, :87-98 model / torch Adam / torch BCEWithLogitsLoss(pos_weight = host scalar),
    :122 prepare, :190-216 model.train(); zero_grad(); output = model(batch); loss = criterion(output, labels);
    accelerator.backward(loss); optimizer.step()) — what a maintainer runs after INTEGRATION.md §1's import swap.
    `deferred=True` (the default of the package): model(batch) returns a DeferredLogits handle that torch's criterion resolves
    through the one-pass training decoder; False: round 4's literal route (inference decoder in forward, loss kernel, S with
    the given dL/dlogits + T in backward).  Also timed: the same steps followed by the loop's per-batch reporting lines
    (:218-222 loss.item() — a host sync —, sigmoid(output.detach()), threshold, confusion-matrix update)."""
    import pangnn_amd
    from accelerate import Accelerator
    from accelerate.state import AcceleratorState
    from pangnn_amd import functional as PF
    from pangnn_amd.metrics import BinaryConfusionMatrix
    AcceleratorState._reset_state(True)
    accelerator = Accelerator(mixed_precision="no")                                               # pangnn.py:25
    torch.manual_seed(0)
    model = pangnn_amd.AlternateGCN(device=accelerator.device, dataset=None, categorical_nodes=False, dims=[d, h],
                                    deferred_logits=deferred)                                     # pangnn.py:87
    optimizer = torch.optim.Adam(model.parameters(), lr=0.001)                                    # pangnn.py:88
    criterion = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor(float(class_balance)))         # pangnn.py:98
    model, optimizer = accelerator.prepare(model, optimizer)                                      # pangnn.py:122
    conf = BinaryConfusionMatrix(0.5, device=dev)
    batch = graph

    def step(report):
        model.train()                                                                             # pangnn.py:190
        optimizer.zero_grad()                                                                     # pangnn.py:194
        output = model(batch)                                                                     # pangnn.py:200
        loss = criterion(output, labels)                                                          # pangnn.py:203
        accelerator.backward(loss)                                                                # pangnn.py:207
        optimizer.step()                                                                          # pangnn.py:216
        if report:
            loss.item()                                                                           # pangnn.py:218
            probabilities = torch.sigmoid(output.detach())                                        # pangnn.py:220
            conf.update((probabilities >= 0.5).int(), labels)                                     # pangnn.py:221-222
        return loss, output

    for _ in range(3):
        step(True)
    torch.cuda.synchronize()
    old, PF.KERNEL_TIMER = PF.KERNEL_TIMER, {"dec.fwd": [], "dec.bwd": [], "dec.dgrad": []}
    try:
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, output = step(False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tm = PF.KERNEL_TIMER
    finally:
        PF.KERNEL_TIMER = old
    route = getattr(output, "route", "plain tensor")
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    torch.cuda.synchronize()
    dt_rep = time.perf_counter() - t0
    per_step = lambda tag: (sum(a.elapsed_time(b) for a, b in tm[tag]) / steps) if tm.get(tag) else 0.0    # noqa: E731
    AcceleratorState._reset_state(True)
    ms = dt / steps * 1e3
    return {"what": "pangnn.py:190-216 literally under accelerate.Accelerator(mixed_precision='no'): model.train(); "
                    "optimizer.zero_grad(); output = model(batch); loss = torch.nn.BCEWithLogitsLoss(pos_weight)(output, labels); "
                    "accelerator.backward(loss); optimizer.step() with torch.optim.Adam, same graph as the headline" +
                    ("" if deferred else "; deferred_logits=False: forward launches the inference decoder, backward S + T"),
            "steps": steps, "ms_per_step": ms, "value": e_sim * steps / dt, "unit": "edges/s",
            "output_type": type(output).__name__, "route": route,
            "decoder_fwd_ms_per_step": per_step("dec.fwd"), "decoder_S_ms_per_step": per_step("dec.bwd"),
            "decoder_T_ms_per_step": per_step("dec.dgrad"),
            "decoder_share_of_step": (per_step("dec.fwd") + per_step("dec.bwd") + per_step("dec.dgrad")) / ms,
            "with_reporting_ms_per_step": dt_rep / steps * 1e3,
            "with_reporting_what": "the same steps each followed by pangnn.py:218-222: loss.item() (host sync), "
                                   "torch.sigmoid(output.detach()), (p >= 0.5).int(), confusion-matrix update on the device",
            "final_loss": float(loss.item())}


def eval_sibling(dev, model, graph, labels, pos_weight, e_sim, passes=3):
    """the validation pass of pangnn.py:241-289 / predict.py:34 on the headline graph: no_grad forward (inference decoder
    kernel), loss, confusion counts, ROC-AUC and PR-AUC accumulated on the device (pangnn_amd.train.evaluate)"""
    from pangnn_amd.train import evaluate
    evaluate(model, [graph], pos_weight)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        res = evaluate(model, [graph], pos_weight)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    with torch.no_grad():
        model.eval()
        model(graph)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(passes):
            model(graph)
        torch.cuda.synchronize()
        dt_fwd = time.perf_counter() - t1
        model.train()
    return {"what": "validation pass over the whole graph (pangnn.py:241-289): no_grad forward + loss + confusion counts + ROC-AUC + "
                    "PR-AUC, all on the device, one read-out at the end", "passes": passes, "ms_per_pass": dt / passes * 1e3,
            "value": e_sim * passes / dt, "unit": "edges/s", "forward_only_ms": dt_fwd / passes * 1e3,
            "roc_auc": res["roc_auc"], "pr_auc": res["pr_auc"], "loss": res["loss"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg4", choices=sorted(WORKLOADS))
    ap.add_argument("--genes", type=int, default=None, help="override genes per genome (debug)")
    ap.add_argument("--cpu-genes", type=int, default=2500, help="genes per genome of the CPU-baseline sample (1/20 of cfg 4)")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-siblings", action="store_true",
                    help="keep the general-feature and strict-fp32 timings but skip the siblings that build their own data "
                         "(cfg5slice, mini-batch regimes) and the reference_loop / eval ones: what the counter passes of "
                         "tools/pmc_traffic.sh want")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the sibling timings (general-feature step, strict fp32, cfg5slice, mini-batch regimes)")
    ap.add_argument("--reference-loop", action="store_true", help="time the reference_loop / eval siblings with --genes too")
    ap.add_argument("--emulate-rank", type=int, default=None,
                    help="ONE process stands in for rank r of an --of W rank job (compute side of the partitioned step on one GPU; "
                         "the halo exchange is a self-exchange of the true row counts over RCCL: dist.HaloPlan._emulate)")
    ap.add_argument("--of", type=int, default=8, help="world size the emulated rank belongs to")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    # stdout carries exactly ONE JSON line: everything native libraries print there (RCCL's version
    # banner at communicator creation) is sent to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"WORLD_SIZE={world} but --gpus {args.gpus}: launch with python -m torch.distributed.run "
                 f"--nproc-per-node N ... bench.py --gpus N")
    # rehearsal hooks (tests/test_bench_ranks.py): several ranks on ONE GPU over gloo — RCCL refuses two ranks on a
    # device — to run this file's N > 1 branch end to end; the product run is one rank per GPU over RCCL
    backend = os.environ.get("PANGNN_BENCH_BACKEND", "nccl")
    if os.environ.get("PANGNN_BENCH_ONE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def all_reduce_(t, op):
        """small control-plane reductions; staged through the host on gloo"""
        if backend == "nccl":
            torch.distributed.all_reduce(t, op=op)
        else:
            c = t.cpu()
            torch.distributed.all_reduce(c, op=op)
            t.copy_(c)
        return t

    import pangnn_amd
    from pangnn_amd import functional as PF
    from pangnn_amd import simulate
    from pangnn_amd.train import make_optimizer, train_step

    genes, G, frac, frags, shuf, d, h = WORKLOADS[args.workload]
    if args.genes:
        genes = args.genes
    if args.workload in ("cfg2mb", "cfg2mb_fresh"):
        return minibatch_bench(args, dev, json_fd)
    emu = args.emulate_rank is not None
    if emu and (world != 1 or not 0 <= args.emulate_rank < args.of):
        sys.exit("--emulate-rank r --of W runs as ONE process with 0 <= r < W")
    force_dist = os.environ.get("PANGNN_FORCE_DIST") == "1" or emu   # exercise the partitioned path at world = 1
    if force_dist and world == 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    partitioned = world > 1 or force_dist
    cfg5 = args.workload in CFG5
    mean_neg = 220 if args.workload == "cfg5slice" else None       # floor(49 / 2 * 9): config 5's law on a 6-genome slice
    model_flags = dict(skip_connections=True) if cfg5 else {}
    amp = torch.autocast("cuda", dtype=torch.bfloat16) if cfg5 else contextlib.nullcontext()
    if args.workload == "cfg5" and world < 8 and not args.genes and not (emu and args.of >= 8):
        sys.exit("cfg5 (4.3e9 edges) only runs partitioned over 8 GPUs; on one GPU use --workload cfg5slice")
    replicated = os.environ.get("PANGNN_BENCH_REPLICATED") == "1"    # round-1 way: every rank builds the whole graph
    # node ranges with equal expected in-edge counts (the two end genomes have one neighbour genome, the others two):
    # PANGNN_PARTITION=nodes keeps equal node ranges
    bounds = None
    gen_rank, gen_world = (args.emulate_rank, args.of) if emu else (rank, world)
    if partitioned and gen_world > 1 and os.environ.get("PANGNN_PARTITION", "edges") == "edges":
        from pangnn_amd.dist import balanced_bounds
        bounds = balanced_bounds(genes, G, gen_world)
    t_gen = time.perf_counter()
    g = None
    if partitioned and not replicated:
        # rank-local generation: a rank draws only the genome pairs around its node range (simulate.simulate_shard,
        # bit-identical to partitioning the whole graph: tests/test_construct.py) — what lets config 5 exist at all
        part = simulate.simulate_shard(genes, G, frac, frags, shuf, seed=args.seed, device=dev, rank=gen_rank, world=gen_world,
                                       mean_neg=mean_neg, bounds=bounds)
        if emu:
            part.emulated_world = gen_world      # the halo plan of the real rank, served by a self-exchange (HaloPlan._emulate)
        cnt = torch.tensor([part.e_sim_local, part.n_pos_local, part.neighbour_edge_index.shape[1]], dtype=torch.int64,
                           device=dev)
        if world > 1:
            all_reduce_(cnt, torch.distributed.ReduceOp.SUM)
        e_sim, n_pos, e_nb = (int(v) for v in cnt.tolist())
        part.e_sim_total = e_sim * (gen_world if emu else 1)     # an emulated rank only knows its own count: the loss
        #                                                          denominator (a scale) is taken as W times it
        n = part.n_global
        class_balance = torch.tensor((e_sim - n_pos) / max(n_pos, 1), dtype=torch.float32, device=dev)   # dataset.py:346
    else:
        g = simulate.simulate_graph(genes, G, frac, frags, shuf, seed=args.seed, device=dev, mean_neg=mean_neg,
                                    adjacent_only=cfg5)
        n, e_sim, e_nb = g.num_nodes, g.edge_index.shape[1], g.neighbour_edge_index.shape[1]
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen

    torch.manual_seed(0)
    if partitioned:
        from pangnn_amd import dist as pdist
        if g is not None:
            # every rank generated the graph itself from the same seed: verify they agree before slicing it
            chk = torch.stack([g.edge_index.sum(), g.edge_index[0].max(), torch.tensor(e_sim, device=dev),
                               (g.edge_attr.double().sum() * 1e3).long()]).long()
            lo_, hi_ = chk.clone(), chk.clone()
            if world > 1:
                all_reduce_(lo_, torch.distributed.ReduceOp.MIN)
                all_reduce_(hi_, torch.distributed.ReduceOp.MAX)
            if not torch.equal(lo_, hi_):
                raise RuntimeError("ranks generated different graphs from the same seed")
            part = pdist.partition_graph(g, rank, world, bounds)
            class_balance = g.class_balance
            del g
        model = pdist.DistAlternateGCN(dev, dims=[d, h], part=part, categorical_nodes=cfg5, **model_flags)
        graph, labels = part, part.y
        pos_weight = class_balance

        def step_fn():
            with amp:
                return pdist.train_step(model, opt, graph, labels, pos_weight)
    else:
        model = pangnn_amd.AlternateGCN(dev, None, cfg5, dims=[d, h], num_nodes=n,
                                        fold_activation=os.environ.get("PANGNN_FOLD_ACT", "1") == "1",   # A/B switch
                                        **model_flags)
        if cfg5:
            g.x = torch.arange(n, device=dev)         # --categorical_node: x = node ids (build-defined)
        graph, labels, pos_weight = g, g.y, g.class_balance

        def step_fn():
            with amp:
                return train_step(model, opt, graph, labels, pos_weight)
    opt = make_optimizer(model)

    t_struct = time.perf_counter()
    for _ in range(args.warmup):          # first warm-up step also builds CSR/CSC + norms (cached)
        step_fn()
    torch.cuda.synchronize()
    t_struct = time.perf_counter() - t_struct

    PF.KERNEL_TIMER = {"sim.fwd": [], "sim.bwd": [], "dec.bwd": [], "dec.dgrad": []}
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step_fn()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    timer, PF.KERNEL_TIMER = PF.KERNEL_TIMER, None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        all_reduce_(t, torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        # a rank's loss is its share of the global mean (local sum / global edge count): report the whole job's
        loss = all_reduce_(loss.detach().clone().reshape(1), torch.distributed.ReduceOp.SUM)[0]

    def _avg(tag, tm=None):
        ev = (tm or timer).get(tag, [])
        return (sum(a.elapsed_time(b) for a, b in ev) * 1e-3 / len(ev)) if ev else None

    if emu:
        # one line per emulated rank: the compute side of the partitioned step on ONE GPU — NOT a scaling curve
        plans = getattr(graph, "_dist_plans", {})
        s_ev, t_ev = timer.get("dec.bwd", []), timer.get("dec.dgrad", [])
        per = max(len(s_ev) // max(args.steps, 1), 1)
        s_ms = [sum(a.elapsed_time(b) for a, b in s_ev[k::per]) / max(len(s_ev[k::per]), 1) for k in range(per)]
        rows_bytes = 2 if cfg5 else 4
        sim = plans.get("sim")
        line = {"emulated_rank": gen_rank, "of": gen_world, "workload": args.workload, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": dt / args.steps * 1e3, "nodes_local": int(graph.n_local), "sim_edges_local": int(graph.e_sim_local),
                "neighbour_edges_local": int(graph.neighbour_edge_index.shape[1]),
                "node_range": [int(graph.lo), int(graph.hi)], "partition": "equal expected in-edge counts" if bounds else "equal node ranges",
                "halo_rows": {k: int(p.n_halo) for k, p in plans.items()},
                "halo_rows_by_owner": {k: getattr(p, "peer_counts", None) for k, p in plans.items()},
                "own_source_edges": int(sim.e_hi - sim.e_lo) if sim is not None else None,
                "halo_source_edges": int(sim.edge_index.shape[1] - (sim.e_hi - sim.e_lo)) if sim is not None else None,
                "per_step_exchange_bytes": {
                    "decoder_P_halo_forward": int(sim.n_halo) * 64 * rows_bytes if sim is not None else None,
                    "decoder_P_halo_gradient_back": int(sim.n_halo) * 64 * rows_bytes if sim is not None else None,
                    "neighbour_graph_halos_fwd_plus_bwd": 2 * int(plans["nb"].n_halo) * 64 * 4 if "nb" in plans else None,
                    "gradient_all_reduce": sum(p.numel() for k_, p in model.named_parameters() if p.requires_grad and not
                                               (getattr(model, "sharded_embedding", False) and k_ == "embedding.weight")) * 4},
                "decoder_S_launch_ms": s_ms, "decoder_S_launch_note": "own-source edges first (the P halo travels under it), then the halo-source edges",
                "decoder_T_ms_per_step": (sum(a.elapsed_time(b) for a, b in t_ev) / max(args.steps, 1)) if t_ev else None,
                "xgmi_time_at_link_rate_ms": (int(sim.n_halo) * 64 * rows_bytes / 2 / 153e9 * 1e3) if sim is not None else None,
                "xgmi_note": "the P halo comes from two neighbour ranks over two xGMI links (153 GB/s each, MI355X_MICROARCH.md): bytes / 2 / 153 GB/s",
                "exchange_here": "self-exchange of the same row counts over RCCL on one GPU (values meaningless, launches / bytes / streams real)",
                "graph_build_s": round(t_gen, 3), "final_loss": float(loss.item()), "dtype": "bf16 rows / f32" if cfg5 else "f32"}
        os.write(json_fd, (json.dumps(line) + "\n").encode())
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        return

    t_prop, t_dec, t_dgr = _avg("sim.fwd"), _avg("dec.bwd"), _avg("dec.dgrad")
    n_dgr = len(timer.get("dec.dgrad", [])) // max(args.steps, 1)
    # launches of the S kernel per step: 1 on a whole graph, 2 on a shard (own-source edges, then halo-source edges);
    # the roofline figures below are per STEP's worth of S launches (all of the rank's edges), not per launch
    n_dec = max(len(timer.get("dec.bwd", [])) // max(args.steps, 1), 1)
    if t_dec:
        t_dec *= n_dec
    if t_dgr:
        t_dgr *= max(n_dgr, 1)               # likewise for the T kernel (own-source / halo-source ranges on a shard)

    # ---- outside the headline: (i) the conv_in propagate and its transpose — the * kernel of SURVEY.md §8.  The default
    # step evaluates conv_in(embedding(x)) by linearity and runs that propagate once per graph, not per step (config 5's
    # --categorical_node, the union / hidden layers and every mini-batch run it per step): timed here on a layer-by-layer
    # (fuse_embedding=False) model over the same graph;
    # (ii) the strict-fp32 step (every decoder product on f32 MFMA: PANGNN_DECODER_PRECISION=0)
    extra = {}
    if world == 1 and not force_dist and not args.no_extras and not cfg5:
        old_mode = PF.DECODER_PRECISION
        try:
            # (i) general-feature step: the same graph, the same model layer by layer (fuse_embedding=False) — the * propagate
            # over the similarity graph and its transpose run inside every step, as they do for --categorical_node, the union /
            # hidden layers and any non-constant node feature
            PF.KERNEL_TIMER = {"sim.fwd": [], "sim.bwd": []}
            m2 = pangnn_amd.AlternateGCN(dev, None, False, dims=[d, h], fuse_embedding=False)
            o2 = make_optimizer(m2)
            for _ in range(3):
                train_step(m2, o2, graph, labels, pos_weight)
            torch.cuda.synchronize()
            PF.KERNEL_TIMER = {"sim.fwd": [], "sim.bwd": []}
            n_gen = max(args.steps // 2, 5)
            t1 = time.perf_counter()
            for _ in range(n_gen):
                l2, _ = train_step(m2, o2, graph, labels, pos_weight)
            torch.cuda.synchronize()
            dt_gen = time.perf_counter() - t1
            extra["bwd_avg_launch_ms"] = _avg("sim.bwd", PF.KERNEL_TIMER) * 1e3
            extra["prop_fwd_s"] = _avg("sim.fwd", PF.KERNEL_TIMER)
            extra["general_features"] = {
                "what": "the same cfg-4 step evaluated layer by layer (fuse_embedding=False): conv_in's similarity-graph propagate "
                        "(the * kernel, spmm_row_kernel) and its transpose inside every step — the step of --categorical_node, of "
                        "the union / hidden layers and of any non-constant node feature",
                "steps": n_gen, "ms_per_step": dt_gen / n_gen * 1e3, "value": e_sim * n_gen / dt_gen, "unit": "edges/s",
                "propagate_fwd_ms": extra["prop_fwd_s"] * 1e3, "propagate_bwd_ms": extra["bwd_avg_launch_ms"],
                "propagate_share_of_step": (extra["prop_fwd_s"] + extra["bwd_avg_launch_ms"] * 1e-3) / (dt_gen / n_gen),
                "final_loss": float(l2.item())}
            del m2, o2
            PF.KERNEL_TIMER = None
            # (ii) strict fp32: every decoder product on the f32 matrix instructions
            PF.DECODER_PRECISION = 0
            for _ in range(2):
                step_fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_strict = max(args.steps // 2, 3)
            for _ in range(n_strict):
                step_fn()
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            extra["strict_fp32"] = {
                "what": "the same step with every decoder product on v_mfma_f32_32x32x2_f32 (bit-exact fp32 FMA chains, "
                        "PANGNN_DECODER_PRECISION=0; [E,64] dL/dh1 round trip of round 1)",
                "steps": n_strict, "ms_per_step": dt1 / n_strict * 1e3, "value": e_sim * n_strict / dt1,
                "unit": "edges/s"}
        except Exception as ex:                                   # never lose the headline over an extra
            extra["extras_error"] = repr(ex)
        finally:                                                  # whatever happened, the process is back in the mode
            PF.DECODER_PRECISION = old_mode                       # the emitted line describes
            PF.KERNEL_TIMER = None
        # (ii-b) the reference's own loop under accelerate on the same graph (deferred logits = the package default, and
        # round 4's literal route beside it), and the validation pass
        if (not args.genes and not args.no_siblings) or args.reference_loop:
            n_ref = max(args.steps, 10)
            for key, fn in (("reference_loop", lambda: reference_loop_sibling(dev, graph, labels, pos_weight, d, h, e_sim, n_ref)),
                            ("reference_loop_literal", lambda: reference_loop_sibling(dev, graph, labels, pos_weight, d, h, e_sim,
                                                                                      n_ref, deferred=False)),
                            ("eval", lambda: eval_sibling(dev, model, graph, labels, pos_weight, e_sim))):
                try:
                    extra[key] = fn()
                except Exception as ex:
                    extra[key] = {"error": repr(ex)}
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
        # (iii) the other regimes of the path, timed in this process after the headline (each builds its own data):
        # config 5's per-GPU slice, and the reference's mini-batch regime replayed / with a fresh Batch per step
        if args.workload == "cfg4" and not args.genes and not args.no_siblings:
            for key, fn in (("cfg5slice", lambda: cfg5slice_sibling(dev, max(args.steps // 2, 5), args.seed)),
                            ("cfg2mb", lambda: _mb_sibling("cfg2mb", dev, args.seed)),
                            ("cfg2mb_fresh", lambda: _mb_sibling("cfg2mb_fresh", dev, args.seed))):
                try:
                    extra[key] = fn()
                except Exception as ex:
                    extra[key] = {"error": repr(ex)}
                torch.cuda.synchronize()
                torch.cuda.empty_cache()

    if rank == 0:
        rows_local = getattr(graph, "n_local", n)
        e_local = getattr(graph, "e_sim_local", e_sim)
        f_spmm = min(d, h)        # GCNConv propagates on the narrower side of its dense layer
        prof = {}
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                prof = json.load(open(tf))
            except Exception:
                prof = {}
        src_note = str(prof.get("_source", "profiles/"))
        step_s = dt / args.steps
        # run-part rows written by the decoder's S (by source) and T (by target) kernels on this graph
        st_sim = getattr(graph, "_pangnn_structs", {}).get("sim", (None, None))[1]
        ct = PF.d16_chunk(st_sim.num_edges) if st_sim is not None else PF.d16_chunk()
        pl_s = st_sim.runsum_plan(ct) if st_sim is not None else None
        pl_d = st_sim.csr_plan("dst", ct) if st_sim is not None else None
        n_parts_s = pl_s.n_parts_exact() if pl_s is not None else 0        # (the buffers are sized by an upper bound)
        n_parts_d = pl_d.n_parts_exact() if pl_d is not None else 0
        line = {
            "metric": "edges/sec in GNN forward+backward (link-pred train step)",
            "value": e_sim * args.steps / dt, "unit": "edges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16 rows / f32" if cfg5 else "f32", "data": "synthetic",
            "config": {"workload": f"--simulate_dataset {genes} {G} {frac} {frags} {shuf}, whole graph as one batch, "
                                   f"node_dim={d} hidden_dim={h}, AlternateGCN default topology, mlp decoder" +
                                   (", --skip_connections --categorical_node, bf16 autocast (the Linear outputs — hidden "
                                    "pre-activation, propagated rows, decoder P|Q — are stored / gathered as bfloat16; products and "
                                    "sums fp32-level)" if cfg5 else "") +
                                   (", one GPU's share of config 5 (6 of 50 genomes, m = 220 negatives per gene)"
                                    if args.workload == "cfg5slice" else ""),
                       "nodes": n, "sim_edges": e_sim, "neighbour_edges": e_nb,
                       "partition": "none" if world == 1 else f"destination-partitioned x{world} ({'node ranges with equal expected in-edge counts' if bounds else 'equal node ranges'}), rank-local generation, halo rows by all-to-all-v (conv_in needs no exchange; the decoder's P halo and its gradient travel on a side stream under the own-source pass / the by-target pass)",
                       "arithmetic": "fp32 storage and accumulation everywhere; the three per-edge decoder products run on the bf16 "
                                     "matrix pipe with fp32-exact operand handling (W2 h1: both operands split into three bf16 terms, "
                                     "six partial products; dL/dh1 and dL/dW2: the relu mask is the exact bf16 operand, the other "
                                     "operand is split three ways): logits within 1.6e-5 of an fp64 evaluation, gradients within 1e-6 "
                                     "of their scale (asserted: tests/test_hip_parity.py::test_decoder_training_kernels_vs_fp64); "
                                     "strict_fp32 = the f32-MFMA decoder; the node-level dense layers use the same exact "
                                     "three-term splits on the bf16 pipe (error vs fp64 1.2-2.5e-6 at unit scale, as the f32-MFMA "
                                     "product they replaced; tools/time_linear.py)",
                       "first_layer": "conv_in(embedding(x)) evaluated by linearity: A_hat (x w^T + 1 b^T) W^T + b_in = r a^T + s c^T + "
                                      "b_in; the node vectors r = A_hat x, s = A_hat 1 come from ONE similarity-graph propagate per graph "
                                      "(in the warm-up, cached like gcn_norm); its [N, H] rows are generated inside conv_out's dense "
                                      "kernels (forward product, weight gradient, and the three weighted column sums of dL/dh that are "
                                      "the layer's whole backward) and never stored; same logits / gradients as the layer-by-layer form "
                                      "up to fp32 re-association (tests/test_hip_parity.py::test_fused_embedding_layer_equals_layerwise_"
                                      "form_and_oracle, ::test_model_with_first_dense_layer_fused_equals_unfused_model)"
                                      if not cfg5 else "categorical embedding: layer by layer (propagate every step)",
                       "graph_build_s": round(t_gen, 3), "warmup_incl_structure_s": round(t_struct, 3),
                       "final_loss": float(loss.item())},
        }
        if world == 1 and not partitioned and not cfg5:
            # step-level check: the per-kernel algorithmic bytes of the step AS BUILT, summed (DESIGN.md §4 formulas)
            terms = step_alg_bytes(n, e_sim, e_nb, d, h, n_parts_s, n_parts_d)
            tot = float(sum(terms.values()))
            line["step_alg_bytes"] = tot
            line["step_hbm_frac"] = tot / step_s / HBM_PEAK
            line["step_alg_bytes_terms"] = terms
            line["step_alg_bytes_note"] = ("sum over the step's kernel launches of their algorithmic bytes (every gathered row "
                                           "counted once per edge, no cache credit; run-part rows (one per (16-tile chunk, key) run): S " + str(n_parts_s) + ", T " +
                                           str(n_parts_d) + "); step_hbm_frac = step_alg_bytes / ms_per_step / 8 TB/s")
        if t_dec:
            # Dominant kernel of the step: decoder_train16_kernel (csrc/decoder16.hip; S in DESIGN.md §4).
            # Algorithmic bytes per edge: ids 16 + P row 256 + Q row 256 + label 4 + logit 4 + record 20, plus 256 B per
            # (32-edge tile, source) part row.  Matrix-pipe work per 16 edges: 72 v_mfma_f32_16x16x32_bf16 + 12
            # v_mfma_f32_32x32x16_bf16 = 1 572 864 flop ISSUED, of which 3 products x 2*64*64 = 24 576 flop per edge
            # are useful (the rest is the three-way exact operand split).
            flop_issued = 1572864.0 / 16.0 * e_local
            flop_useful = 24576.0 * e_local
            b_dec = e_local * 556.0 + n_parts_s * 256.0
            ent = traffic_entry(prof, "decoder_train", args.workload, world, e_sim)
            line["roofline"] = {
                "bound": "hbm", "limiter": "issue",
                "kernel": "decoder_train16_kernel<fused loss, run sums> (largest kernel of the step)",
                "bound_note": "bound names the roof that achieved / peak / frac are priced against (8 TB/s HBM, the nearer of the "
                              "kernel's two hardware roofs); limiter names what the counters say binds the kernel: vector-"
                              "instruction issue plus matrix-pipe time (what_binds), neither roof",
                "achieved": b_dec / t_dec / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": b_dec / t_dec / HBM_PEAK,
                "frac_note": "algorithmic bytes per step's worth of S launches / launch time / 8 TB/s: HBM is the nearer of the "
                             "kernel's two roofs (useful matrix flops are at mfma.useful_frac of the bf16 peak)",
                "alg_bytes_per_launch": b_dec, "achievable_peak": 6300.0,
                "traffic": ent["bytes_fetch_doubled"] if ent else None,
                "traffic_note": ("rocprofv3 --pmc FETCH_SIZE (x2 on gfx950) + WRITE_SIZE of " + src_note + ", same workload / "
                                 "edge count / kernel name / kernel sources (sha " + str(ent.get("kernel_source_sha16")) +
                                 ", collected at " + str(ent.get("collected_at_head")) + "), not this run; L2-fabric bytes, "
                                 "Infinity-Cache hits included") if ent
                else "no counter collection in profiles/traffic.json matches this workload, edge count, kernel name and the "
                     "hash of this tree's kernel sources",
                "mfma": {"peak_tflops": 2500.0, "useful_tflops": flop_useful / t_dec / 1e12,
                         "useful_frac": flop_useful / t_dec / 1e12 / 2500.0,
                         "issued_tflops": flop_issued / t_dec / 1e12, "issued_mfma_frac": flop_issued / t_dec / 1e12 / 2500.0,
                         "note": "useful = 3 products x 2*64*64 flop per edge; issued = the bf16 MFMAs executed (x4: exact "
                                 "three-term operand splits)"},
                "avg_launch_ms": t_dec * 1e3, "launches_per_step": n_dec, "share_of_step": t_dec / step_s,
                "avg_launch_note": "duration of the step's S launches together (all of this rank's edges)" if n_dec > 1 else None,
                "what_binds": S_BINDS}
        if t_dgr:
            b_dg = e_local * 28.0 + n_parts_d * 256.0        # perm 4 + key 4 + record 20 per edge, part rows written
            ent = traffic_entry(prof, "decoder_dgrad", args.workload, world, e_sim)
            line["roofline_dgrad"] = {
                "bound": "hbm", "limiter": "issue",
                "kernel": "decoder_dgrad16_kernel (dL/dh1 run sums by target from the per-edge records)",
                "launches_per_step": n_dgr, "avg_launch_ms": t_dgr * 1e3, "alg_bytes_per_launch": b_dg,
                "achieved": b_dg / t_dgr / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": b_dg / t_dgr / HBM_PEAK,
                "mfma_tflops": 24 * 16384.0 / 16.0 * e_local / t_dgr / 1e12,
                "traffic": ent["bytes_fetch_raw"] if ent else None,
                "traffic_note": "FETCH_SIZE NOT doubled here (random 32-byte record gathers: the raw counter is one 64-B "
                                "sector per edge) + WRITE_SIZE, " + src_note + ", not this run"}
        per_step_prop = bool(t_prop)
        if not t_prop:
            t_prop = extra.get("prop_fwd_s")
        if t_prop:
            b_alg = spmm_alg_bytes(e_local, rows_local, f_spmm)
            ent = traffic_entry(prof, "spmm_fwd", args.workload, world, e_sim)
            traffic = ent["bytes_fetch_doubled"] if ent else None
            line["roofline_propagate"] = {
                "bound": "hbm", "kernel": f"spmm_row_kernel<{f_spmm}> (conv_in propagate fwd, the * kernel of SURVEY.md §8)",
                "achieved": b_alg / t_prop / 1e9, "peak": HBM_PEAK / 1e9, "achievable_peak": 6300.0, "unit": "GB/s",
                "frac": (b_alg / t_prop) / HBM_PEAK,
                "frac_note": "algorithmic bytes (SURVEY.md §8d: every gathered row counted once per edge, no cache credit) over "
                             "the 8 TB/s spec; > 1 because the 256 MB source table is served from L2 / Infinity Cache",
                "traffic": traffic,
                "traffic_note": "L2-fabric bytes (FETCH_SIZE x2 + WRITE_SIZE, Infinity-Cache hits included) from " +
                                src_note + ", same workload / edge count / kernel name, not this run",
                "fabric_rate_GBps": (traffic / t_prop / 1e9) if traffic else None,
                "fabric_rate_note": "7.4-7.9 TB/s is the guide's gather ceiling for a table of this size "
                                    "(MI355X_MICROARCH.md, indexed rows): the kernel sits at it",
                "alg_bytes_per_launch": b_alg, "avg_launch_ms": t_prop * 1e3,
                "share_of_step": (t_prop / step_s) if per_step_prop else 0.0,
                "in_step": "every step" if per_step_prop else
                           "once per graph (r = A_hat x, s = A_hat 1: conv_in(embedding(x)) = r a^T + s c^T + b_in by linearity, "
                           "functional._EmbedConvIn); timed here on the layer-by-layer model, outside the headline",
                "bwd_avg_launch_ms": extra.get("bwd_avg_launch_ms", (_avg("sim.bwd") or 0) * 1e3 or None)}
        for k_ in ("reference_loop", "reference_loop_literal", "eval", "general_features", "strict_fp32", "cfg5slice", "cfg2mb",
                   "cfg2mb_fresh", "extras_error"):
            if k_ in extra:
                line[k_] = extra[k_]
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args, d, h)
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
