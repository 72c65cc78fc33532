"""world_size-2 (and 3) gloo runs of the partition + exchange logic of pangnn_amd/dist.py on CPU.

The HIP kernels cannot run here, so the per-rank arithmetic is a torch restatement plugged in through
the `ops=` hook (TEST-ONLY; the product only ever builds `HipOps`).  What is under test is everything
around the kernels: ownership of edges, local/global id remapping, padding, the all-gather /
reduce-scatter autograd pair, the global-mean loss scaling and the flat gradient all-reduce — checked
against the single-process oracle on the reference-built golden graph."""
import os
import sys
import tempfile
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, whole_graph_from_golden
from oracle import gcn_oracle as go


class TorchOps:
    """torch restatement of the four kernel entry points DistAlternateGCN uses"""

    def structure(self, edge_index, n_dst, n_src):
        return SimpleNamespace(ei=edge_index, n_dst=n_dst, n_src=n_src)

    def norm(self, st, w, gather_dis):
        src, dst = st.ei
        w = torch.ones(src.shape[0]) if w is None else w
        deg = torch.zeros(st.n_dst).scatter_add_(0, dst, w)
        dis = deg.pow(-0.5)
        dis[dis == float("inf")] = 0
        dis_src = gather_dis(dis)
        assert dis_src.shape[0] == st.n_src
        return dis_src[src] * w * dis[dst]

    def propagate(self, x_full, bias, st, norm, tag=None):
        src, dst = st.ei
        out = torch.zeros(st.n_dst, x_full.shape[1]).index_add(0, dst, norm.view(-1, 1) * x_full[src])
        return out if bias is None else out + bias

    def embed_propagate(self, x_tab, w, b, st, norm, tag=None):
        return self.propagate(x_tab.view(-1, 1) * w.view(1, -1) + b, None, st, norm)

    def embed_conv_in(self, x_tab, w, b, w_in, b_in, st, norm):
        """conv_in(embedding(x)) of the own rows (HipOps: by linearity, functional._EmbedConvIn)"""
        return torch.nn.functional.linear(self.embed_propagate(x_tab, w, b, st, norm), w_in, b_in)

    def embed_conv_in_linear(self, x_tab, w, b, w_in, b_in, w_out, bias_out, st, norm):
        """... followed by ELU and the next dense layer (HipOps: generated inside that layer's kernels)"""
        self.fused_first_dense_calls = getattr(self, "fused_first_dense_calls", 0) + 1
        h = self.embed_conv_in(x_tab, w, b, w_in, b_in, st, norm)
        return torch.nn.functional.linear(torch.nn.functional.elu(h), w_out, bias_out)

    def accumulate_back(self, g_local, back, plan):
        g_local.index_add_(0, plan.send_idx, back)

    def decoder(self, p_full, q_local, st, extra, cvec, w2, b2, w3, b3):
        src, dst = st.ei
        h = p_full[src] + q_local[dst]
        if extra is not None:
            h = h + extra.view(-1, 1) * cvec
        return torch.relu(torch.relu(h) @ w2.t() + b2) @ w3 + b3

    def decoder_train(self, table, q_local, st, extra, cvec, w2, b2, w3, b3, y, pos_weight, denom, after_p=None, out_q=None,
                      accumulate_q=False, out_p=None):
        """the one-pass training decoder of HipOps, by autograd: everything finished on return (out_p / out_q: where dL/dtable and
        dL/dq go; accumulate_q: dL/dq is added to out_q)"""
        leaves = [t.detach().clone().requires_grad_() for t in (table, q_local, w2, b2, w3, b3)]
        cv = None if cvec is None else cvec.detach().clone().requires_grad_()
        with torch.enable_grad():
            logits = self.decoder(leaves[0], leaves[1], st, extra, cv, *leaves[2:])
            loss = self.bce_sum_over(logits, y, pos_weight, denom)
            if logits.numel():
                grads = torch.autograd.grad(loss, leaves + ([cv] if cv is not None else []))
            else:
                grads = [torch.zeros_like(t) for t in leaves + ([cv] if cv is not None else [])]
        gp = grads[0] if out_p is None else out_p.copy_(grads[0])
        if after_p is not None:
            after_p(gp)
        gq = grads[1] if out_q is None else (out_q.add_(grads[1]) if accumulate_q else out_q.copy_(grads[1]))
        g_cv = grads[6] if cv is not None else None
        return (loss.detach(), logits.detach(), gp, gq, g_cv, grads[2], grads[3], grads[4], grads[5])

    def linear(self, x, w, b, in_act=0):
        return torch.nn.functional.linear(torch.nn.functional.elu(x) if in_act else x, w, b)

    def bce_sum_over(self, logits, labels, pos_weight, denom):
        return torch.nn.functional.binary_cross_entropy_with_logits(logits, labels, pos_weight=pos_weight,
                                                                    reduction="sum") / denom


def _worker(rank, world, init_file, flags, out_dir, exchange="halo", overlap=True, uneven=False, force=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    if force:          # one rank that exchanges its outer quarters with itself (dist.force_exchange)
        os.environ["PANGNN_FORCE_EXCHANGE"] = "1"
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    from pangnn_amd import dist as pdist
    g = whole_graph_from_golden("sim_200x4")
    if flags.get("union_edge_weights"):
        g.edge_attr = g.union_edge_attr
    else:
        # canonical (src, dst) order of construct.whole_graph (the fixture keeps the reference's set-iteration order)
        o = torch.argsort(g.edge_index[0] * g.x.shape[0] + g.edge_index[1], stable=True)
        g.edge_index, g.edge_attr, g.y = g.edge_index[:, o].contiguous(), g.edge_attr[o].contiguous(), g.y[o].contiguous()
    canonical = not flags.get("union_edge_weights")
    torch.manual_seed(0)
    flags = dict(flags)
    categorical = flags.pop("categorical_nodes", False)          # config 5: --skip_connections --categorical_node
    n = g.x.shape[0]
    oracle = go.AlternateGCNOracle(dims=(64, 128), flags=go.default_flags(**flags), categorical_nodes=categorical,
                                   num_nodes=n)
    with torch.no_grad():
        for k, p in oracle.named_parameters():
            if k.endswith("bias"):
                p.uniform_(-0.5, 0.5)
    if categorical:
        g.x = torch.arange(n)
    # uneven: node ranges of different sizes (what dist.balanced_bounds produces for a pan-genome)
    bounds = [0] + [n * (2 * r + 1) // (2 * world + 1) for r in range(1, world)] + [n] if uneven else None
    shard = pdist.partition_graph(g, rank, world, bounds)
    model = pdist.DistAlternateGCN(None, dims=[64, 128], ops=TorchOps(), exchange=exchange, part=shard,
                                   categorical_nodes=categorical, **flags)
    model.overlap = overlap
    # the exchange hides under the decoder exactly when there is one (halo exchange, source-sorted shard)
    assert model._overlap_ok(shard) == (overlap and exchange == "halo" and canonical)
    sd = oracle.state_dict()
    if categorical:
        # ranks draw DIFFERENT rows at construction although every rank seeds alike (keyed on the first owned node)
        first = model.embedding.weight[0].detach().clone()
        others = [torch.zeros_like(first) for _ in range(world)]
        dist.all_gather(others, first)
        assert world == 1 or not torch.equal(others[0], others[1])
    # reference-layout checkpoint in, this rank's rows of the categorical embedding kept (a rank holds the embedding
    # rows of its own nodes); and back out: every rank reassembles the reference's [N, D] table
    model.load_full_state_dict(sd)
    if categorical:
        assert torch.equal(model.embedding.weight[: shard.hi - shard.lo], sd["embedding.weight"][shard.lo:shard.hi])
    back = model.full_state_dict()
    assert list(back) == list(sd)
    for k in sd:
        assert torch.equal(back[k], sd[k]), k
    assert (shard.n_pad is None if uneven else shard.n_pad == shard.n_local * world)
    assert shard.e_sim_total == g.edge_index.shape[1] and shard.n_local == shard.hi - shard.lo or not uneven
    pw = torch.tensor(float((g.y == 0).sum() / g.y.sum()))
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    opt_o = torch.optim.Adam(oracle.parameters(), lr=1e-3)

    # forward parity in the whole graph's edge order
    out = model(shard)
    full = pdist.gather_logits(out.detach(), shard)
    ref = oracle(g)
    assert torch.allclose(full, ref.detach(), atol=1e-4, rtol=1e-4), (full - ref).abs().max()

    # one train step: the loss and every all-reduced gradient must match the single-process oracle
    # (gradients, not post-Adam parameters: Adam's g/(|g|+eps) amplifies rounding noise on ~0 entries)
    lo, _ = go.train_step(oracle, opt_o, g, g.y, pw)
    ll, _ = pdist.train_step(model, opt, shard, shard.y, pw)
    tot = ll.clone()
    dist.all_reduce(tot)
    assert abs(float(tot) - float(lo)) < 1e-5
    for (k, p), (_, q) in zip(model.named_parameters(), oracle.named_parameters()):
        if q.grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        if categorical and k == "embedding.weight":              # sharded rows, not all-reduced
            scale = float(q.grad.abs().max()) + 1e-12
            assert torch.allclose(p.grad[: shard.hi - shard.lo], q.grad[shard.lo:shard.hi], atol=1e-4 * scale + 1e-8,
                                  rtol=1e-3), k
            continue
        scale = float(q.grad.abs().max()) + 1e-12
        assert torch.allclose(p.grad, q.grad, atol=1e-4 * scale + 1e-8, rtol=1e-3), (k, (p.grad - q.grad).abs().max())
    # the scalar-feature default / base topologies take the fused first-two-layers operator on every shard (forward of the
    # parity check + forward of the train step); union (propagate-first consumer at these dims) and categorical do not
    fused_calls = getattr(model.ops, "fused_first_dense_calls", 0)
    assert fused_calls == (0 if (categorical or flags.get("union_edge_weights")) else 2), fused_calls
    if exchange == "halo":
        plan = model._plan(shard, "sim")
        # [sources of lower ranks | own sources | sources of higher ranks]
        ts = plan.edge_index[0]
        assert plan.sorted_by_src == canonical and 0 <= plan.e_lo <= plan.e_hi <= ts.numel()
    if exchange == "halo" and canonical:
        assert bool((ts[: plan.e_lo] < plan.n_low).all()) and bool((ts[plan.e_hi:] >= plan.n_low + plan.n_local).all())
        mid = ts[plan.e_lo:plan.e_hi]
        assert bool(((mid >= plan.n_low) & (mid < plan.n_low + plan.n_local)).all())
        a, b = plan.split_edge_values(shard.y)
        assert torch.equal(plan.merge_edge_values(a, b), shard.y)
        # the halo is exactly the set of remote sources this rank's edges reference
        src = shard.edge_index[0]
        qf = shard.n_local // 4 if force else 0            # forced self-exchange: the outer quarters count as remote
        rem = (src < shard.lo + qf) | (src >= shard.lo + shard.n_local - qf)
        assert plan.n_halo == int(torch.unique(src[rem]).numel())
        assert plan.n_halo < n - shard.n_local or world == 2 or force
        if force:
            assert plan.any_exchange and plan.n_halo > 0 and plan.n_low > 0 and plan.e_lo > 0 and plan.e_hi < ts.numel()
            assert plan.send_splits == [plan.n_halo] and plan.recv_splits == [plan.n_halo]
        assert sum(plan.recv_splits) == plan.n_halo and int(plan.edge_index[0].max()) < plan.n_table
    # owned-edge bookkeeping: every similarity edge has exactly one owner
    cnt = shard.owned_mask.to(torch.int32).clone()
    dist.all_reduce(cnt)
    assert int(cnt.min()) == 1 and int(cnt.max()) == 1
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("flags", [dict(), dict(skip_connections=True), dict(base_model=True),
                                   dict(union_edge_weights=True),
                                   dict(skip_connections=True, categorical_nodes=True)],
                         ids=["default", "skip", "base", "union", "cfg5-skip-categorical"])
def test_partitioned_model_matches_single_process_oracle(world, flags):
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "rdzv")
        mp.spawn(_worker, args=(world, init_file, flags, d, "halo"), nprocs=world, join=True)
        assert all(os.path.exists(os.path.join(d, f"ok{r}")) for r in range(world))


@pytest.mark.parametrize("flags", [dict(), dict(skip_connections=True)], ids=["default", "skip"])
def test_partitioned_model_without_overlap(flags):
    """PANGNN_DIST_OVERLAP=0: the table is gathered first and the decoder runs once over the shard"""
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "rdzv")
        mp.spawn(_worker, args=(2, init_file, flags, d, "halo", False), nprocs=2, join=True)
        assert all(os.path.exists(os.path.join(d, f"ok{r}")) for r in range(2))


@pytest.mark.parametrize("flags", [dict(), dict(skip_connections=True, categorical_nodes=True)],
                         ids=["default", "cfg5-skip-categorical"])
def test_partitioned_model_on_unequal_node_ranges(flags):
    """node ranges of different sizes (edge-balanced partition): owners by boundary search, per-rank row counts"""
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "rdzv")
        mp.spawn(_worker, args=(3, init_file, flags, d, "halo", True, True), nprocs=3, join=True)
        assert all(os.path.exists(os.path.join(d, f"ok{r}")) for r in range(3))


@pytest.mark.parametrize("uneven", [False, True], ids=["equal-ranges", "unequal-ranges"])
def test_partitioned_model_on_eight_ranks(uneven):
    """the rank count the 8-GPU node runs (gloo on the CPU): halo plans with empty sides, 100-node shards, both halo
    exchanges of the decoder under its two passes"""
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "rdzv")
        mp.spawn(_worker, args=(8, init_file, dict(skip_connections=True), d, "halo", True, uneven), nprocs=8, join=True)
        assert all(os.path.exists(os.path.join(d, f"ok{r}")) for r in range(8))


@pytest.mark.parametrize("flags,overlap", [(dict(), True), (dict(), False), (dict(union_edge_weights=True), True),
                                           (dict(skip_connections=True, categorical_nodes=True), True)],
                         ids=["default", "default-gather-first", "union", "cfg5-skip-categorical"])
def test_one_rank_forced_self_exchange_matches_oracle(flags, overlap):
    """PANGNN_FORCE_EXCHANGE=1 (the hook tests/test_dist_gpu.py uses to run the N > 1 code over RCCL on a one-GPU box): a
    single rank treats the outer quarters of its node range as remote rows owned by itself — every exchange of the
    partitioned path runs (here over gloo) and the result is still the single-process oracle's"""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(1, os.path.join(d, "rdzv"), flags, d, "halo", overlap, False, True), nprocs=1, join=True)
        assert os.path.exists(os.path.join(d, "ok0"))


@pytest.mark.parametrize("flags", [dict(), dict(union_edge_weights=True)], ids=["default", "union"])
def test_partitioned_model_allgather_exchange(flags):
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "rdzv")
        mp.spawn(_worker, args=(2, init_file, flags, d, "allgather"), nprocs=2, join=True)
        assert all(os.path.exists(os.path.join(d, f"ok{r}")) for r in range(2))


def _ag_worker(rank, world, init_file, out_dir):
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    from pangnn_amd.dist import AllGatherRows
    x = torch.full((3, 2), float(rank + 1), requires_grad=True)
    full = AllGatherRows.apply(x, None)
    assert full.shape == (3 * world, 2) and float(full[3 * (world - 1), 0]) == world
    coeff = torch.arange(3 * world, dtype=torch.float32).view(-1, 1) * (rank + 1)
    (full * coeff).sum().backward()
    # d/dx_local = sum over ranks r' of coeff_{r'}[my rows] = my rows' index * sum(r'+1)
    expect = torch.arange(3 * rank, 3 * rank + 3, dtype=torch.float32).view(-1, 1) * sum(range(1, world + 1))
    assert torch.equal(x.grad, expect.expand(3, 2))
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_rows_backward_is_reduce_scatter():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_ag_worker, args=(2, os.path.join(d, "rdzv"), d), nprocs=2, join=True)
        assert os.path.exists(os.path.join(d, "ok0")) and os.path.exists(os.path.join(d, "ok1"))


def test_emulated_rank_plan_is_the_real_ranks_layout():
    """dist.HaloPlan(..., emulated_world=W) in a ONE-process job (bench.py --emulate-rank r --of W): the remote-source set, the
    table layout [halo of lower ranks | own | halo of higher ranks], the re-indexed edge list and the own-source / halo-source
    split are those of the real rank; only the exchange is a self-exchange of the same row count."""
    from pangnn_amd import dist as pdist
    g = whole_graph_from_golden("cfg2_sim_1000x5")
    order = torch.argsort(g.edge_index[0], stable=True)        # source-sorted, as the simulator / construct.py emit their lists
    g.edge_index, g.edge_attr, g.y = g.edge_index[:, order].contiguous(), g.edge_attr[order], g.y[order]
    world = 4
    port = 29791
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        bounds = pdist.balanced_bounds(1000, 5, world)
        for r in range(world):
            shard = pdist.partition_graph(g, r, world, bounds)
            plan = pdist.HaloPlan(shard.edge_index, shard.lo, shard.n_local, None, None, bounds, emulated_world=world)
            src = shard.edge_index[0]
            remote = (src < shard.lo) | (src >= shard.hi)
            need = torch.unique(src[remote])
            assert plan.n_halo == need.numel() and plan.n_table == shard.n_local + need.numel()
            assert plan.send_splits == [plan.n_halo] and plan.recv_splits == [plan.n_halo] and plan.send_idx.numel() == plan.n_halo
            assert sum(plan.peer_counts) == plan.n_halo and plan.peer_counts[r] == 0
            # the table in global ids: what row k of the table stands for
            table_ids = torch.cat([need[need < shard.lo], torch.arange(shard.lo, shard.hi), need[need >= shard.hi]])
            assert torch.equal(table_ids[plan.edge_index[0]], src) and torch.equal(plan.edge_index[1], shard.edge_index[1])
            assert plan.sorted_by_src
            own = plan.edge_index[0][plan.e_lo:plan.e_hi]
            assert bool(((own >= plan.n_low) & (own < plan.n_low + shard.n_local)).all())
            rest = torch.cat([plan.edge_index[0][:plan.e_lo], plan.edge_index[0][plan.e_hi:]])
            assert bool(((rest < plan.n_low) | (rest >= plan.n_low + shard.n_local)).all())
            assert plan.any_exchange == (plan.n_halo > 0)
    finally:
        dist.destroy_process_group()
