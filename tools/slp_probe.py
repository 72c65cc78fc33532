"""Which build of csrc/decoder16.hip reproduces the SLP-packed-f32 miscompute of the S kernel?
Runs the training decoder of every build_variants/libpangnn_hip_<variant>.so (tools/slp_probe.sh) on ONE graph and
compares its outputs — per-(tile, source) part rows, per-edge records, logits, dL/dW2 — BIT FOR BIT with the product
library's.  One pass, a few launches per variant; prints where the differing part rows sit (wave slot, column block,
position of the run in its tile).      python tools/slp_probe.py [genes_per_genome] [launches] [variant ...]"""
import collections
import ctypes as C
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pangnn_amd import _lib, simulate            # noqa: E402
from pangnn_amd.graph import structure_of        # noqa: E402

dev = torch.device("cuda:0")
genes = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ONLY = set(sys.argv[3:])
g = simulate.simulate_graph(genes, 20, 0.2, 100, 20, seed=0, device=dev)
n, e = g.num_nodes, g.edge_index.shape[1]
st = structure_of(g.edge_index, n, holder=g, name="sim")
plan = st.runsum_plan(int(_lib.load().pangnn_decoder_chunk_tiles_for(st.num_edges)))
assert plan is not None
torch.manual_seed(0)
P, Q = torch.randn(n, 64, device=dev), torch.randn(n, 64, device=dev)
W2, b2, w3, b3 = torch.randn(64, 64, device=dev) / 8, torch.randn(64, device=dev), torch.randn(64, device=dev), torch.randn(1, device=dev)
gl = torch.randn(e, device=dev) / e
cv = torch.randn(64, device=dev)
extra = (g.edge_attr / 40).contiguous()
P16, Q16 = P.bfloat16().contiguous(), Q.bfloat16().contiguous()
pw = g.class_balance.reshape(1).float().contiguous()
print(f"N={n} E={e} tiles={(e + 31) // 32} parts={plan.n_parts}", flush=True)


def bind(path):
    lib = C.CDLL(path)
    for name in ("pangnn_decoder_train_mixed", "pangnn_decoder_train_workspace_bytes", "pangnn_last_error"):
        res, args = _lib.SIGNATURES[name]
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def run(lib, fused, skip=False, pq16=False, timing=None):
    logits = torch.zeros(e, device=dev)
    loss = torch.zeros(1, device=dev)
    rec = torch.zeros(e, 8, dtype=torch.int32, device=dev)
    parts = torch.zeros(plan.n_parts, 64, device=dev)
    gw2, gw3, gb3 = torch.zeros(64, 64, device=dev), torch.zeros(64, device=dev), torch.zeros(1, device=dev)
    wsb = lib.pangnn_decoder_train_workspace_bytes()
    ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
    gcv = torch.zeros(64, device=dev)
    p_, q_ = (P16, Q16) if pq16 else (P, Q)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    rc = lib.pangnn_decoder_train_mixed(
        p_.data_ptr(), 64, q_.data_ptr(), 64, 1 if pq16 else 0, n, st.edge_index.data_ptr(), e, e,
        extra.data_ptr() if skip else None, cv.data_ptr() if skip else None, W2.data_ptr(), b2.data_ptr(),
        w3.data_ptr(), b3.data_ptr(), 64, g.y.data_ptr() if fused else None, pw.data_ptr() if fused else None,
        e if fused else 0, None if fused else gl.data_ptr(), logits.data_ptr(), loss.data_ptr() if fused else None,
        rec.data_ptr(), parts.data_ptr(), plan.part_off.data_ptr(), gw2.data_ptr(), gw3.data_ptr(), gb3.data_ptr(),
        gcv.data_ptr() if skip else None, None, ws.data_ptr(), wsb, _lib.stream_ptr())
    ev1.record()
    assert rc == 0, lib.pangnn_last_error()
    torch.cuda.synchronize()
    if timing is not None:
        timing.append(ev0.elapsed_time(ev1))
    return dict(parts=parts, rec=rec[:, :5].contiguous(), logits=logits, gw2=gw2, gw3=gw3, gcv=gcv)


ref_lib = bind(_lib.LIB_PATH)
part_tile = torch.searchsorted(plan.part_off.long().contiguous(), torch.arange(plan.n_parts, device=dev), right=True) - 1
COMBOS = [(f, s_, h) for f in (True, False) for s_ in (False, True) for h in (False, True)]
if os.environ.get("PROBE_COMBOS"):           # e.g. "1,1,0;0,1,0" = (fused_loss, skip, bf16_tables) triples
    COMBOS = [tuple(bool(int(x)) for x in c.split(",")) for c in os.environ["PROBE_COMBOS"].split(";")]
for fused, skip, pq16 in COMBOS:
    ref = run(ref_lib, fused, skip, pq16)
    again = run(ref_lib, fused, skip, pq16)
    assert all(torch.equal(ref[k], again[k]) for k in ref), "the product library is not reproducible run to run"
    for path in sorted(glob.glob(os.path.join(ROOT, "build_variants", "libpangnn_hip_*.so"))):
        name = os.path.basename(path)[len("libpangnn_hip_"):-3]
        if ONLY and name not in ONLY:
            continue
        lib = bind(path)
        WAVES = 4 if name.endswith('onewave') else 8
        tot = collections.Counter()
        where = collections.Counter()
        ms = []
        for rep in range(REPS):
            out = run(lib, fused, skip, pq16, ms)
            for k in ref:
                a, b = out[k], ref[k]
                neq = (a.view(torch.int32) != b.view(torch.int32)) if a.dtype == torch.float32 else (a != b)
                tot[k] += int(neq.sum())
            bad = (out["parts"].view(torch.int32) != ref["parts"].view(torch.int32))
            rows = bad.any(1).nonzero().view(-1)
            tot["bad part rows"] += int(rows.numel())
            if rows.numel():
                t = part_tile[rows]
                for w_ in (t % WAVES).tolist():
                    where[f"wave{w_}"] += 1
                for r_ in (rows - plan.part_off.long()[t]).tolist():
                    where[f"run{min(r_, 3)}"] += 1
                cols = bad[rows].nonzero()[:, 1]
                for kb in (cols // 16).tolist():
                    where[f"kb{kb}"] += 1
                vals = out["parts"][rows][bad[rows]]
                refv = ref["parts"][rows][bad[rows]]
                where["zero"] += int((vals == 0).sum())
                where["elements"] += int(vals.numel())
                where[f"launch{rep}_rows"] = int(rows.numel())
                if rep == 0:
                    rel = ((vals - refv).abs() / refv.abs().clamp_min(1e-30))
                    where["rel_err_median_x1e6"] = int(rel.median() * 1e6)
                    where["rel_err_max_x1e6"] = int(rel.max().clamp(max=1e6) * 1e6)
                    # single-edge runs of 1-run tiles are one dL/dh1 row as the kernel produced it: look at a few
                    print("   examples (got, ref, got/ref):", [(f"{a:.4e}", f"{b:.4e}", f"{a / b if b else float('nan'):.4f}")
                                                             for a, b in zip(vals[:12].tolist(), refv[:12].tolist())], flush=True)
                    r0 = int(rows[0])
                    print("   first bad row", r0, "tile", int(part_tile[r0]), "cols", bad[r0].nonzero().view(-1).tolist()[:40], flush=True)
        ms = sorted(ms)[: max(len(ms) - 1, 1)]
        print(f"fused_loss={int(fused)} skip={int(skip)} bf16_tables={int(pq16)} {name:14s} S kernel {sum(ms) / len(ms):7.3f} ms; differing "
              f"elements over {REPS} launches x {(e + 31) // 32} tiles: {dict(tot)}  where: {dict(sorted(where.items()))}", flush=True)
