"""Which ELEMENTS of dL/dh1 does the SLP build of the S kernel get wrong, and what does it put there?
Needs the two PANGNN_D16_DEBUG builds of tools/slp_probe.sh (the kernel dumps its [E, 64] dL/dh1 rows right after the
second product's epilogue).  Compares the SLP build's rows with the no-SLP build's, bit for bit, and tests every wrong
element against the hypotheses "multiplied by the g_e of a neighbouring edge" / "mask of a neighbouring edge".
    python tools/slp_probe_rows.py [genes_per_genome]"""
import collections
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pangnn_amd import _lib, simulate            # noqa: E402
from pangnn_amd.graph import structure_of        # noqa: E402

dev = torch.device("cuda:0")
genes = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
g = simulate.simulate_graph(genes, 20, 0.2, 100, 20, seed=0, device=dev)
n, e = g.num_nodes, g.edge_index.shape[1]
st = structure_of(g.edge_index, n, holder=g, name="sim")
plan = st.runsum_plan(int(_lib.load().pangnn_decoder_chunk_tiles_for(st.num_edges)))
torch.manual_seed(0)
P, Q = torch.randn(n, 64, device=dev), torch.randn(n, 64, device=dev)
W2, b2, w3, b3 = torch.randn(64, 64, device=dev) / 8, torch.randn(64, device=dev), torch.randn(64, device=dev), torch.randn(1, device=dev)
gl = torch.randn(e, device=dev)
cv = torch.randn(64, device=dev)
extra = (g.edge_attr / 40).contiguous()
print(f"N={n} E={e}", flush=True)


def run(name):
    lib = C.CDLL(os.path.join(ROOT, "build_variants", f"libpangnn_hip_{name}.so"))
    for fn_name in ("pangnn_decoder_train_mixed", "pangnn_decoder_train_workspace_bytes", "pangnn_last_error"):
        res, args = _lib.SIGNATURES[fn_name]
        fn = getattr(lib, fn_name)
        fn.restype, fn.argtypes = res, args
    lib.pangnn_debug_set_v.argtypes = [C.c_void_p]
    dump = torch.full((e, 64), float("nan"), device=dev)
    assert lib.pangnn_debug_set_v(dump.data_ptr()) == 0
    logits = torch.zeros(e, device=dev)
    rec = torch.zeros(e, 8, dtype=torch.int32, device=dev)
    parts = torch.zeros(plan.n_parts, 64, device=dev)
    gw2, gw3, gb3, gcv = (torch.zeros(64, 64, device=dev), torch.zeros(64, device=dev), torch.zeros(1, device=dev),
                          torch.zeros(64, device=dev))
    wsb = lib.pangnn_decoder_train_workspace_bytes()
    ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
    rc = lib.pangnn_decoder_train_mixed(
        P.data_ptr(), 64, Q.data_ptr(), 64, 0, n, st.edge_index.data_ptr(), e, e, extra.data_ptr(), cv.data_ptr(),
        W2.data_ptr(), b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), 64, None, None, 0, gl.data_ptr(), logits.data_ptr(), None,
        rec.data_ptr(), parts.data_ptr(), plan.part_off.data_ptr(), gw2.data_ptr(), gw3.data_ptr(), gb3.data_ptr(),
        gcv.data_ptr(), None, ws.data_ptr(), wsb, _lib.stream_ptr())
    assert rc == 0, lib.pangnn_last_error()
    torch.cuda.synchronize()
    return dump, rec


ref, rec = run("noslp_dbg")
ref2, _ = run("noslp_dbg")
assert torch.equal(ref.view(torch.int32), ref2.view(torch.int32))
got, rec_s = run("slp_dbg")
assert torch.equal(rec, rec_s)
bad = got.view(torch.int32) != ref.view(torch.int32)
idx = bad.nonzero()
print("wrong elements", idx.shape[0], "of", e * 64, "in", int(bad.any(1).sum()), "edges", flush=True)
if idx.shape[0]:
    ee, kk = idx[:, 0], idx[:, 1]
    pos = (ee % 16).tolist()
    print("position of the edge in its 16-edge half tile:", dict(sorted(collections.Counter(pos).items())))
    print("column block (k // 16):", dict(sorted(collections.Counter((kk // 16).tolist()).items())))
    print("wave slot (tile % 8):", dict(sorted(collections.Counter(((ee // 32) % 8).tolist()).items())))
    ge = rec[:, 4].view(torch.float32)
    gv, rv = got[ee, kk].double(), ref[ee, kk].double()
    # hypotheses: the element was multiplied by the g_e of edge e + d instead of e (same dL/dh1 pre-factor, same mask)
    for d in (-3, -2, -1, 1, 2, 3):
        other = (ee + d).clamp(0, e - 1)
        pred = rv * (ge[other].double() / ge[ee].double())
        hit = ((pred - gv).abs() <= 1e-5 * gv.abs() + 1e-30) & (rv != 0)
        print(f"  got == ref * g_e[e{d:+d}] / g_e[e]: {int(hit.sum())} of {int((rv != 0).sum())} wrong elements whose reference is non-zero")
    print("  got == 0 where ref != 0:", int(((gv == 0) & (rv != 0)).sum()), "  got != 0 where ref == 0:", int(((gv != 0) & (rv == 0)).sum()))
    # the UNMASKED product of a masked-out element would be v * g_e: cannot be rebuilt here, but for ref == 0 elements
    # test whether got equals the value the SAME column holds at edge e + d
    for d in (-1, 1):
        other = (ee + d).clamp(0, e - 1)
        hit = (got[other, kk].double() == gv) & (gv != 0)
        print(f"  got == got[e{d:+d}][k]: {int(hit.sum())}")
    for j in range(min(10, idx.shape[0])):
        a, k = int(ee[j]), int(kk[j])
        print(f"   edge {a} (pos {a % 32}) k {k}: got {float(gv[j]):.6e} ref {float(rv[j]):.6e} g_e {float(ge[a]):.4e} "
              f"g_e[e-1] {float(ge[max(a - 1, 0)]):.4e} g_e[e+1] {float(ge[min(a + 1, e - 1)]):.4e}")
