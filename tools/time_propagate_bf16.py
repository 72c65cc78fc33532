"""Diagnostic: conv_in-sized propagate (cfg 4 similarity graph) with fp32 vs bfloat16 / float16 row storage."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangnn_amd import functional as PF
from pangnn_amd.graph import structure_of
from pangnn_amd.simulate import simulate_graph
dev = torch.device("cuda:0")
g = simulate_graph(50000, 20, 0.2, 100, 20, seed=0, device=dev)
n, e = g.x.shape[0], g.edge_index.shape[1]
st = structure_of(g.edge_index, n, holder=g, name="sim")
norm = st.gcn_norm(g.edge_attr)
for F in (64, 128):
    x = torch.randn(n, F, device=dev)
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        xx = x.to(dt)
        for _ in range(2):
            PF.spmm_csr(st.by_dst, norm.by_dst, xx, n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            PF.spmm_csr(st.by_dst, norm.by_dst, xx, n)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        s = 4 if dt == torch.float32 else 2
        alg = e * (8 + F * s) + n * F * 4 + (n + 1) * 8        # result rows are fp32 in both cases
        print(f"F={F} {str(dt):15s} {ms:.3f} ms   algorithmic {alg / ms / 1e6:.0f} GB/s")
