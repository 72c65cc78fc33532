"""Round 5: what a hub row costs the wave-per-row propagate, with and without the long-row split (graph.CSR.long_rows).
A similarity graph of N nodes and E edges in which ONE target receives `hub` of them; forward propagate (F = 64, fp32 rows)
timed with events over 50 launches, same process, the split switched by graph.LONG_ROW."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangnn_amd import functional as PF   # noqa: E402
from pangnn_amd import graph as G         # noqa: E402

dev = torch.device("cuda")


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for n, e, hub in ((5000, 100_000, 0), (5000, 100_000, 20_000), (5000, 200_000, 100_000), (20_000, 1_000_000, 300_000),
                  (1_000_000, 20_000_000, 1_000_000)):
    gen = torch.Generator().manual_seed(0)
    src, dst = torch.randint(0, n, (e,), generator=gen), torch.randint(0, n, (e,), generator=gen)
    dst[:hub] = 7
    ei = torch.stack([src, dst]).to(dev)
    w = (torch.rand(e, generator=gen) * 80 + 1).to(dev)
    x = torch.randn(n, 64, device=dev)
    res = {}
    for name, limit in (("one wave per row", 10 ** 12), (f"rows > {G.LONG_ROW} as segments of ~sqrt(longest row)", G.LONG_ROW)):
        keep = G.LONG_ROW
        G.LONG_ROW = limit
        G.clear_cache()
        st = G.EdgeStructure(ei, n)
        nrm = st.gcn_norm(w)
        split = st.by_dst.long_rows() is not None
        res[name] = (timed(lambda: PF.spmm_csr(st.by_dst, nrm.by_dst, x, n)), split)
        G.LONG_ROW = keep
    line = "   ".join(f"{k}: {v[0]:8.1f} us{' (split)' if v[1] else ''}" for k, v in res.items())
    print(f"N {n:8d}  E {e:9d}  hub in-degree {hub:8d}:   {line}", flush=True)
