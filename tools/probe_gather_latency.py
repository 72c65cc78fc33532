"""Diagnostic: how much of the S / T / inference kernels' time is gather latency?  The same kernels on three edge lists of the
cfg-4 size and source structure (74 sources' edges in a row) whose TARGETS are (a) uniformly random inside the next genome
(the cfg-4 law: Q rows come from L2 / Infinity Cache / HBM), (b) the source + 1 (Q rows arrive from the same lines as
their neighbours: every gather an L1 / L2 hit), (c) random over ALL nodes (worst case).  python tools/probe_gather_latency.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangnn_amd import functional as PF          # noqa: E402
from pangnn_amd.graph import EdgeStructure       # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
n, e = 1_000_000, 74_000_000
gen = torch.Generator(device=dev).manual_seed(0)
src = torch.arange(e, device=dev) // 74
cases = {
    "a: targets random inside the next genome (cfg-4 law)":
        (src // 50000 * 50000 + 50000 + torch.randint(0, 50000, (e,), device=dev, generator=gen)).clamp_(max=n - 1),
    "b: target = source + 1 (cache-resident gathers)": (src + 1).clamp_(max=n - 1),
    "c: targets random over all nodes": torch.randint(0, n, (e,), device=dev, generator=gen),
}
pq = torch.randn(n, 128, device=dev)
par = [torch.randn(64, 64, device=dev) / 8, torch.randn(64, device=dev) * 0.1, torch.randn(64, device=dev) / 8,
       torch.randn(1, device=dev)]
y = (torch.rand(e, device=dev) < 0.03).float()
pw = torch.tensor(30.0, device=dev)
for name, dst in cases.items():
    st = EdgeStructure(torch.stack([src, dst]).contiguous(), n, hints={"valid_ids": True, "sorted_by_src": True})
    PF.KERNEL_TIMER = {"dec.bwd": [], "dec.dgrad": [], "dec.fwd": []}
    for _ in range(5):
        loss, logits = PF.decoder_loss_pq(pq, st, None, None, *par, y, pw, e)
    with torch.no_grad():
        for _ in range(5):
            PF.decoder_mlp_pq(pq, st, None, None, *par)
    torch.cuda.synchronize()
    med = lambda tag: sorted(a.elapsed_time(b) for a, b in PF.KERNEL_TIMER[tag][1:])[len(PF.KERNEL_TIMER[tag][1:]) // 2]   # noqa: E731
    print(f"{name}: S {med('dec.bwd'):.3f} ms  T {med('dec.dgrad'):.3f} ms  inference {med('dec.fwd'):.3f} ms")
    del st
    PF.KERNEL_TIMER = None
