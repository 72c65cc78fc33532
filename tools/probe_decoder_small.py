"""S / T kernel durations on mini-batch-sized edge lists, to be run under `rocprofv3 --kernel-trace` (durations are read
from the trace by tools/probe_decoder_small_report.py): 40 launches per variant, variants in a fixed order:
  E = 64, 2048, 6400 source-sorted with ~5 edges per source  x  (run sums in S, no run sums in S)"""
import sys

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangnn_amd import functional as PF          # noqa: E402
from pangnn_amd.graph import structure_of        # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
rng = np.random.default_rng(0)
w2 = (torch.randn(64, 64) * 0.1).to(dev)
b2, w3, b3 = (torch.randn(64) * 0.1).to(dev), (torch.randn(64) * 0.1).to(dev), torch.zeros(1, device=dev)
for e in (64, 2048, 6400):
    n = max(e // 5, 8)
    src = np.sort(rng.integers(0, n, e))
    dst = rng.integers(0, n, e)
    ei = torch.tensor(np.stack([src, dst]), dtype=torch.int64, device=dev)
    st = structure_of(ei, n)
    p, q = torch.randn(n, 64, device=dev), torch.randn(n, 64, device=dev)
    y = (torch.rand(e, device=dev) < 0.1).float()
    pw = torch.tensor(3.0, device=dev)
    for need_p in (True, False):
        for _ in range(40):
            PF._decoder_train16(p, q, st, None, None, w2, b2, w3, b3, y=y, pw=pw, denom=e, need_p=need_p)
        torch.cuda.synchronize()
print("done")
