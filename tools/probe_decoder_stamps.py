"""Where a SMALL launch of the S kernel spends its time: wall_clock64() stamps of wave 0 of workgroup 0 (diagnostic build:
decoder16.hip with -DPANGNN_D16_STAMP, loaded through PANGNN_HIP_LIB).  Every stamp waits for the wave's outstanding memory
operations first, so the differences are completed work, not issue time."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangnn_amd import _lib, functional as PF      # noqa: E402
from pangnn_amd.graph import structure_of          # noqa: E402

NAMES = ["entry", "weights staged (this wave)", "barrier", "prologue (ids, rows requested, scalars)",
         "half 0: h1 formed", "half 0: P1 + logit", "half 0: loss, masks, P3, records", "half 0: P2 + run sums",
         "half 1: h1 formed", "half 1: P1 + logit", "half 1: loss, masks, P3, records", "half 1: P2 + run sums",
         "lane reductions", "tree over the waves", "slab written"]
dev = torch.device("cuda:0")
lib = _lib.load()
lib.pangnn_debug_set_stamps.argtypes = [ctypes.c_void_p]
stamps = torch.zeros(32, dtype=torch.int64, device=dev)
assert lib.pangnn_debug_set_stamps(stamps.data_ptr()) == 0
rng = np.random.default_rng(0)
w2 = (torch.randn(64, 64) * 0.1).to(dev)
b2, w3, b3 = (torch.randn(64) * 0.1).to(dev), (torch.randn(64) * 0.1).to(dev), torch.zeros(1, device=dev)
for e in (64, 6400):
    n = max(e // 5, 8)
    src, dst = np.sort(rng.integers(0, n, e)), rng.integers(0, n, e)
    st = structure_of(torch.tensor(np.stack([src, dst]), dtype=torch.int64, device=dev), n)
    p, q = torch.randn(n, 64, device=dev), torch.randn(n, 64, device=dev)
    y, pw = (torch.rand(e, device=dev) < 0.1).float(), torch.tensor(3.0, device=dev)
    rows = []
    for it in range(6):
        stamps.zero_()
        PF._decoder_train16(p, q, st, None, None, w2, b2, w3, b3, y=y, pw=pw, denom=e)
        torch.cuda.synchronize()
        rows.append(stamps[:15].cpu().numpy().astype(np.float64) / 100.0)        # 100 MHz -> us
    t = np.median(np.stack(rows[2:]), axis=0)
    print(f"E = {e}: S kernel, wave 0 of workgroup 0, microseconds since entry (median of 4 launches)")
    for i, name in enumerate(NAMES):
        print(f"  {t[i] - t[0]:7.2f}  (+{(t[i] - t[i - 1]) if i else 0.0:5.2f})  {name}")
