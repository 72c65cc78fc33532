"""Shader clock and package power while ONE decoder kernel runs back to back (round 5): the S / T / inference kernels are
timed per launch with events while a side thread samples `rocm-smi --showclocks --showpower` (sysfs reads, no privileges).
Question: is the chip at its power / current limit under these kernels (profiles/r05_mfma_shape_mix.txt: the sustained shader
clock falls from 2.35 GHz under matrix instructions alone to 1.5-1.6 GHz under a mixed matrix + vector load)?"""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangnn_amd import functional as PF          # noqa: E402
from pangnn_amd import simulate                  # noqa: E402
from pangnn_amd.graph import structure_of        # noqa: E402

dev = torch.device("cuda")
g = simulate.simulate_graph(50000, 20, 0.2, 100, 20, seed=0, device=dev)
n, e = g.num_nodes, g.edge_index.shape[1]
st = structure_of(g.edge_index, n, g, "sim")
torch.manual_seed(0)
pq = torch.randn(n, 128, device=dev)
par = [torch.randn(64, 64, device=dev) / 8, torch.randn(64, device=dev) * 0.1, torch.randn(64, device=dev) / 8,
       torch.randn(1, device=dev)]
x = torch.randn(n, 64, device=dev)
nrm = st.gcn_norm(g.edge_attr)
samples, stop = [], threading.Event()


def sampler():
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = re.search(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)Mhz", out)
            pw = re.search(r"Power \(W\):\s*([\d.]+)", out)
            samples.append((time.perf_counter(), int(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None))
        except Exception as ex:                               # keep going: the timing below is the main result
            samples.append((time.perf_counter(), None, repr(ex)))
        time.sleep(0.2)


def phase(name, fn, seconds=4.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        k += 20
    t1 = time.perf_counter()
    mine = [(c, p) for t, c, p in samples if t0 + 0.5 < t < t1 and c is not None]
    clk = sorted(c for c, _ in mine)
    pws = sorted(p for _, p in mine if isinstance(p, float))
    print(f"{name:34s} {1e3 * (t1 - t0) / k:8.3f} ms/launch   sclk MHz median {clk[len(clk) // 2] if clk else None} "
          f"(min {clk[0] if clk else None}, max {clk[-1] if clk else None}, {len(clk)} samples)   power W median "
          f"{pws[len(pws) // 2] if pws else None} (max {pws[-1] if pws else None})", flush=True)


th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(1.0)
idle = [(c, p) for _, c, p in samples if c is not None]
print("idle:", idle[-1] if idle else samples[-1:], flush=True)
phase("S + T (decoder_loss_pq)", lambda: PF.decoder_loss_pq(pq, st, None, None, *par, g.y, g.class_balance, e))
with torch.no_grad():
    phase("inference decoder", lambda: PF.decoder_mlp_pq(pq, st, None, None, *par))
    phase("propagate (spmm_row_kernel<64>)", lambda: PF.spmm_csr(st.by_dst, nrm.by_dst, x, n))
stop.set()
raw = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower"], capture_output=True, text=True).stdout
print(raw[-1500:])
