"""Decoder kernels alone on the cfg-4 graph (or `--workload cfg5slice`): S (training), T (by-target pass), inference —
event-timed per launch, for same-box A/B of two library builds:

    python tools/time_decoder_ab.py                                   # the in-tree library
    PANGNN_HIP_LIB=build_variants/libpangnn_hip_r03.so python tools/time_decoder_ab.py
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangnn_amd import functional as PF          # noqa: E402
from pangnn_amd import simulate                  # noqa: E402
from pangnn_amd.graph import structure_of        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg4", choices=["cfg4", "cfg5slice"])
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--skip", action="store_true", help="with the skip feature (config 5's instances)")
ap.add_argument("--bf16", action="store_true", help="P | Q stored as bfloat16")
args = ap.parse_args()
dev = torch.device("cuda")
if args.workload == "cfg4":
    g = simulate.simulate_graph(50000, 20, 0.2, 100, 20, seed=0, device=dev)
else:
    g = simulate.simulate_graph(200000, 6, 0.1, 500, 50, seed=0, device=dev, mean_neg=220, adjacent_only=True)
n, e = g.num_nodes, g.edge_index.shape[1]
st = structure_of(g.edge_index, n, g, "sim")
torch.manual_seed(0)
pq = torch.randn(n, 128, device=dev)
if args.bf16:
    pq = pq.bfloat16()
par = [torch.randn(64, 64, device=dev) / 8, torch.randn(64, device=dev) * 0.1, torch.randn(64, device=dev) / 8,
       torch.randn(1, device=dev)]
ex = (g.edge_attr / 40).contiguous() if args.skip else None
cv = torch.randn(64, device=dev) if args.skip else None
pw = g.class_balance
PF.KERNEL_TIMER = {"dec.bwd": [], "dec.dgrad": [], "dec.fwd": []}
for _ in range(args.reps + 1):
    loss, logits = PF.decoder_loss_pq(pq, st, ex, cv, *par, g.y, pw, e)
with torch.no_grad():
    for _ in range(args.reps + 1):
        out = PF.decoder_mlp_pq(pq, st, ex, cv, *par)
torch.cuda.synchronize()
lib = os.environ.get("PANGNN_HIP_LIB", "in-tree")
for tag, name in (("dec.bwd", "S"), ("dec.dgrad", "T"), ("dec.fwd", "inference")):
    ts = sorted(a.elapsed_time(b) for a, b in PF.KERNEL_TIMER[tag][1:])
    print(f"[{lib}] {args.workload}{' skip' if args.skip else ''}{' bf16' if args.bf16 else ''} E={e} {name}: "
          f"min {ts[0]:.3f} median {ts[len(ts) // 2]:.3f} max {ts[-1]:.3f} ms over {len(ts)} launches")
print(f"[{lib}] loss {float(loss):.8f} max |inference - training logit| {float((out - logits).abs().max()):.3e}")
