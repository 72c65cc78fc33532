"""Diagnostic: the cfg-4 whole-graph train step launched eagerly vs replayed from ONE captured HIP graph
(train.GraphedTrainStep) — same kernels, same order; what the ~40 launches' gaps cost.  python tools/time_graphed_whole.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangnn_amd                                  # noqa: E402
from pangnn_amd import simulate                    # noqa: E402
from pangnn_amd.train import GraphedTrainStep, make_optimizer, train_step    # noqa: E402

dev = torch.device("cuda")
g = simulate.simulate_graph(50000, 20, 0.2, 100, 20, seed=0, device=dev)
e = g.edge_index.shape[1]
torch.manual_seed(0)
model = pangnn_amd.AlternateGCN(dev, None, False, dims=[64, 128], num_nodes=g.num_nodes)
opt = make_optimizer(model, capturable=True)
for _ in range(3):
    train_step(model, opt, g, g.y, g.class_balance)
torch.cuda.synchronize()


def timed(fn, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


for rnd in range(2):
    t_e, (l_e, _) = timed(lambda: train_step(model, opt, g, g.y, g.class_balance))
    print(f"eager   {t_e:.3f} ms/step  {e / t_e / 1e6:.3f} G edges/s  loss {float(l_e):.6f}")
gs = GraphedTrainStep(model, opt, g, g.y, g.class_balance, warmup=1)
for rnd in range(3):
    t_g, (l_g, _) = timed(gs)
    print(f"graphed {t_g:.3f} ms/step  {e / t_g / 1e6:.3f} G edges/s  loss {float(l_g):.6f}")
