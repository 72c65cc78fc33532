"""The fresh mini-batch step of `bench.py --workload cfg2mb_fresh` (a new Batch per step: collation + structure build +
train step) on one box, one process: (a) interleaved A/B timings of the host-side switches — boxes differ by 2x in host
speed, so only same-process numbers compare — and (b) a cProfile of the default variant: where the host time of a
launch-bound step goes.  usage: python tools/profile_fresh_step.py [steps] [rounds]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangnn_amd  # noqa: E402
from pangnn_amd import functional as PF, graph as G, simulate, subgraphs as SG  # noqa: E402
from pangnn_amd.train import make_optimizer, train_step  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
ds = simulate.simulate_subgraph_dataset(1000, 5, 0.3, 10, 2, seed=0, device=dev)       # bench.py: cfg2mb_fresh
n_train = int(len(ds) * 0.7)
spans = [(i, min(i + 32, n_train)) for i in range(0, n_train, 32)]
pw = ds.class_balance()
batches = [ds.batch(*s) for s in spans]

VARIANTS = {        # name: (small structure kernel, fused Adam, dispatcher ops, collation kernel, fresh batch per step)
    "fresh: default": (True, True, "auto", True, True),
    "fresh: general structure build (radix sorts + index-op plans)": (False, True, "auto", True, True),
    "fresh: index-op collation": (True, True, "auto", False, True),
    "fresh: foreach Adam": (True, False, "auto", True, True),
    "fresh: registered dispatcher ops forced on": (True, True, True, True, True),
    "fresh: as at the start of round 3 (general build, index-op collation, foreach Adam)": (False, False, False, False, True),
    "cached batches, eager launches": (True, True, "auto", True, False),
    "cached batches, eager launches, dispatcher ops forced on": (True, True, True, True, False),
}


def make(variant):
    small, fused, disp, collate, fresh = VARIANTS[variant]
    torch.manual_seed(0)
    model = pangnn_amd.AlternateGCN(dev, None, False, dims=[64, 64])
    opt = make_optimizer(model, fused=fused)

    def step(k):
        G.SMALL_STRUCTURE, PF.USE_DISPATCHER_OPS, SG.COLLATE_KERNEL = small, disp, collate
        if fresh:
            G.clear_cache()
            b = ds.batch(*spans[k % len(spans)])
        else:
            b = batches[k % len(batches)]
        return train_step(model, opt, b, b.y, pw)
    return step


fns = {v: make(v) for v in VARIANTS}
best = {v: float("inf") for v in VARIANTS}
for r in range(rounds):
    for v, fn in fns.items():
        for k in range(5):
            fn(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            fn(k)
        torch.cuda.synchronize()
        best[v] = min(best[v], (time.perf_counter() - t0) / steps * 1e3)
for v in VARIANTS:
    print(f"{best[v]:7.3f} ms/step  {v}")
G.SMALL_STRUCTURE, PF.USE_DISPATCHER_OPS, SG.COLLATE_KERNEL = True, "auto", True

fn = fns["fresh: default"]
pr = cProfile.Profile()
pr.enable()
for k in range(steps):
    fn(k)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(40)
st.sort_stats("tottime").print_stats(25)
