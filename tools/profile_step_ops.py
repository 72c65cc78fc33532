"""Diagnostic: aten / custom ops of one cfg-4 train step by GPU time (torch.profiler)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pangnn_amd
from pangnn_amd.simulate import simulate_graph
from pangnn_amd.train import make_optimizer, train_step
dev = torch.device("cuda:0")
g = simulate_graph(50000, 20, 0.2, 100, 20, seed=0, device=dev)
torch.manual_seed(0)
model = pangnn_amd.AlternateGCN(dev, None, False, dims=[64, 128])
opt = make_optimizer(model)
for _ in range(3):
    train_step(model, opt, g, g.y, g.class_balance)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(2):
        train_step(model, opt, g, g.y, g.class_balance)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=60))
print("---- aten ops on edge-sized tensors")
for ev in prof.key_averages(group_by_input_shape=True):
    if ev.key.startswith("aten::") and any(len(sh) and max(sh) > 10_000_000 for sh in (ev.input_shapes or []) if isinstance(sh, (list, tuple))):
        print(ev.key, ev.input_shapes, ev.count, round(ev.device_time_total / 1e3, 3), "ms")
