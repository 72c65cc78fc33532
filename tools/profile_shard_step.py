"""Diagnostic (round 5): the aten / custom ops of ONE emulated rank's partitioned train step (rank 3 of 8, cfg 4) with the
Python call site that issued them — where the small fills / copies / cats / index ops of the shard path come from."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist                      # noqa: E402
from pangnn_amd import dist as pdist                  # noqa: E402
from pangnn_amd import simulate                       # noqa: E402
from pangnn_amd.train import make_optimizer           # noqa: E402

dev = torch.device("cuda:0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29555")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
R, W = int(os.environ.get("EMU_RANK", 3)), 8
bounds = pdist.balanced_bounds(50000, 20, W)
part = simulate.simulate_shard(50000, 20, 0.2, 100, 20, seed=0, device=dev, rank=R, world=W, bounds=bounds)
part.emulated_world = W
part.e_sim_total = part.e_sim_local * W
cb = torch.tensor((part.e_sim_local - part.n_pos_local) / max(part.n_pos_local, 1), dtype=torch.float32, device=dev)
torch.manual_seed(0)
model = pdist.DistAlternateGCN(dev, dims=[64, 128], part=part)
opt = make_optimizer(model)
for _ in range(3):
    pdist.train_step(model, opt, part, part.y, cb)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
STEPS = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(STEPS):
        pdist.train_step(model, opt, part, part.y, cb)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=70))
# call sites of the small glue ops
WATCH = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::cat", "aten::index_select", "aten::add", "aten::add_", "aten::index",
         "aten::contiguous", "aten::clone", "aten::_to_copy", "aten::zeros", "aten::index_put_", "aten::mul", "aten::sum")
sites = collections.Counter()
for ev in prof.events():
    if ev.name in WATCH and ev.stack:
        fr = [f for f in ev.stack if "pangnn_amd" in f or "bench.py" in f]
        sites[(ev.name, fr[0].split("/")[-1] if fr else ev.stack[0][-60:])] += 1
print(f"---- call sites per step (of {STEPS})")
for (name, site), c in sorted(sites.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{c / STEPS:6.2f}  {name:22s} {site}")
dist.destroy_process_group()
