import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, f"{name}.npz"))


def whole_graph_from_golden(name, device="cpu"):
    f = load_golden(name)
    g = SimpleNamespace(
        x=torch.from_numpy(f["whole_x"]).to(device),
        edge_index=torch.from_numpy(f["whole_edge_index"]).to(device),
        edge_attr=torch.from_numpy(f["whole_edge_attr"]).to(device),
        y=torch.from_numpy(f["whole_y"]).to(device),
        neighbour_edge_index=torch.from_numpy(f["whole_neighbour_edge_index"]).to(device),
    )
    # union graph exactly as dataset.py:373-381 builds it for --union_edge_weights
    g.union_edge_index = torch.cat([g.edge_index, g.neighbour_edge_index], dim=1)
    g.union_edge_attr = torch.cat([g.edge_attr, torch.ones(g.neighbour_edge_index.shape[1], device=device)])
    return g


def sub_graphs_from_golden(name, count=None):
    """list of per-ortholog-group sub-graphs (CPU tensors) as the reference built them."""
    f = load_golden(name)
    no, eo, bo = f["sub_node_off"], f["sub_edge_off"], f["sub_nb_off"]
    out = []
    k = len(no) - 1 if count is None else min(count, len(no) - 1)
    for i in range(k):
        n = int(no[i + 1] - no[i])
        out.append(SimpleNamespace(
            x=torch.ones(n, 1),
            edge_index=torch.from_numpy(f["sub_edge_index"][:, eo[i]:eo[i + 1]].copy()),
            edge_attr=torch.from_numpy(f["sub_edge_attr"][eo[i]:eo[i + 1]].copy()),
            y=torch.from_numpy(f["sub_y"][eo[i]:eo[i + 1]].copy()),
            neighbour_edge_index=torch.from_numpy(f["sub_neighbour_edge_index"][:, bo[i]:bo[i + 1]].copy()),
        ))
    return out


def copy_graph(g, device):
    out = SimpleNamespace()
    for k, v in g.__dict__.items():
        setattr(out, k, v.to(device) if torch.is_tensor(v) else v)
    return out


def random_graph(n, e, seed=0, self_loops=True, dup=True, isolated=0.1, hub=None):
    """adversarial random COO: isolated targets, duplicate edges, self loops, an optional hub row"""
    gen = torch.Generator().manual_seed(seed)
    live = max(1, int(n * (1 - isolated)))
    src = torch.randint(0, n, (e,), generator=gen)
    dst = torch.randint(0, live, (e,), generator=gen)
    if hub is not None and e > 0:
        k = min(hub, e)
        dst[:k] = live - 1                     # one row with a huge in-degree
    if dup and e > 4:
        src[1], dst[1] = src[0], dst[0]
        src[3], dst[3] = src[2], dst[2]
    if self_loops and e > 6:
        src[5] = dst[5]
    w = torch.rand(e, generator=gen) * 80 + 1
    return torch.stack([src, dst]), w
