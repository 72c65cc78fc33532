"""`Data` / `Batch` / `DataLoader` — the input layout of the hot path.

Restates the part of torch_geometric the reference relies on (dataset.py:302-310,380-384 build
`Data(x, edge_index, edge_attr, y)` + `neighbour_edge_index` / `union_edge_index`;
pangnn.py:121,152-155 wraps lists of them in a `DataLoader`): a mini-batch is the disjoint union
of its graphs — `x`, `edge_attr`, `y` concatenated on dim 0, every attribute whose name contains
"index" concatenated on the last dim with the cumulative node count added, plus `batch` / `ptr`.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.utils.data


class Data:
    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, **kwargs):
        self.x, self.edge_index, self.edge_attr, self.y = x, edge_index, edge_attr, y
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    def keys(self) -> List[str]:
        return [k for k, v in self.__dict__.items() if not k.startswith("_") and v is not None]

    def to(self, device, non_blocking: bool = False):
        out = self.__class__.__new__(self.__class__)
        for k, v in self.__dict__.items():
            if k.startswith("_pangnn"):
                continue                      # device structures are rebuilt for the new tensors
            out.__dict__[k] = v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v
        return out

    def pin_memory(self):
        for k, v in self.__dict__.items():
            if torch.is_tensor(v) and not v.is_cuda:
                self.__dict__[k] = v.pin_memory()
        return self

    def __repr__(self):
        parts = [f"{k}={list(v.shape)}" if torch.is_tensor(v) else f"{k}=..." for k, v in self.__dict__.items()
                 if not k.startswith("_") and v is not None]
        return f"{self.__class__.__name__}({', '.join(parts)})"


class Batch(Data):
    @classmethod
    def from_data_list(cls, graphs: Sequence[Data]) -> "Batch":
        if len(graphs) == 0:
            raise ValueError("empty batch")
        keys = [k for k in graphs[0].keys()]
        offs = [0]
        for g in graphs:
            offs.append(offs[-1] + g.num_nodes)
        out = cls()
        for k in keys:
            vals = [getattr(g, k) for g in graphs]
            if not all(torch.is_tensor(v) for v in vals):
                setattr(out, k, vals)         # e.g. gene_lst becomes a list of lists (dataset.py:313)
                continue
            if "index" in k:
                setattr(out, k, torch.cat([v + o for v, o in zip(vals, offs[:-1])], dim=-1))
            else:
                setattr(out, k, torch.cat(vals, dim=0))
        ptr = torch.tensor(offs, dtype=torch.long)
        out.ptr = ptr
        out.batch = torch.repeat_interleave(torch.arange(len(graphs)), ptr[1:] - ptr[:-1])
        out.num_graphs = len(graphs)
        return out


class DataLoader(torch.utils.data.DataLoader):
    """torch_geometric.loader.DataLoader as called at pangnn.py:121,152-153: a `torch.utils.data.DataLoader` over a list of
    `Data` whose collate function is `Batch.from_data_list` (what PyG's loader is), so `accelerator.prepare(loader)`
    (pangnn.py:122,155) wraps it like any torch loader and moves every `Batch` to the device through its `.to()`.
    `device=` does that move here for loops that do not go through accelerate."""

    def __init__(self, dataset: Sequence[Data], batch_size: int = 1, shuffle: bool = False, device=None, **kwargs):
        kwargs.pop("collate_fn", None)
        self._device = device
        super().__init__(list(dataset), batch_size=int(batch_size), shuffle=bool(shuffle), collate_fn=self._collate, **kwargs)

    def _collate(self, graphs):
        b = Batch.from_data_list(graphs)
        if self._device is not None and self.num_workers == 0:
            b = b.to(self._device, non_blocking=bool(self.pin_memory))
        return b
