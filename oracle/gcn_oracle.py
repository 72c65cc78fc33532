"""CPU oracle for the panGNN hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  Nothing under `pangnn_amd/` imports it; the product path has no CPU fallback.

PARITY UNPINNED (arithmetic): the convolution arithmetic of the reference is not in the reference
tree — it is third-party `torch_geometric` (GCNConv / gcn_norm / MessagePassing / Batch), which is
neither vendored nor installed here, and the reference ships no tests, golden logits or usable
checkpoints.  This file restates PyG 2.x's documented semantics for exactly the
configuration the reference instantiates, anchored on the reference's call sites:

  GCNConv(in, out, add_self_loops=False)          /root/reference/src/gnn.py:100-102
  conv(x, edge_index[, edge_weight])              /root/reference/src/gnn.py:129,135,138,147,158,165
  AlternateGCN wiring / decoders                  /root/reference/src/gnn.py:84-207
  EdgeConv(MessagePassing, aggr='max')            /root/reference/src/convolution.py:5-23
  DataLoader/Batch collation                      /root/reference/pangnn.py:121,152-155,180
  BCEWithLogitsLoss(pos_weight) + Adam(1e-3)      /root/reference/pangnn.py:88,98,194-216

To keep the restatement from being merely self-consistent it is cross-checked in tests against an
independent dense formulation (`gcn_conv_dense`: A_hat built as a dense [N,N] matrix) and in fp64.

The graph-CONSTRUCTION half of the oracle (oracle/construct_oracle.py) IS pinned: it is checked
against fixtures produced by the reference's own construction code (tests/golden/*.npz).
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Optional, Sequence

import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------------------------------
# gcn_norm  (PyG torch_geometric.nn.conv.gcn_conv.gcn_norm, add_self_loops=False,
#            improved=False, flow='source_to_target'); call path gnn.py:158 -> GCNConv.forward
# --------------------------------------------------------------------------------------
def gcn_norm(edge_index: torch.Tensor, edge_weight: Optional[torch.Tensor], num_nodes: int,
             dtype=torch.float32) -> torch.Tensor:
    row, col = edge_index[0], edge_index[1]
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.shape[1], dtype=dtype, device=edge_index.device)
    edge_weight = edge_weight.to(dtype)
    deg = torch.zeros(num_nodes, dtype=dtype, device=edge_index.device)
    deg.scatter_add_(0, col, edge_weight)            # weighted IN-degree at the target
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float("inf"), 0.0)       # isolated targets -> 0
    return dis[row] * edge_weight * dis[col]


def propagate_add(x: torch.Tensor, edge_index: torch.Tensor, norm: torch.Tensor) -> torch.Tensor:
    """MessagePassing.propagate with GCNConv.message (norm.view(-1,1) * x_j) and aggr='add',
    issued exactly as PyG's edge_index mode does: index_select -> mul -> index_add_."""
    x_j = x.index_select(0, edge_index[0])           # gather SOURCE rows   [E,F]
    msg = norm.view(-1, 1) * x_j                     #                      [E,F]
    out = torch.zeros(x.shape, dtype=msg.dtype, device=x.device)   # PyG scatter: src.new_zeros (dtype of the messages)
    out.index_add_(0, edge_index[1], msg)            # scatter-add at TARGET
    return out


def gcn_conv(x, edge_index, edge_weight, weight, bias):
    """GCNConv.forward: norm -> lin (no bias) -> propagate -> + bias."""
    norm = gcn_norm(edge_index, edge_weight, x.shape[0], dtype=x.dtype)
    xw = x @ weight.t()
    out = propagate_add(xw, edge_index, norm)
    if bias is not None:
        out = out + bias
    return out


def gcn_conv_dense(x, edge_index, edge_weight, weight, bias):
    """Independent formulation: build A_hat[c, r] = d_r^-1/2 w_rc d_c^-1/2 densely (duplicates
    accumulate), out = A_hat @ (x W^T) + b.  Small graphs only."""
    n = x.shape[0]
    row, col = edge_index[0], edge_index[1]
    w = torch.ones(edge_index.shape[1], dtype=x.dtype) if edge_weight is None else edge_weight.to(x.dtype)
    a = torch.zeros(n, n, dtype=x.dtype)
    a.index_put_((col, row), w, accumulate=True)     # A[target, source]
    deg = a.sum(dim=1)                               # weighted in-degree of each target
    dis = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))
    a_hat = dis.view(-1, 1) * a * dis.view(1, -1)
    out = a_hat @ (x @ weight.t())
    return out + bias if bias is not None else out


# --------------------------------------------------------------------------------------
# modules with the reference's state_dict key names  
# --------------------------------------------------------------------------------------
class _LinNoBias(nn.Module):
    """PyG `Linear(in, out, bias=False, weight_initializer='glorot')` -> key `lin.weight`."""
    def __init__(self, i, o):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i))
        a = math.sqrt(6.0 / (i + o))                 # glorot uniform
        nn.init.uniform_(self.weight, -a, a)

    def forward(self, x):
        return x @ self.weight.t()


class GCNConvOracle(nn.Module):
    def __init__(self, in_channels, out_channels, add_self_loops=False):
        super().__init__()
        assert not add_self_loops, "reference only instantiates add_self_loops=False (gnn.py:100-102)"
        self.bias = nn.Parameter(torch.zeros(out_channels))     # registered BEFORE lin: key order
        self.lin = _LinNoBias(in_channels, out_channels)

    def forward(self, x, edge_index, edge_weight=None):
        return gcn_conv(x, edge_index, edge_weight, self.lin.weight, self.bias)


def default_flags(**kw):
    f = dict(union_edge_weights=False, base_model=False, skip_connections=False, decoder="mlp",
             neighbours=1)
    f.update(kw)
    return SimpleNamespace(**f)


class AlternateGCNOracle(nn.Module):
    """gnn.py:84-207 wired over the oracle GCNConv.  Flags that the reference reads from the global
    `args` (gnn.py:111,128,132,143,171-180) are constructor kwargs here."""

    def __init__(self, device=None, dataset=None, categorical_nodes=False, dims=(64, 128), flags=None,
                 num_nodes: Optional[int] = None):
        super().__init__()
        self.flags = flags or default_flags()
        d, h = dims
        if categorical_nodes:
            # Reference is broken here (gnn.py:93 takes len() of a python list of graphs and feeds
            # float ones to nn.Embedding).  Build-defined semantics: x = arange(N) positions.
            assert num_nodes is not None
            self.embedding = nn.Embedding(num_nodes, d)
        else:
            self.embedding = nn.Linear(1, d)
        self.conv_in = GCNConvOracle(d, h)
        self.conv_hidden = GCNConvOracle(h, h)
        self.conv_out = GCNConvOracle(h, d)
        self.linear_out = nn.Linear(h, d)
        self.activation_fct = nn.ELU()
        self.mlp = nn.Sequential(
            nn.Linear(2 * d + (1 if self.flags.skip_connections else 0), d), nn.ReLU(),
            nn.Linear(d, d), nn.ReLU(), nn.Linear(d, 1))

    def encode(self, graph):
        fl = self.flags
        h = self.embedding(graph.x)
        if fl.union_edge_weights:
            h = self.activation_fct(self.conv_in(h, graph.union_edge_index, graph.edge_attr))
            for _ in range(max(fl.neighbours - 2, 1)):
                h = self.activation_fct(self.conv_hidden(h, graph.union_edge_index, graph.edge_attr))
            h = self.activation_fct(self.conv_out(h, graph.union_edge_index))
        elif fl.base_model:
            h = self.activation_fct(self.conv_in(h, graph.edge_index, graph.edge_attr))
            h = self.activation_fct(self.linear_out(h))
        else:
            h = self.activation_fct(self.conv_in(h, graph.edge_index, graph.edge_attr))
            h = self.activation_fct(self.conv_out(h, graph.neighbour_edge_index))
        return h

    def forward(self, graph):
        fl = self.flags
        z = self.encode(graph)
        ei = graph.edge_index
        out = None
        if "mlp" in fl.decoder:
            parts = [z[ei[0]], z[ei[1]]]
            if fl.skip_connections:
                parts.append(graph.edge_attr[: ei.shape[1]].unsqueeze(1))
            out = self.mlp(torch.cat(parts, dim=1)).squeeze(-1)
        if "cosine" in fl.decoder:
            out = F.cosine_similarity(z[ei[0]], z[ei[1]], dim=1)
        if "dot" in fl.decoder:
            # Reference `decode` (gnn.py:202-204) is z[src] @ z[dst]: a shape error unless E == D.
            # Build-defined semantics: per-edge dot product.
            out = (z[ei[0]] * z[ei[1]]).sum(dim=1)
        return out


# --------------------------------------------------------------------------------------
# EdgeConv (convolution.py:5-23): aggr='max', message = mlp(cat[x_i, x_j - x_i])
# --------------------------------------------------------------------------------------
def segment_max(msg: torch.Tensor, index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """PyG scatter(..., reduce='max') semantics: rows that receive nothing are 0."""
    out = torch.full((num_nodes, msg.shape[1]), float("-inf"), dtype=msg.dtype)
    out = out.scatter_reduce(0, index.view(-1, 1).expand_as(msg), msg, reduce="amax", include_self=True)
    return torch.where(torch.isinf(out) & (out < 0), torch.zeros_like(out), out)


class EdgeConvOracle(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(2 * in_channels, out_channels), nn.ReLU(),
                                 nn.Linear(out_channels, out_channels))

    def forward(self, x, edge_index):
        x_i = x[edge_index[1]]                       # target
        x_j = x[edge_index[0]]                       # source
        msg = self.mlp(torch.cat([x_i, x_j - x_i], dim=1))
        return segment_max(msg, edge_index[1], x.shape[0])


# --------------------------------------------------------------------------------------
# Batch.from_data_list (PyG) as used by DataLoader at pangnn.py:152-153
# --------------------------------------------------------------------------------------
def collate(graphs: Sequence) -> SimpleNamespace:
    """x / edge_attr / y concatenated on dim 0; every attribute whose name contains 'index' is
    concatenated on dim -1 with the cumulative node count added; `batch` and `ptr` added."""
    xs, eas, ys, offs = [], [], [], [0]
    idx_attrs = {}
    for g in graphs:
        n = g.x.shape[0]
        for name in ("edge_index", "neighbour_edge_index", "union_edge_index"):
            t = getattr(g, name, None)
            if t is not None:
                idx_attrs.setdefault(name, []).append(t + offs[-1])
        xs.append(g.x); eas.append(g.edge_attr)
        if getattr(g, "y", None) is not None:
            ys.append(g.y)
        offs.append(offs[-1] + n)
    out = SimpleNamespace(x=torch.cat(xs, 0), edge_attr=torch.cat(eas, 0),
                          y=torch.cat(ys, 0) if ys else None)
    for name, parts in idx_attrs.items():
        setattr(out, name, torch.cat(parts, dim=-1))
    out.ptr = torch.tensor(offs, dtype=torch.long)
    out.batch = torch.repeat_interleave(torch.arange(len(graphs)), out.ptr[1:] - out.ptr[:-1])
    out.num_graphs = len(graphs)
    return out


# --------------------------------------------------------------------------------------
# train step (pangnn.py:194-216) on CPU: the `cpu_baseline` leg of bench.py times this
# --------------------------------------------------------------------------------------
def train_step(model: nn.Module, optimizer, graph, labels, pos_weight: torch.Tensor):
    optimizer.zero_grad()
    out = model(graph)
    loss = F.binary_cross_entropy_with_logits(out, labels, pos_weight=pos_weight)
    loss.backward()
    optimizer.step()
    return loss.detach(), out.detach()
