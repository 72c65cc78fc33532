"""Operator API of the hot path, MI355X-native.

Mirrors what /root/reference/src/gnn.py and /root/reference/src/convolution.py import from
torch_geometric, with the same names, argument meaning and state_dict layout:

  * `MessagePassing(aggr='add'|'max')` with `propagate(edge_index, **kwargs)` that collects
    `x_j = x[edge_index[0]]` (source) / `x_i = x[edge_index[1]]` (target) for `message(...)` and
    aggregates at the target (flow 'source_to_target', node_dim 0) — convolution.py:3-23.
  * `GCNConv(in_channels, out_channels, add_self_loops=False)` called as
    `conv(x, edge_index[, edge_weight])` — gnn.py:100-102,129,135,138,147,158,165.  Parameters:
    `bias [out]` then `lin.weight [out,in]` (PyG >= 2.0 key order).
  * `EdgeConv(in_channels, out_channels)` — convolution.py:5-23, body unchanged.

Every gather / scatter / reduction runs in libpangnn_hip.so (pangnn_amd/csrc), and so do the dense
node-level products (`lin`: csrc/linear.hip, `pangnn_linear_*`, for in / out widths in {64, 128};
any other width goes to hipBLASLt through torch).  There is no CPU path.
"""
from __future__ import annotations

import inspect
import math
from typing import Optional

import torch
from torch import nn
from torch.nn import Linear, ReLU, Sequential as Seq

from . import _lib
from . import functional as PF
from .graph import EdgeStructure, structure_of


class MessagePassing(nn.Module):
    def __init__(self, aggr: str = "add", flow: str = "source_to_target", node_dim: int = 0):
        super().__init__()
        if aggr not in ("add", "sum", "max"):
            raise ValueError(f"aggr={aggr!r} not supported (reference uses 'add' and 'max')")
        if flow != "source_to_target" or node_dim != 0:
            raise ValueError("only flow='source_to_target', node_dim=0 (what the reference uses)")
        self.aggr = "add" if aggr == "sum" else aggr
        self._msg_params = list(inspect.signature(self.message).parameters)

    # -- user hooks -------------------------------------------------------------------
    def message(self, x_j):
        return x_j

    def update(self, aggr_out):
        return aggr_out

    # -- driver -----------------------------------------------------------------------
    def propagate(self, edge_index: torch.Tensor, size=None, **kwargs):
        x = kwargs.get("x")
        if x is None:
            raise ValueError("propagate(...) needs x=... (node features)")
        _lib.require_device(x, edge_index)
        n = x.shape[0] if size is None else int(size[1] if isinstance(size, (tuple, list)) else size)
        st = structure_of(edge_index, n)
        args = {}
        lifted = None
        for name in self._msg_params:
            if name.endswith("_i") or name.endswith("_j"):
                base = kwargs[name[:-2]]
                if base is None:
                    args[name] = None
                    continue
                if base is x:
                    if lifted is None:           # one fused gather gives both endpoints
                        lifted = PF.edge_gather_concat(x, st)
                    c = x.shape[1]
                    args[name] = lifted[:, :c] if name.endswith("_j") else lifted[:, c:]
                else:
                    both = PF.edge_gather_concat(base, st)
                    c = base.shape[1]
                    args[name] = both[:, :c] if name.endswith("_j") else both[:, c:]
            elif name in kwargs:
                args[name] = kwargs[name]
        msg = self.message(**args)
        out = PF.segment_max(msg, st) if self.aggr == "max" else PF.segment_sum(msg, st)
        return self.update(out)


class _GlorotLinear(nn.Module):
    """PyG `Linear(in, out, bias=False, weight_initializer='glorot')`: state_dict key `weight`."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        self.reset_parameters()

    def reset_parameters(self):
        a = math.sqrt(6.0 / (self.in_channels + self.out_channels))
        nn.init.uniform_(self.weight, -a, a)

    def forward(self, x, in_act: int = 0, out_dtype=None):
        return PF.linear(x, self.weight, None, in_act, out_dtype)


class GCNConv(MessagePassing):
    """out = D^-1/2 A_w D^-1/2 (x W^T) + b over the given edges, no self loops added.

    forward(x [N,in] f32, edge_index [2,E] i64, edge_weight [E] f32 | None) -> [N,out] f32.
    `graph=` (optional) lets the structure/normalisation cache live on a Data/Batch object."""

    def __init__(self, in_channels: int, out_channels: int, improved: bool = False, cached: bool = False,
                 add_self_loops: bool = False, normalize: bool = True, bias: bool = True):
        super().__init__(aggr="add")
        if add_self_loops or improved or not normalize:
            raise NotImplementedError("reference instantiates GCNConv(in, out, add_self_loops=False) only "
                                      "(src/gnn.py:100-102)")
        self.in_channels, self.out_channels = in_channels, out_channels
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))   # registered before `lin`: key order
        else:
            self.register_parameter("bias", None)
        self.lin = _GlorotLinear(in_channels, out_channels)

    def reset_parameters(self):
        self.lin.reset_parameters()
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, x, edge_index, edge_weight: Optional[torch.Tensor] = None, graph=None, name: str = "",
                in_elu: bool = False, dense_done: bool = False):
        """`in_elu=True`: `x` is the pre-activation of an ELU (alpha 1) the caller deferred — the layer computes
        conv(ELU(x)); when its dense part comes first the activation is folded into that kernel.
        `dense_done=True`: `x` already is `self.lin(...)` of the layer's input (the caller fused this layer's dense part into
        the producer of its input, functional._EmbedConvInLinear) — propagate and bias are what is left."""
        _lib.require_device(x, edge_index, edge_weight)
        st = edge_index if isinstance(edge_index, EdgeStructure) else \
            structure_of(edge_index, x.shape[0], holder=graph, name=name)
        if edge_weight is not None and edge_weight.shape[0] != st.num_edges:
            # reference passes the sim-edge weights with the union edge index (gnn.py:135); PyG would
            # raise on the length mismatch, so do we
            raise ValueError(f"edge_weight has {edge_weight.shape[0]} entries for {st.num_edges} edges")
        norm = st.gcn_norm(edge_weight)
        if dense_done:
            if self.in_channels < self.out_channels or x.shape[1] != self.out_channels:
                raise ValueError("dense_done=True needs a dense-first layer (in >= out) and x = lin(input)")
            return PF.propagate_any(x, self.bias, st, norm, edge_weight is None, tag=name or None)
        if not (x.dtype in PF.ROWS16 and self.in_channels >= self.out_channels):
            x = x.float()          # (a 16-bit-stored input of a dense-first layer is read as stored by the linear kernel)
        # Under bf16 / fp16 autocast (`accelerate` mixed precision, SURVEY.md §8b, src/setup.py:50) PyG's propagate gathers
        # 16-bit rows — the output of its autocast Linear — and multiplies / accumulates in fp32: the rows this layer
        # propagates are stored in that type (half the gather bytes, pangnn_spmm_csr_bf16 / _f16); the dense part stays fp32.
        rows16 = PF.autocast_rows_dtype(x)                       # torch.bfloat16 / torch.float16 / None
        if x.dtype in PF.ROWS16 and x.dtype != rows16:
            x = x.float()          # (rows stored in the other 16-bit type than the autocast one: one call never mixes the two)
        if in_elu and self.in_channels < self.out_channels:
            x, in_elu = torch.nn.functional.elu(x), False       # propagate comes first: nothing to fold into
        if self.in_channels < self.out_channels:
            # A_hat (x W^T) == (A_hat x) W^T: propagate on the narrower side (half the gather bytes for
            # 64 -> 128), then the dense layer with the bias fused
            # ... and their sum is what the autocast Linear casts to the autocast type first thing: cast once, stored
            # (out_dtype), so that the Linear reads 2-byte rows and the transposed propagate gathers its 16-bit gradient as stored
            agg = PF.propagate_any(x.to(rows16) if rows16 else x, None, st, norm, edge_weight is None,
                                   tag=name or None, out_dtype=rows16)
            # the autocast Linear's output is a 16-bit tensor (src/gnn.py:111 under mixed precision): stored as such
            return PF.linear(agg, self.lin.weight, self.bias, 0, rows16)
        # dense part first: under autocast its result is WRITTEN in the autocast type by the linear kernel (no separate cast)
        xw = self.lin(x, 1 if in_elu else 0, rows16)
        return PF.propagate_any(xw, self.bias, st, norm, edge_weight is None, tag=name or None)

    def message(self, x_j, edge_weight):            # kept for API parity; forward() is fused
        return edge_weight.view(-1, 1) * x_j


class EdgeConv(MessagePassing):
    """convolution.py:5-23: max-aggregated mlp(cat[x_i, x_j - x_i])."""

    def __init__(self, in_channels, out_channels):
        super().__init__(aggr="max")
        self.mlp = Seq(Linear(2 * in_channels, out_channels), ReLU(), Linear(out_channels, out_channels))

    def forward(self, x, edge_index):
        return self.propagate(edge_index, x=x)

    def message(self, x_i, x_j):
        tmp = torch.cat([x_i, x_j - x_i], dim=1)
        return self.mlp(tmp)
