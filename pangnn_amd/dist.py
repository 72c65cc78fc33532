"""Whole-graph training over several MI355X: 1-D destination partition + RCCL exchanges over xGMI.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL).  The reference has no
multi-device graph mode at all; this is the partitioning BASELINE.json's north
star asks for, for the simulated graphs that do not need to — or cannot — sit on one GPU.

Layout.  Nodes are split into `world` contiguous, equally sized ranges (the last one padded with
isolated rows so that every collective is a fixed-size all-gather / reduce-scatter).  Rank r owns
  * the rows [lo, hi) of every node tensor,
  * every edge whose TARGET it owns (similarity and neighbour graphs alike): a CSR over local target
    rows with GLOBAL source ids, so the propagate kernel reads an all-gathered source table and
    writes only local rows,
  * the supervised edges (= similarity edges) it owns, hence its slice of logits and labels.
Exchanges per step (fp32, F = feature width of the exchanged tensor):
  * forward, per GCN layer: all-gather of the local X W^T rows  -> [N_pad, F]
  * backward, per GCN layer: reduce-scatter(sum) of dL/d(X W^T) [N_pad, F] -> local rows
  * decoder: all-gather of P = z W1a^T rows (sources may be remote; Q = z W1b^T + b is indexed by
    the local target only), reduce-scatter of dL/dP in backward
  * parameters: ONE flat all-reduce of all gradients (54 k floats = 216 KB) per step
  * gcn_norm (once per graph): all-gather of deg^-1/2 [N_pad]
xGMI is a full mesh of point-to-point links, so fixed-size all-gather / reduce-scatter (each shard
crosses one link once) are the right primitives; nothing here is a ring of small messages.

The arithmetic on each rank is the same HIP kernels as the single-GPU path (`HipOps`).  The
`ops=` hook exists so the partition / exchange logic can be exercised on CPU with gloo, where the
tests plug in a torch restatement of the kernels (tests/test_dist_cpu.py).  The product never
constructs anything but `HipOps`.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import functional as PF
from .gnn import AlternateGCN
from .graph import EdgeStructure


# --------------------------------------------------------------------------------------
# collectives with autograd
# --------------------------------------------------------------------------------------
def _all_gather_rows(x: torch.Tensor, group) -> torch.Tensor:
    world = dist.get_world_size(group)
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out


def _reduce_scatter_rows(full: torch.Tensor, group) -> torch.Tensor:
    world = dist.get_world_size(group)
    n_local = full.shape[0] // world
    out = torch.empty((n_local,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
    full = full.contiguous()
    if dist.get_backend(group) == "gloo":          # gloo has no reduce_scatter: all-reduce + slice
        dist.all_reduce(full, group=group)
        r = dist.get_rank(group)
        out.copy_(full[r * n_local:(r + 1) * n_local])
    else:
        dist.reduce_scatter_tensor(out, full, group=group)
    return out


class AllGatherRows(torch.autograd.Function):
    """[n_local, F] on every rank -> [world * n_local, F]; backward = reduce-scatter (sum)."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return _all_gather_rows(x, group)

    @staticmethod
    def backward(ctx, g):
        return _reduce_scatter_rows(g, ctx.group), None


# --------------------------------------------------------------------------------------
# partition
# --------------------------------------------------------------------------------------
def partition_graph(g, rank: int, world: int):
    """Slice a whole graph (x, edge_index, edge_attr, y, neighbour_edge_index[, union_edge_index]) into
    rank's shard.  Index tensors keep GLOBAL source ids and get LOCAL target ids."""
    n = int(g.x.shape[0])
    n_local = (n + world - 1) // world
    lo, hi = rank * n_local, min((rank + 1) * n_local, n)
    dev = g.x.device

    def own(ei):
        m = (ei[1] >= lo) & (ei[1] < hi)
        loc = ei[:, m].clone()
        loc[1] -= lo
        return loc.contiguous(), m

    ei, m_sim = own(g.edge_index)
    x = torch.zeros((n_local,) + tuple(g.x.shape[1:]), dtype=g.x.dtype, device=dev)
    x[: hi - lo] = g.x[lo:hi]
    shard = SimpleNamespace(
        x=x, edge_index=ei, edge_attr=g.edge_attr[: g.edge_index.shape[1]][m_sim].contiguous(),
        y=g.y[m_sim].contiguous() if getattr(g, "y", None) is not None else None,
        n_local=n_local, n_pad=n_local * world, n_global=n, lo=lo, hi=hi, rank=rank, world=world,
        e_sim_local=int(ei.shape[1]), e_sim_total=int(g.edge_index.shape[1]), owned_mask=m_sim)
    if getattr(g, "neighbour_edge_index", None) is not None:
        shard.neighbour_edge_index, _ = own(g.neighbour_edge_index)
    if getattr(g, "union_edge_index", None) is not None:
        shard.union_edge_index, m_u = own(g.union_edge_index)
        if g.edge_attr.shape[0] == g.union_edge_index.shape[1]:   # dataset.py:380: edge_attr = union weights
            shard.union_edge_attr = g.edge_attr[m_u].contiguous()
    return shard


# --------------------------------------------------------------------------------------
# compute back end (HIP) — the only one the product uses
# --------------------------------------------------------------------------------------
class HipOps:
    def structure(self, edge_index, n_dst, n_src):
        return EdgeStructure(edge_index, n_dst, n_src)

    def norm(self, st, edge_weight, gather_dis):
        return st.gcn_norm(edge_weight, gather_dis)

    def propagate(self, x_full, bias, st, norm, tag=None):
        return PF.propagate(x_full, bias, st, norm, tag)

    def decoder(self, p_full, q_local, st, extra, cvec, w2, b2, w3, b3):
        return PF.decoder_mlp(p_full, q_local, st, extra, cvec, w2, b2, w3, b3)

    def pair_rows(self, z_full, z_local, st):
        """(z[src], z[dst]) per owned edge for the cosine / dot decoders"""
        raise NotImplementedError("cosine / dot decoders are single-GPU only")


class DistAlternateGCN(AlternateGCN):
    """AlternateGCN (same parameters / state_dict) evaluated on a destination-partitioned shard."""

    def __init__(self, device=None, dims=(64, 128), part=None, group=None, ops=None, **kw):
        super().__init__(device, None, False, dims=dims, **kw)
        self.group = group
        self.ops = ops or HipOps()
        self._structs = {}
        if self.flags.decoder != "mlp":
            raise NotImplementedError("partitioned mode implements the mlp decoder")
        if dims[0] != 64 and isinstance(self.ops, HipOps):
            raise NotImplementedError("partitioned HIP decoder is built for node_dim 64")

    # structures / norms are per shard tensor and cached on the shard object
    def _st(self, shard, name):
        cache = shard.__dict__.setdefault("_dist_structs", {})
        if name not in cache:
            ei = {"sim": shard.edge_index, "nb": getattr(shard, "neighbour_edge_index", None),
                  "union": getattr(shard, "union_edge_index", None)}[name]
            cache[name] = self.ops.structure(ei, shard.n_local, shard.n_pad)
        return cache[name]

    def _norm(self, shard, name, weight, wkey):
        cache = shard.__dict__.setdefault("_dist_norms", {})
        key = (name, wkey)
        if key not in cache:
            gather = lambda d: _all_gather_rows(d, self.group)       # noqa: E731
            cache[key] = self.ops.norm(self._st(shard, name), weight, gather)
        return cache[key]

    def _linear(self, x, w, b):
        return PF.linear(x, w, b) if x.is_cuda else F.linear(x, w, b)      # CPU only in the gloo tests

    def _conv(self, conv, h_local, shard, name, weight, wkey, tag):
        st, norm = self._st(shard, name), self._norm(shard, name, weight, wkey)
        if conv.in_channels < conv.out_channels:
            # propagate (and exchange) on the narrower side: half the all-gather bytes for 64 -> 128
            h_full = AllGatherRows.apply(h_local, self.group)
            agg = self.ops.propagate(h_full, None, st, norm, tag)
            return self._linear(agg, conv.lin.weight, conv.bias)
        xw = self._linear(h_local, conv.lin.weight, None)
        xw_full = AllGatherRows.apply(xw, self.group)
        return self.ops.propagate(xw_full, conv.bias, st, norm, tag)

    def encode(self, shard):
        fl, act = self.flags, self.activation_fct
        h = shard.x.float().view(-1, 1) * self.embedding.weight.view(1, -1) + self.embedding.bias
        if fl.union_edge_weights:
            w = shard.union_edge_attr
            h = act(self._conv(self.conv_in, h, shard, "union", w, "w", "union"))
            for _ in range(max(fl.neighbours - 2, 1)):
                h = act(self._conv(self.conv_hidden, h, shard, "union", w, "w", "union"))
            h = act(self._conv(self.conv_out, h, shard, "union", None, "1", "union"))
        elif fl.base_model:
            h = act(self._conv(self.conv_in, h, shard, "sim", shard.edge_attr, "w", "sim"))
            h = act(self._linear(h, self.linear_out.weight, self.linear_out.bias))
        else:
            h = act(self._conv(self.conv_in, h, shard, "sim", shard.edge_attr, "w", "sim"))
            h = act(self._conv(self.conv_out, h, shard, "nb", None, "1", "nb"))
        return h

    def decode_mlp(self, z, shard):
        fl = self.flags
        d = z.shape[1]
        lin0 = self.mlp[0]
        w = lin0.weight
        p = self._linear(z, w[:, :d].contiguous(), None)
        q = self._linear(z, w[:, d:2 * d].contiguous(), lin0.bias)
        p_full = AllGatherRows.apply(p, self.group)
        extra = shard.edge_attr if fl.skip_connections else None
        cvec = w[:, 2 * d].contiguous() if fl.skip_connections else None
        return self.ops.decoder(p_full, q, self._st(shard, "sim"), extra, cvec, self.mlp[2].weight,
                                self.mlp[2].bias, self.mlp[4].weight.view(-1), self.mlp[4].bias)

    def forward(self, shard):
        return self.decode_mlp(self.encode(shard), shard)

    def sync_gradients(self):
        """one flat all-reduce (sum) of every parameter gradient: 216 KB at default dims"""
        grads = [p.grad for p in self.parameters() if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, group=self.group)
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n


def train_step(model: DistAlternateGCN, optimizer, shard, labels, pos_weight):
    """pangnn.py:194-216 on a shard: the loss is the GLOBAL mean, so every rank divides its local sum
    by the global edge count and the gradient all-reduce is a plain sum."""
    optimizer.zero_grad(set_to_none=True)
    out = model(shard)
    if out.is_cuda:
        loss = PF.bce_with_logits(out, labels, pos_weight, denom=shard.e_sim_total)
    else:                                               # gloo CPU tests
        loss = F.binary_cross_entropy_with_logits(out, labels, pos_weight=pos_weight,
                                                  reduction="sum") / shard.e_sim_total
    loss.backward()
    model.sync_gradients()
    optimizer.step()
    return loss.detach(), out.detach()


def gather_logits(out_local: torch.Tensor, shard, group=None) -> torch.Tensor:
    """logits of all ranks back in the whole graph's edge order (evaluation / tests)."""
    world = dist.get_world_size(group)
    counts = torch.zeros(world, dtype=torch.long, device=out_local.device)
    counts[shard.rank] = out_local.shape[0]
    dist.all_reduce(counts, group=group)
    mx = int(counts.max())
    buf = torch.zeros(mx, dtype=out_local.dtype, device=out_local.device)
    buf[: out_local.shape[0]] = out_local
    allb = _all_gather_rows(buf, group).view(world, mx)
    masks = _all_gather_rows(shard.owned_mask.to(torch.uint8), group).view(world, -1).bool()
    full = torch.empty(shard.e_sim_total, dtype=out_local.dtype, device=out_local.device)
    for r in range(world):
        full[masks[r]] = allb[r, : int(counts[r])]
    return full
