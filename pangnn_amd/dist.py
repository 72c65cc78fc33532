"""Whole-graph training over several MI355X: 1-D destination partition + RCCL exchanges over xGMI.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL).  The reference has no
multi-device graph mode at all; this is the partitioning BASELINE.json's north
star asks for, for the simulated graphs that do not need to — or cannot — sit on one GPU.

Layout.  Nodes are split into `world` contiguous ranges: by default (`bounds=balanced_bounds(...)`, what bench.py
uses) ranges with equal EXPECTED in-edge counts, exchanged by halo all-to-all-v; equal node ranges (the last one
padded with isolated rows) when `bounds` is None — the only layout the fixed-size all-gather / reduce-scatter
exchange accepts.  Rank r owns
  * the rows [lo, hi) of every node tensor,
  * every edge whose TARGET it owns (similarity and neighbour graphs alike): a CSR over local target
    rows with GLOBAL source ids, so the propagate kernel reads an all-gathered source table and
    writes only local rows,
  * the supervised edges (= similarity edges) it owns, hence its slice of logits and labels.
Exchanges per step (fp32, F = width of the exchanged rows; conv layers exchange on their narrower
side, so F = 64 everywhere at default dims):
  * default `exchange="halo"`: per graph a one-time plan lists, for every peer, exactly the owned rows
    that peer's edges reference (boundary-node embeddings).  Forward per GCN layer / decoder:
    ONE all-to-all-v of those rows; backward: the reverse all-to-all-v of their gradients, summed into
    the owner's rows through a fixed CSR (no atomics, reproducible).  For the simulated pan-genome
    graphs the halo is one genome on each side of a rank's range (negatives only link adjacent genomes,
    node ids are genome-major) and 1 row per side for the neighbour graph, i.e. ~10 % of an all-gather.
  * `exchange="allgather"`: all-gather of the local rows -> [N_pad, F], reduce-scatter of the gradient
    (the right primitive when the halo is dense).
  * parameters: ONE flat all-reduce of all gradients (54 k floats = 216 KB) per step
  * gcn_norm (once per graph): the same exchange applied to deg^-1/2.
xGMI is a full mesh of point-to-point links: an all-to-all-v sends each peer's rows over the direct link
to that peer, once; nothing here is a ring of small messages.

The arithmetic on each rank is the same HIP kernels as the single-GPU path (`HipOps`).  The
`ops=` hook exists so the partition / exchange logic can be exercised on CPU with gloo, where the
tests plug in a torch restatement of the kernels (tests/test_dist_cpu.py).  The product never
constructs anything but `HipOps`.
"""
from __future__ import annotations

import os
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
def _host_staged(group, *tensors) -> bool:
    """gloo moves host memory only: device tensors are staged through the CPU (used by the 2-rank
    one-GPU test; the product runs on RCCL and never takes this path)"""
    return dist.get_backend(group) == "gloo" and any(t.is_cuda for t in tensors)


def _all_gather_rows(x: torch.Tensor, group) -> torch.Tensor:
    world = dist.get_world_size(group)
    if _host_staged(group, x):
        return _all_gather_rows(x.cpu(), group).to(x.device)
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out


def _all_reduce_sum_(t: torch.Tensor, group) -> torch.Tensor:
    if _host_staged(group, t):
        c = t.cpu()
        dist.all_reduce(c, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, group=group)
    return t


def _reduce_scatter_rows(full: torch.Tensor, group) -> torch.Tensor:
    world = dist.get_world_size(group)
    n_local = full.shape[0] // world
    out = torch.empty((n_local,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
    full = full.contiguous()
    if dist.get_backend(group) == "gloo":          # gloo has no reduce_scatter: all-reduce + slice
        full = _all_reduce_sum_(full.clone(), group)
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


def _all_to_all_v(recv: torch.Tensor, send: torch.Tensor, recv_splits, send_splits, group):
    """rows of `send` (split by destination rank) -> rows of `recv` (split by source rank)"""
    if _host_staged(group, recv, send):
        r_c = torch.empty(recv.shape, dtype=recv.dtype)
        _all_to_all_v(r_c, send.cpu(), recv_splits, send_splits, group)
        recv.copy_(r_c)
        return
    if dist.get_backend(group) != "gloo":
        dist.all_to_all_single(recv, send, recv_splits, send_splits, group=group)
        return
    # gloo (CPU tests): point-to-point
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    so = [0]
    ro = [0]
    for c in send_splits:
        so.append(so[-1] + c)
    for c in recv_splits:
        ro.append(ro[-1] + c)
    ops = []
    for r in range(world):
        if r == rank:
            recv[ro[r]:ro[r + 1]] = send[so[r]:so[r + 1]]
            continue
        if send_splits[r]:
            ops.append(dist.P2POp(dist.isend, send[so[r]:so[r + 1]].contiguous(), r, group))
        if recv_splits[r]:
            ops.append(dist.P2POp(dist.irecv, recv[ro[r]:ro[r + 1]], r, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


_FORCED_LOGGED = False


def force_exchange() -> bool:
    """PANGNN_FORCE_EXCHANGE=1: test hook that makes a ONE-rank run exchange rows with itself (HaloPlan.__init__)"""
    return os.environ.get("PANGNN_FORCE_EXCHANGE") == "1"


class HaloPlan:
    """Which owned rows every peer needs from this rank, and the local edge list re-indexed into the
    compact table [halo rows of lower ranks | owned rows | halo rows of higher ranks].  Table order is
    global-id order, so an edge list that was sorted by source stays sorted (the decoder's per-source
    partial sums keep working on shards).  Built once per (graph, partition)."""

    def __init__(self, ei_local: torch.Tensor, lo: int, n_local: int, group=None, make_csr=None, bounds=None,
                 emulated_world: Optional[int] = None):
        """`bounds` [world + 1]: first node of every rank's range when the ranges are not equal (balanced_bounds).
        `emulated_world` (bench.py --emulate-rank r --of W): this ONE process stands in for one rank of a W-rank job — see
        `_emulate`."""
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        dev = ei_local.device
        src = ei_local[0]
        hi = lo + n_local
        if emulated_world is not None and emulated_world > 1:
            if world != 1:
                raise ValueError("an emulated rank runs as a one-process job")
            return self._emulate(ei_local, lo, n_local, group, make_csr, bounds, int(emulated_world))
        # Test hook (PANGNN_FORCE_EXCHANGE=1, one rank only): the lower and upper quarter of the node range count as
        # "remote" rows — owned by rank 0 itself — so that every exchange of the N > 1 path (all-to-all-v with split
        # lists, the side-stream decoder exchange, the halo gradient return, the flat all-reduce) executes on the real
        # back end (RCCL on a one-GPU box) and can be compared with the single-GPU model.  Sources are then read from
        # the halo COPIES of those rows, their gradients return through the back-exchange like any boundary row's.
        q = n_local // 4 if (world == 1 and force_exchange()) else 0
        if q:
            global _FORCED_LOGGED
            if not _FORCED_LOGGED:
                _FORCED_LOGGED = True
                import warnings
                warnings.warn("pangnn_amd.dist: PANGNN_FORCE_EXCHANGE=1 — this ONE-rank run treats the outer quarters of its node "
                              "range as remote rows and exchanges them with itself (a test hook; unset it for real runs)")
        lo_own, hi_own = lo + q, hi - q                                    # sources in [lo_own, hi_own) are read in place
        remote = (src < lo_own) | (src >= hi_own)
        need = torch.unique(src[remote])                                   # sorted global ids
        if bounds is None:
            owner = torch.div(need, n_local, rounding_mode="floor")
        else:
            b = torch.as_tensor(list(bounds), dtype=torch.int64, device=dev)
            owner = torch.searchsorted(b, need, right=True) - 1
        need_counts = torch.bincount(owner, minlength=world)
        all_counts = _all_gather_rows(need_counts.view(1, world), group)       # [world(asker), world(owner)]
        counts_host = all_counts.tolist()                                      # ONE read-back for every size below
        mx = max(max(max(row) for row in counts_host), 1)
        # the ids this rank asks of owner r, left-aligned in row r of a [world, mx] table (need is sorted by id, hence by
        # owner): position inside the owner's row = index in `need` - first index of that owner — no per-rank loop
        first = torch.cumsum(need_counts, 0) - need_counts
        padded = torch.full((world, mx), -1, dtype=torch.int64, device=dev)
        if need.numel():
            padded[owner, torch.arange(need.numel(), device=dev) - first[owner]] = need
        everyone = _all_gather_rows(padded.view(1, world, mx), group)          # [asker, owner, mx]
        self.send_splits = [int(counts_host[r][rank]) for r in range(world)]   # rows I send to rank r
        self.recv_splits = [int(c) for c in counts_host[rank]]                 # rows I receive from rank r
        asked = everyone[:, rank, :]                                           # [asker, mx]: what every rank asks of me
        self.send_idx = (asked[asked >= 0] - lo).contiguous()                  # row-major = by asker, ids ascending
        assert self.send_idx.numel() == 0 or (int(self.send_idx.min()) >= 0 and int(self.send_idx.max()) < n_local)
        self.n_local, self.n_halo = n_local, int(need.numel())
        self.n_table = n_local + self.n_halo
        self.n_low = int((need < lo_own).sum())                   # halo rows owned by lower ranks
        pos = torch.searchsorted(need, src)                       # rank of a remote source among the needed ids
        new_src = torch.where(remote, torch.where(src < lo_own, pos, pos + n_local), src - lo + self.n_low)
        self.edge_index = torch.stack([new_src, ei_local[1]]).contiguous()
        self.group = group
        # A source-sorted list is [sources of lower ranks | own sources | sources of higher ranks]: the middle range
        # needs no exchanged row, so the decoder can run on it while the halo rows travel (decoder_loss_overlapped)
        sorted_here = bool((new_src[1:] >= new_src[:-1]).all()) if new_src.numel() > 1 else True
        # Whether the collectives run at all, and which decoder path issues them, must be the SAME decision on every
        # rank (a rank without boundary edges that skipped its all-to-all would leave the others' sequence numbers
        # behind — and hang a back end that implements all-to-all as a true collective): both flags are global.
        flag = torch.tensor([[0 if sorted_here else 1]], dtype=torch.int64, device=dev)
        self.sorted_by_src = int(_all_gather_rows(flag, group).sum()) == 0      # MIN over ranks of "sorted here"
        self.any_exchange = any(c > 0 for row in counts_host for c in row)      # anyone needs any row of anyone (host copy: no second read-back)
        self.e_lo = int((new_src < self.n_low).sum())
        self.e_hi = int((new_src < self.n_low + n_local).sum())
        self._split_cache = {}
        # fixed-order accumulation of returned halo gradients into the owner's rows
        self.back_csr = make_csr(self.send_idx, n_local) if (make_csr is not None and self.send_idx.numel()) else None


    def _emulate(self, ei_local, lo, n_local, group, make_csr, bounds, emulated_world):
        """The plan of ONE rank of an `emulated_world`-rank job, built by a one-process job (a one-GPU box: the only per-rank
        evidence obtainable without a node).  The remote-source set is the shard's TRUE one (every source outside [lo, hi)),
        the table layout, the re-indexed edge list, the own-source / halo-source split of the decoder and the back
        accumulation are what the real rank would build; what cannot exist is the peers: the exchange is served by a
        SELF-exchange of the same row counts over the process group (rows `k mod n_local` of the own block stand in for halo
        row k, and as many rows are sent as received — the adjacent-genome halo is symmetric), so kernels, launches, bytes
        through RCCL and stream structure are the real rank's while the VALUES of the halo rows are not.  For timing only."""
        dev = ei_local.device
        src = ei_local[0]
        hi = lo + n_local
        remote = (src < lo) | (src >= hi)
        need = torch.unique(src[remote])
        if bounds is None:
            owner = torch.div(need, n_local, rounding_mode="floor")
        else:
            owner = torch.searchsorted(torch.as_tensor(list(bounds), dtype=torch.int64, device=dev), need, right=True) - 1
        self.peer_counts = torch.bincount(owner, minlength=emulated_world).tolist()       # halo rows asked of every rank
        self.emulated_world = emulated_world
        n_halo = int(need.numel())
        self.send_splits, self.recv_splits = [n_halo], [n_halo]
        self.send_idx = (torch.arange(n_halo, device=dev, dtype=torch.int64) % max(n_local, 1)).contiguous()
        self.n_local, self.n_halo = n_local, n_halo
        self.n_table = n_local + n_halo
        self.n_low = int((need < lo).sum())
        pos = torch.searchsorted(need, src)
        new_src = torch.where(remote, torch.where(src < lo, pos, pos + n_local), src - lo + self.n_low)
        self.edge_index = torch.stack([new_src, ei_local[1]]).contiguous()
        self.group = group
        self.sorted_by_src = bool((new_src[1:] >= new_src[:-1]).all()) if new_src.numel() > 1 else True
        self.any_exchange = n_halo > 0
        self.e_lo = int((new_src < self.n_low).sum())
        self.e_hi = int((new_src < self.n_low + n_local).sum())
        self._split_cache = {}
        self.back_csr = make_csr(self.send_idx, n_local) if (make_csr is not None and n_halo) else None

    def split_edges(self):
        """(own-source edges, halo-source edges) of the re-indexed list, each still sorted by source"""
        ei = self.edge_index
        loc = ei[:, self.e_lo:self.e_hi].contiguous()
        halo = torch.cat([ei[:, : self.e_lo], ei[:, self.e_hi:]], dim=1).contiguous()
        return loc, halo

    def split_edge_values(self, t: Optional[torch.Tensor]):
        """a per-edge tensor of the shard (labels, skip feature) in the two ranges; cached on the tensor's identity"""
        if t is None:
            return None, None
        key = (t.data_ptr(), t._version, tuple(t.shape))
        hit = self._split_cache.get(key)
        if hit is None:
            if len(self._split_cache) >= 4:
                self._split_cache.pop(next(iter(self._split_cache)))
            hit = self._split_cache[key] = (t, t[self.e_lo:self.e_hi].contiguous(),
                                            torch.cat([t[: self.e_lo], t[self.e_hi:]]).contiguous())
        return hit[1], hit[2]

    def merge_edge_values(self, loc: torch.Tensor, halo: torch.Tensor) -> torch.Tensor:
        """inverse of split_edge_values: back to the shard's edge order"""
        return torch.cat([halo[: self.e_lo], loc, halo[self.e_lo:]])


class HaloGather(torch.autograd.Function):
    """[n_local, F] -> table [n_low + n_local + n_high, F]: the referenced rows of lower ranks, the own rows,
    the referenced rows of higher ranks."""

    @staticmethod
    def forward(ctx, x, plan: HaloPlan, accumulate_back):
        ctx.plan, ctx.acc = plan, accumulate_back
        x = x.contiguous()
        if not plan.any_exchange:
            return x.view_as(x)                                 # no rank exchanges anything (one rank): the table IS the block
        send = x.index_select(0, plan.send_idx)
        recv = torch.empty((plan.n_halo,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        _all_to_all_v(recv, send, plan.recv_splits, plan.send_splits, plan.group)
        return torch.cat([recv[: plan.n_low], x, recv[plan.n_low:]], dim=0)    # global-id order

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        g = g.contiguous()
        if not plan.any_exchange:
            return g, None, None
        g_halo = torch.cat([g[: plan.n_low], g[plan.n_low + plan.n_local:]], dim=0)
        back = torch.empty((plan.send_idx.numel(),) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        _all_to_all_v(back, g_halo, plan.send_splits, plan.recv_splits, plan.group)
        # a bf16-stored table (config 5) exchanges its gradient rows as bf16 too; the sums are fp32
        g_local = g[plan.n_low: plan.n_low + plan.n_local].to(torch.float32, copy=True)
        if back.shape[0]:
            ctx.acc(g_local, back.float(), plan)
        return g_local, None, None


class _BandTinyHalo(torch.autograd.Function):
    """A_hat x + bias for the positional-neighbour graph of a shard (dataset.py:356-361: a band of half-width k over the WHOLE
    node range, unit weights): the edges whose source this rank owns are a band over its own rows — propagated by the band
    kernel straight from the [n_local, F] block, no table — and the few edges that cross the partition boundary (at most k
    rows on either side) are added from the exchanged halo rows.  Against HaloGather + the generic propagate this saves the
    concatenation of the block into a table, the copy of its gradient back out, the accumulation pass over it and the separate
    column sum of the bias gradient (one step of rank 0 of 2, profiles/r05z_*: 57 + 46 + 46 + 43 us of a 6.4 ms step)."""

    @staticmethod
    def forward(ctx, x, bias, plan, dis, k, h_src, h_dst, h_val):
        ctx.plan, ctx.k, ctx.has_bias = plan, int(k), bias is not None
        ctx.save_for_backward(dis, h_src, h_dst, h_val)
        out = PF._band_call(x, None if bias is None else PF._f32c(bias), dis, int(k), False)[0]
        if plan.any_exchange:
            send = x.index_select(0, plan.send_idx)
            recv = x.new_empty((plan.n_halo, x.shape[1]))
            _all_to_all_v(recv, send, plan.recv_splits, plan.send_splits, plan.group)
            if h_src.numel():
                out.index_add_(0, h_dst, recv.float().index_select(0, h_src) * h_val.unsqueeze(1))
        return out

    @staticmethod
    def backward(ctx, g):
        dis, h_src, h_dst, h_val = ctx.saved_tensors
        plan = ctx.plan
        want_b = ctx.has_bias and ctx.needs_input_grad[1]
        g = PF._f32c(g)
        gx, gb = PF._band_call(g, None, dis, ctx.k, want_b)           # the band is symmetric: the same kernel, + column sums
        if plan.any_exchange:
            gh = g.new_zeros((plan.n_halo, g.shape[1]))
            if h_src.numel():
                gh.index_add_(0, h_src, g.index_select(0, h_dst) * h_val.unsqueeze(1))
            back = g.new_empty((plan.send_idx.numel(), g.shape[1]))
            _all_to_all_v(back, gh, plan.send_splits, plan.recv_splits, plan.group)
            if back.shape[0]:
                gx.index_add_(0, plan.send_idx, back)
        return gx, gb, None, None, None, None, None, None


_SIDE_STREAMS = {}


def _side_stream(device):
    """one extra stream per device for the exchanges that run under compute (None on the CPU)"""
    if device.type != "cuda":
        return None
    s = _SIDE_STREAMS.get(device)
    if s is None:
        s = _SIDE_STREAMS[device] = torch.cuda.Stream(device)
    return s


class _Side:
    """`with _Side(device, *tensors)`: run the block on the side stream, after everything enqueued so far on the
    current one; the listed tensors are marked as used there (caching-allocator lifetime).  `.done()` afterwards makes
    the current stream wait for the block.  On the CPU both are no-ops and the block simply runs in order."""

    def __init__(self, device, *tensors):
        self.side = _side_stream(device)
        self.tensors = tensors
        self.event = None

    def __enter__(self):
        if self.side is not None:
            self.main = torch.cuda.current_stream(self.side.device)
            self.side.wait_stream(self.main)
            for t in self.tensors:
                if t is not None:
                    t.record_stream(self.side)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def keep(self, *tensors):
        """tensors allocated inside the block that the main stream will read"""
        if self.side is not None:
            for t in tensors:
                t.record_stream(self.main)

    def __exit__(self, *exc):
        if self.side is not None:
            self.event = self.side.record_event()
            self.ctx.__exit__(*exc)
        return False

    def done(self):
        if self.event is not None:
            torch.cuda.current_stream(self.side.device).wait_event(self.event)


class _OverlappedDecoderLoss(torch.autograd.Function):
    """The training decoder of one shard with both halo exchanges hidden under compute.

      side stream:  rows of P to the ranks that read them  ............  halo rows' dL/dP back to their owners
      main stream:  decoder over the own-source edges | decoder over the halo-source edges (S | dL/dP | T) | add

    `ops.decoder_train` is the one-pass training decoder (loss, logits and every gradient in forward), so like
    functional._DecoderLoss this function finishes all gradients in forward() and backward() only hands them out.
    Inputs are this rank's blocks p_local, q_local [n_local, D]; the table [halo | own | halo] exists only inside.
    `q_local=None`: `p_local` is the JOINT block P | Q [n_local, 2 D] of one node-level product (AlternateGCN's layout) and
    the gradient comes back as one [n_local, 2 D] matrix — one dense layer forward and backward instead of two plus the
    addition of their two dL/dz."""

    @staticmethod
    def forward(ctx, p_local, q_local, ops, plan: HaloPlan, st_loc, st_halo, extra, cvec, w2, b2, w3, b3, y,
                pos_weight, denom):
        dev = p_local.device
        joint = q_local is None
        if joint:
            pq = p_local if p_local.stride(1) == 1 else p_local.contiguous()
            dj = pq.shape[1] // 2
            p_local, q_local = pq[:, :dj], pq[:, dj:]
            # [dL/dP | dL/dQ] of the own rows — what the dense layer's backward reads: the own-source range's by-source sums are
            # written for the own rows only, straight into its left half; the halo-source range's for the halo rows only, into the
            # compact buffer that travels back to their owners (no sum over rows a range cannot touch, no copy, no concatenation)
            gpq = torch.empty((plan.n_local, pq.shape[1]), dtype=torch.float32, device=dev)
            g_halo32 = torch.empty((plan.n_halo, dj), dtype=torch.float32, device=dev)
            merged = torch.empty(plan.edge_index.shape[1], dtype=torch.float32, device=dev)      # logits in the shard's edge order
        else:
            p_local = p_local.contiguous()
        n_low, n_loc, d = plan.n_low, plan.n_local, p_local.shape[1]
        table = p_local.new_empty((plan.n_table, d))
        with _Side(dev, p_local, table) as fwd:
            send = p_local.index_select(0, plan.send_idx)
            recv = p_local.new_empty((plan.n_halo, d))
            _all_to_all_v(recv, send, plan.recv_splits, plan.send_splits, plan.group)
            table[:n_low] = recv[:n_low]
            table[n_low + n_loc:] = recv[n_low:]
        table[n_low:n_low + n_loc] = p_local
        ex_l, ex_h = plan.split_edge_values(extra)
        y_l, y_h = plan.split_edge_values(y)
        # own-source edges: read rows n_low .. n_low + n_local of the table only
        r_loc = ops.decoder_train(table, q_local, st_loc, ex_l, cvec, w2, b2, w3, b3, y_l, pos_weight, denom,
                                  **({"out_q": gpq[:, d:], "p_windows": [(n_low, n_low + n_loc, gpq[:, :d])],
                                      "out_logits": merged[plan.e_lo:plan.e_hi]} if joint else {}))
        fwd.done()
        box = {}

        def send_back(gp_table):
            with _Side(dev, *(gp_table if joint else [gp_table])) as bwd:
                g_halo = g_halo32.to(p_local.dtype) if joint else \
                    torch.cat([gp_table[:n_low], gp_table[n_low + n_loc:]], dim=0).to(p_local.dtype)
                back = p_local.new_empty((plan.send_idx.numel(), d))    # travels in the table's storage type
                _all_to_all_v(back, g_halo, plan.send_splits, plan.recv_splits, plan.group)
                bwd.keep(back)
            box["bwd"], box["back"] = bwd, back

        # the halo-source range has the same targets: its by-target pass adds into the own-source range's dL/dQ
        r_halo = ops.decoder_train(table, q_local, st_halo, ex_h, cvec, w2, b2, w3, b3, y_h, pos_weight, denom,
                                   after_p=send_back, out_q=r_loc[3], accumulate_q=True,
                                   **({"p_windows": [(0, n_low, g_halo32[:n_low]), (n_low + n_loc, plan.n_table, g_halo32[n_low:])]}
                                      if joint else {}))
        loss_l, logit_l, gp_l, gq = r_loc[:4]
        loss_h, logit_h = r_halo[:2]
        gp_local = gpq[:, :d] if joint else gp_l[n_low:n_low + n_loc]     # rows of halo sources are zero in gp_l
        box["bwd"].done()
        if box["back"].shape[0]:
            ops.accumulate_back(gp_local, box["back"].float(), plan)
        a_s, b_s = r_loc[4:], r_halo[4:]                          # g_cv, g_w2, g_b2, g_w3, g_b3 of the two ranges
        live = [i for i, a in enumerate(a_s) if a is not None]
        small = [None] * len(a_s)
        for i, t in zip(live, torch._foreach_add([a_s[i] for i in live], [b_s[i] for i in live])):   # one launch
            small[i] = t
        if joint:                                   # the own-source range wrote its logits in place: the halo part is copied in
            merged[:plan.e_lo].copy_(logit_h[:plan.e_lo])
            merged[plan.e_hi:].copy_(logit_h[plan.e_lo:])
            logits = merged
        else:
            logits = plan.merge_edge_values(logit_l, logit_h)
        ctx.has_cv, ctx.joint = small[0] is not None, joint
        if joint:
            gp_local, gq = gpq, gpq.new_empty(0)
        ctx.save_for_backward(gp_local, gq, small[0] if small[0] is not None else gq.new_empty(0), *small[1:])
        ctx.mark_non_differentiable(logits)
        return (loss_l + loss_h).view(()), logits

    @staticmethod
    def backward(ctx, go, _go_logits):
        gp, gq, g_cv, g_w2, g_b2, g_w3, g_b3 = ctx.saved_tensors
        k = (lambda t: t) if PF.is_unit_grad(go) else (lambda t: t * go)
        return (k(gp), None if ctx.joint else k(gq), None, None, None, None, None, k(g_cv) if ctx.has_cv else None, k(g_w2),
                k(g_b2), k(g_w3), k(g_b3), None, None, None)


# --------------------------------------------------------------------------------------
# partition
# --------------------------------------------------------------------------------------
def balanced_bounds(genes: int, genomes: int, world: int):
    """Node ranges [bounds[r], bounds[r + 1]) with equal expected numbers of in-edges for a simulated pan-genome
    (genome-major node ids): after remove_trivial_cases only adjacent genomes are joined and every adjacent pair
    carries the same expected number of edges in both directions, so a node of an end genome receives half the edges
    of any other node.  Computable on every rank without communication (rank-local generation needs it before a
    single edge exists).  Equal node ranges leave the two end ranks ~20 % short of work at 8 ranks x 20 genomes."""
    genes, genomes, world = int(genes), int(genomes), int(world)
    w = [1 if (g == 0 or g == genomes - 1) else 2 for g in range(genomes)] if genomes > 1 else [1]
    total = genes * sum(w)
    bounds = [0]
    for r in range(1, world):
        target = total * r // world               # cumulative weight before the boundary
        g, before = 0, 0
        while g < genomes - 1 and before + genes * w[g] <= target:
            before += genes * w[g]
            g += 1
        bounds.append(min(g * genes + (target - before) // w[g], genes * genomes))
    bounds.append(genes * genomes)
    return bounds


def partition_graph(g, rank: int, world: int, bounds=None):
    """Slice a whole graph (x, edge_index, edge_attr, y, neighbour_edge_index[, union_edge_index]) into
    rank's shard.  Index tensors keep GLOBAL source ids and get LOCAL target ids.  `bounds` [world + 1]: unequal
    node ranges (balanced_bounds); default = equal ranges of ceil(N / world) nodes."""
    n = int(g.x.shape[0])
    if bounds is None:
        n_local = (n + world - 1) // world
        lo, hi = rank * n_local, min((rank + 1) * n_local, n)
        n_pad = n_local * world
    else:
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        n_local = hi - lo
        n_pad = None                     # no all-gather exchange over unequal blocks
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
        n_local=n_local, n_pad=n_pad, n_global=n, lo=lo, hi=hi, rank=rank, world=world,
        bounds=None if bounds is None else [int(b) for b in bounds],
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

    def embed_propagate(self, x_tab, w, b, st, norm, tag=None):
        return PF.embed_propagate(x_tab, w, b, st, norm, tag)

    def embed_conv_in(self, x_tab, w, b, w_in, b_in, st, norm):
        return PF.embed_conv_in(x_tab, w, b, w_in, b_in, st, norm)

    def embed_conv_in_linear(self, x_tab, w, b, w_in, b_in, w_out, bias_out, st, norm):
        return PF.embed_conv_in_linear(x_tab, w, b, w_in, b_in, w_out, bias_out, st, norm)

    def decoder(self, p_full, q_local, st, extra, cvec, w2, b2, w3, b3):
        return PF.decoder_mlp(p_full, q_local, st, extra, cvec, w2, b2, w3, b3)

    def make_back_csr(self, send_idx, n_local):
        """CSR over owned rows of the positions in the returned-gradient buffer (stable => fixed order)"""
        m = send_idx.numel()
        ei = torch.stack([torch.arange(m, device=send_idx.device), send_idx]).contiguous()
        return EdgeStructure(ei, n_local, max(m, 1)).by_dst

    def accumulate_back(self, g_local, back, plan):
        """g_local[send_idx[k]] += back[k], summed per row in buffer order (segment sum, no atomics)"""
        if back.dim() == 1:
            g_local.index_add_(0, plan.send_idx, back)        # only deg^-1/2 (no grad path): never hit in training
            return
        PF.segment_sum_rows(plan.back_csr, back, 0, back.shape[1], plan.n_local, out=g_local, accumulate=True)

    def decoder_loss(self, p_full, q_local, st, extra, cvec, w2, b2, w3, b3, y, pos_weight, denom):
        return PF.decoder_loss(p_full, q_local, st, extra, cvec, w2, b2, w3, b3, y, pos_weight, denom)

    def decoder_train(self, table, q_local, st, extra, cvec, w2, b2, w3, b3, y, pos_weight, denom, after_p=None, out_q=None,
                      accumulate_q=False, out_p=None, p_windows=None, out_logits=None):
        """one-pass training decoder on (a range of) a shard: (loss, logits, dL/dtable, dL/dq, g_cvec, g_w2, g_b2, g_w3,
        g_b3), all finished; `after_p(dL/dtable)` runs before the by-target pass is enqueued; `out_q`: where dL/dq is written
        (a column window of a wider matrix is fine), or ADDED to it with `accumulate_q`"""
        f = PF._f32c
        pw = None if pos_weight is None else f(pos_weight).reshape(-1)
        loss, logits, gp, gq, g_cv, g_w2, g_b2, g_w3, g_b3 = PF._decoder_train16(
            PF._rows_dec(table), PF._rows_dec(q_local), st, None if extra is None else f(extra),
            None if cvec is None else f(cvec), f(w2), f(b2), f(w3), f(b3), y=f(y), pw=pw, denom=denom, after_p=after_p,
            out_q=out_q, accumulate_q=accumulate_q, out_p=out_p, p_windows=p_windows, out_logits=out_logits)
        return loss.view(()), logits, gp, gq, g_cv, g_w2, g_b2, g_w3, g_b3

    def linear(self, x, w, b, in_act: int = 0, out_dtype=None):
        return PF.linear(x, w, b, in_act, out_dtype)

    def bce_sum_over(self, logits, labels, pos_weight, denom):
        """sum_i BCEWithLogits_i / denom (this rank's share of the global mean)"""
        return PF.bce_with_logits(logits, labels, pos_weight, denom=denom)


class DistAlternateGCN(AlternateGCN):
    """AlternateGCN evaluated on a destination-partitioned shard.  Same parameters / state_dict as AlternateGCN (and the
    reference, src/gnn.py:93-116) EXCEPT under `categorical_nodes=True`, where `embedding.weight` holds only the rows of
    the nodes this rank owns: `full_state_dict()` / `load_full_state_dict()` convert to and from the reference layout
    (`embedding.weight [N, D]`), `torch.save(model.state_dict())` of one rank alone is NOT a checkpoint then."""

    def __init__(self, device=None, dims=(64, 128), part=None, group=None, ops=None, exchange="halo",
                 categorical_nodes: bool = False, **kw):
        """`categorical_nodes` (config 5, `--categorical_node`): one embedding row per node, SHARDED like the nodes — a
        rank holds `Embedding(n_local, D)` for the rows it owns (needs `part`), the halo rows travel with the first
        layer's exchange, and `embedding.weight` is a rank-local parameter that the gradient all-reduce skips (a
        replicated [N, D] table would cost a 2.5 GB all-reduce per step at N = 1e7)."""
        super().__init__(device, None, False, dims=dims, **kw)
        self.sharded_embedding = bool(categorical_nodes)
        if self.sharded_embedding:
            if part is None:
                raise ValueError("categorical_nodes=True needs part= (the shard whose rows the embedding covers)")
            self.embedding = torch.nn.Embedding(int(part.n_local), dims[0])
            # nn.Embedding draws N(0, 1) rows from the global generator, which every rank seeds alike: node i of every
            # shard would start from the same vector.  Re-draw from a generator keyed on (seed, first owned node).
            gen = torch.Generator().manual_seed((int(torch.initial_seed()) * 1000003 + int(part.lo) + 1) % (2 ** 63 - 1))
            with torch.no_grad():
                self.embedding.weight.copy_(torch.randn(self.embedding.weight.shape, generator=gen))
            self._rows = (int(part.lo), int(part.hi), int(part.n_global))
            self.categorical_nodes = True
            if device is not None:
                self.embedding.to(device)
        self.group = group
        self.ops = ops or HipOps()
        if exchange not in ("halo", "allgather"):
            raise ValueError("exchange must be 'halo' or 'allgather'")
        self.exchange = exchange
        self.overlap = os.environ.get("PANGNN_DIST_OVERLAP", "1") != "0"
        if self.flags.decoder != "mlp":
            raise NotImplementedError("partitioned mode implements the mlp decoder")
        if dims[0] != 64 and isinstance(self.ops, HipOps):
            raise NotImplementedError("partitioned HIP decoder is built for node_dim 64")

    # structures / norms are per shard tensor and cached on the shard object
    def _plan(self, shard, name):
        cache = shard.__dict__.setdefault("_dist_plans", {})
        if name not in cache:
            ei = {"sim": shard.edge_index, "nb": getattr(shard, "neighbour_edge_index", None),
                  "union": getattr(shard, "union_edge_index", None)}[name]
            cache[name] = HaloPlan(ei, shard.lo, shard.n_local, self.group,
                                   getattr(self.ops, "make_back_csr", None), getattr(shard, "bounds", None),
                                   getattr(shard, "emulated_world", None))
        return cache[name]

    def _st(self, shard, name):
        cache = shard.__dict__.setdefault("_dist_structs", {})
        key = (name, self.exchange)
        if key not in cache:
            if self.exchange == "halo":
                plan = self._plan(shard, name)
                cache[key] = self.ops.structure(plan.edge_index, shard.n_local, plan.n_table)
            else:
                if shard.n_pad is None:
                    raise ValueError("exchange='allgather' needs equal node ranges (partition without bounds=)")
                ei = {"sim": shard.edge_index, "nb": getattr(shard, "neighbour_edge_index", None),
                      "union": getattr(shard, "union_edge_index", None)}[name]
                cache[key] = self.ops.structure(ei, shard.n_local, shard.n_pad)
        return cache[key]

    def _st_split(self, shard):
        """structures of the own-source and halo-source ranges of the shard's sim edges (overlapped decoder)"""
        cache = shard.__dict__.setdefault("_dist_structs", {})
        key = ("sim", "split")
        if key not in cache:
            plan = self._plan(shard, "sim")
            loc, halo = plan.split_edges()
            cache[key] = (self.ops.structure(loc, shard.n_local, plan.n_table),
                          self.ops.structure(halo, shard.n_local, plan.n_table))
        return cache[key]

    def _overlap_ok(self, shard) -> bool:
        """the exchange can hide under the decoder: halo exchange, a back end with the one-pass decoder, a source-
        sorted shard and something to exchange (PANGNN_DIST_OVERLAP=0 switches it off)"""
        if self.exchange != "halo" or not self.overlap or not hasattr(self.ops, "decoder_train"):
            return False
        if isinstance(self.ops, HipOps) and PF.DECODER_PRECISION != 1:
            return False
        plan = self._plan(shard, "sim")
        return plan.sorted_by_src and plan.any_exchange           # both agreed over all ranks (HaloPlan.__init__)

    def _table(self, x_local, shard, name):
        """rows of every node this rank's `name` edges read: [own | halo] or the all-gathered [N_pad]"""
        if self.exchange == "halo":
            return HaloGather.apply(x_local, self._plan(shard, name), self.ops.accumulate_back)
        return AllGatherRows.apply(x_local, self.group)

    def _norm(self, shard, name, weight, wkey):
        """normalisation of (graph `name`, weight tensor): cached on the shard, keyed on the weight tensor's identity
        and version; the entry keeps the tensor alive so that its address cannot be handed to another one"""
        cache = shard.__dict__.setdefault("_dist_norms", {})
        key = (name, wkey, self.exchange)
        ident = None if weight is None else (weight.data_ptr(), weight._version, tuple(weight.shape))
        hit = cache.get(key)
        if hit is None or hit[0] != ident:
            gather = lambda d: self._table(d.view(-1, 1), shard, name).view(-1).contiguous()   # noqa: E731
            hit = cache[key] = (ident, self.ops.norm(self._st(shard, name), weight, gather), weight)
        return hit[1]

    def _linear(self, x, w, b, in_act: int = 0, out_dtype=None):
        if out_dtype is None:
            return self.ops.linear(x, w, b, in_act)
        return self.ops.linear(x, w, b, in_act, out_dtype)

    def _tiny_halo_band(self, shard, name, weight, wkey):
        """(plan, deg^-1/2 of the own rows, k, halo-source edges as (compact halo row, own target row, norm)) when the shard's
        `name` graph is the positional-neighbour band with unit weights, else None.  Decided once per shard (host read-backs),
        the same way on every rank of a genome-major partition (each rank sees the band or none does: the decision uses only
        the shard's own edge list, and a rank whose list is not the band keeps the generic path — both paths exchange the same
        rows with the same split lists)."""
        cache = shard.__dict__.setdefault("_dist_band", {})
        if name not in cache:
            cache[name] = None
            plan = self._plan(shard, name) if (self.exchange == "halo" and weight is None and isinstance(self.ops, HipOps)
                                               and os.environ.get("PANGNN_DIST_BAND", "1") != "0") else None
            if plan is not None and plan.n_local > 16 and plan.edge_index.shape[1] > 0:
                src, dst = plan.edge_index[0], plan.edge_index[1]
                n_low, n_loc = plan.n_low, plan.n_local
                own = (src >= n_low) & (src < n_low + n_loc)
                diff = (src - n_low - dst)[own]
                k = int(diff.abs().max()) if diff.numel() else 0
                n_own = int(own.sum())
                if 1 <= k <= 8 and k < n_loc and n_own == n_loc * (2 * k + 1) - k * (k + 1) \
                        and int(torch.unique(dst[own] * (2 * k + 1) + diff + k).numel()) == n_own \
                        and plan.n_halo <= 2 * k:
                    norm = self._norm(shard, name, weight, wkey)
                    hs = src[~own]
                    h_src = torch.where(hs < n_low, hs, hs - n_loc).contiguous()       # row of the compact [low | high] halo block
                    cache[name] = (plan, norm.deg_inv_sqrt, k, h_src, dst[~own].contiguous(), norm.orig[~own].contiguous())
        return cache[name]

    def _propagate_rows(self, rows_local, bias, shard, name, weight, wkey, tag):
        """A_hat rows + bias of this rank's targets from the [n_local, F] block of its own rows: the band kernel + the boundary
        rows where the graph is the positional-neighbour band (_BandTinyHalo), halo table + generic propagate otherwise"""
        tiny = self._tiny_halo_band(shard, name, weight, wkey) if rows_local.shape[1] in (64, 128) else None
        if tiny is not None and rows_local.dtype == torch.float32:
            return _BandTinyHalo.apply(rows_local, bias, *tiny)
        st, norm = self._st(shard, name), self._norm(shard, name, weight, wkey)
        return self.ops.propagate(self._table(rows_local, shard, name), bias, st, norm, tag)

    def _conv(self, conv, h_local, shard, name, weight, wkey, tag, in_elu: bool = False, dense_done: bool = False):
        """`in_elu`: h_local is the pre-activation of the deferred ELU (see AlternateGCN._encode_pre);
        `dense_done`: h_local already is conv.lin(...) of the layer's input (GCNConv.forward, dense_done)"""
        if dense_done:
            return self._propagate_rows(h_local, conv.bias, shard, name, weight, wkey, tag)
        if in_elu and conv.in_channels < conv.out_channels:
            h_local, in_elu = F.elu(h_local), False
        if conv.in_channels < conv.out_channels:
            # propagate (and exchange) on the narrower side: half the all-gather bytes for 64 -> 128
            agg = self._propagate_rows(h_local, None, shard, name, weight, wkey, tag)
            return self._linear(agg, conv.lin.weight, conv.bias)
        xw = self._linear(h_local, conv.lin.weight, None, 1 if in_elu else 0)
        return self._propagate_rows(xw, conv.bias, shard, name, weight, wkey, tag)

    def _xtab(self, shard, name):
        """the scalar feature of every row the shard's `name` edges read (own + halo), exchanged once and cached"""
        cache = shard.__dict__.setdefault("_dist_xtab", {})
        key = (name, self.exchange)
        if key not in cache:
            with torch.no_grad():
                cache[key] = self._table(shard.x.float().view(-1, 1), shard, name).view(-1).contiguous()
        return cache[key]

    def _embed_conv_in_then_dense(self, shard, name, weight, w_out, bias_out):
        """AlternateGCN._embed_conv_in_then_dense on a shard: the first layer's rows of the OWN nodes generated inside the
        dense layer that consumes them (node-level: no exchange involved); None where it does not apply"""
        conv = self.conv_in
        if self.sharded_embedding or not self.fuse_first_dense or not self._fold_elu() or not self.fuse_embedding \
                or self.fuse_embedding == "propagate" or not hasattr(self.ops, "embed_conv_in_linear"):
            return None
        if w_out.shape[1] != conv.out_channels or not PF.embed_linear_supported(conv.out_channels, w_out.shape[0]) \
                or PF.autocast_bf16(shard.x):
            return None
        st, norm = self._st(shard, name), self._norm(shard, name, weight, "w")
        return self.ops.embed_conv_in_linear(self._xtab(shard, name), self.embedding.weight, self.embedding.bias,
                                             conv.lin.weight, conv.bias, w_out, bias_out, st, norm)

    def _embed_conv_in(self, shard, name, weight):
        """conv_in(embedding(x)): x is constant, so the scalar features of the halo rows are exchanged once
        (cached on the shard) and the first layer runs without any per-step exchange, forward or backward."""
        conv = self.conv_in
        if self.sharded_embedding:
            # rows of the owned nodes are the parameter itself; the exchange of the first layer carries the halo rows
            return self._conv(conv, self.embedding.weight, shard, name, weight, "w", name)
        rank2 = self.fuse_embedding and self.fuse_embedding != "propagate" and hasattr(self.ops, "embed_conv_in")
        if rank2 or (conv.in_channels < conv.out_channels and self.fuse_embedding and hasattr(self.ops, "embed_propagate")):
            xtab = self._xtab(shard, name)
            st, norm = self._st(shard, name), self._norm(shard, name, weight, "w")
            if rank2:      # the whole layer by linearity (functional._EmbedConvIn): r = A_hat x, s = A_hat 1 of the OWN rows
                return self.ops.embed_conv_in(xtab, self.embedding.weight, self.embedding.bias, conv.lin.weight,
                                              conv.bias, st, norm)
            agg = self.ops.embed_propagate(xtab, self.embedding.weight, self.embedding.bias, st, norm, name)
            return self._linear(agg, conv.lin.weight, conv.bias)
        h = shard.x.float().view(-1, 1) * self.embedding.weight.view(1, -1) + self.embedding.bias
        return self._conv(conv, h, shard, name, weight, "w", name)

    def _encode_pre(self, shard):
        """as AlternateGCN._encode_pre, on a shard: (z or its pre-activation, pending)"""
        fl, act, fold = self.flags, self.activation_fct, self._fold_elu()
        pre = (lambda h: h) if fold else act
        if fl.union_edge_weights:
            w = shard.union_edge_attr
            h = self._embed_conv_in(shard, "union", w)
            for _ in range(max(fl.neighbours - 2, 1)):
                h = self._conv(self.conv_hidden, pre(h), shard, "union", w, "w", "union", in_elu=fold)
            h = self._conv(self.conv_out, pre(h), shard, "union", None, "1", "union", in_elu=fold)
        elif fl.base_model:
            h = self._embed_conv_in_then_dense(shard, "sim", shard.edge_attr, self.linear_out.weight, self.linear_out.bias)
            if h is None:
                h = self._embed_conv_in(shard, "sim", shard.edge_attr)
                h = self._linear(pre(h), self.linear_out.weight, self.linear_out.bias, 1 if fold else 0)
        else:
            out = self.conv_out
            y = self._embed_conv_in_then_dense(shard, "sim", shard.edge_attr, out.lin.weight, None) \
                if out.in_channels >= out.out_channels else None
            if y is not None:
                h = self._conv(out, y, shard, "nb", None, "1", "nb", dense_done=True)
            else:
                h = self._embed_conv_in(shard, "sim", shard.edge_attr)
                h = self._conv(out, pre(h), shard, "nb", None, "1", "nb", in_elu=fold)
        return h, True

    def encode(self, shard):
        h, pending = self._encode_pre(shard)
        return self.activation_fct(h) if pending else h

    def _dec_in(self, z, shard, in_act: int = 0, gather: bool = True):
        """gather=False: p stays this rank's block (the overlapped decoder exchanges the halo rows itself)"""
        fl = self.flags
        d = z.shape[1]
        lin0 = self.mlp[0]
        w = lin0.weight
        # bf16 mixed precision (config 5): P and Q are stored as bfloat16 — the halo exchange of P moves half the bytes
        pq_dtype = PF.autocast_rows_dtype(z) if (isinstance(self.ops, HipOps) and d == 64 and PF.DECODER_PRECISION == 1) else None
        p = self._linear(z, w[:, :d].contiguous(), None, in_act, pq_dtype)
        q = self._linear(z, w[:, d:2 * d].contiguous(), lin0.bias, in_act, pq_dtype)
        p_full = self._table(p, shard, "sim") if gather else p
        extra = shard.edge_attr if fl.skip_connections else None
        cvec = w[:, 2 * d].contiguous() if fl.skip_connections else None
        return p_full, q, extra, cvec

    def _dec_in_joint(self, z, shard, in_act: int = 0):
        """(P | Q [n_local, 2 D] of this rank's rows, extra, cvec): the re-associated first decoder layer as one product"""
        fl = self.flags
        d = z.shape[1]
        lin0 = self.mlp[0]
        w_pq, b_pq, cvec = PF.pq_operands(lin0.weight, lin0.bias, d, bool(fl.skip_connections))
        pq_dtype = PF.autocast_rows_dtype(z) if PF.DECODER_PRECISION == 1 else None
        pq = self._linear(z, w_pq, b_pq, in_act, pq_dtype)
        return pq, (shard.edge_attr if fl.skip_connections else None), cvec

    def decode_mlp(self, z, shard):
        p_full, q, extra, cvec = self._dec_in(z, shard)
        return self.ops.decoder(p_full, q, self._st(shard, "sim"), extra, cvec, self.mlp[2].weight,
                                self.mlp[2].bias, self.mlp[4].weight.view(-1), self.mlp[4].bias)

    def forward(self, shard):
        return self.decode_mlp(self.encode(shard), shard)

    def loss_and_logits(self, shard, labels, pos_weight=None):
        """global-mean BCE loss share of this rank + local logits; one decoder pass on the HIP back end"""
        z, pending = self._encode_pre(shard)
        fold = pending and self._fold_elu()
        if not fold:
            z = self.activation_fct(z) if pending else z
        if torch.is_grad_enabled() and self._overlap_ok(shard):
            if isinstance(self.ops, HipOps) and z.shape[1] == 64:
                # ONE node-level product z -> P | Q (AlternateGCN._decoder_inputs' layout): the decoder reads its two column
                # halves in place and hands back one [n_local, 128] gradient
                p, extra, cvec = self._dec_in_joint(z, shard, 1 if fold else 0)
                q = None
            else:
                p, q, extra, cvec = self._dec_in(z, shard, 1 if fold else 0, gather=False)
            st_loc, st_halo = self._st_split(shard)
            return _OverlappedDecoderLoss.apply(p, q, self.ops, self._plan(shard, "sim"), st_loc, st_halo, extra, cvec,
                                                self.mlp[2].weight, self.mlp[2].bias, self.mlp[4].weight.view(-1),
                                                self.mlp[4].bias, labels, pos_weight, shard.e_sim_total)
        if hasattr(self.ops, "decoder_loss") and torch.is_grad_enabled():
            p_full, q, extra, cvec = self._dec_in(z, shard, 1 if fold else 0)
            return self.ops.decoder_loss(p_full, q, self._st(shard, "sim"), extra, cvec, self.mlp[2].weight,
                                         self.mlp[2].bias, self.mlp[4].weight.view(-1), self.mlp[4].bias, labels,
                                         pos_weight, shard.e_sim_total)
        out = self.decode_mlp(self.activation_fct(z) if fold else z, shard)
        return self.ops.bce_sum_over(out, labels, pos_weight, shard.e_sim_total), out.detach()

    def full_state_dict(self):
        """state_dict in the reference layout on EVERY rank: with a sharded categorical embedding the owned rows
        (pad rows excluded) of all ranks are gathered into `embedding.weight [N, D]`; otherwise state_dict() itself"""
        sd = {k: v.detach().clone() for k, v in self.state_dict().items()}
        if not self.sharded_embedding:
            return sd
        lo, hi, n = self._rows
        w = self.embedding.weight.detach()
        world = dist.get_world_size(self.group)
        cnt = torch.zeros(world, dtype=torch.int64, device=w.device)
        cnt[dist.get_rank(self.group)] = hi - lo
        _all_reduce_sum_(cnt, self.group)
        mx = int(cnt.max())
        buf = w.new_zeros((mx, w.shape[1]))
        buf[: hi - lo] = w[: hi - lo]
        allb = _all_gather_rows(buf, self.group).view(world, mx, -1)
        sd["embedding.weight"] = torch.cat([allb[r, : int(cnt[r])] for r in range(world)], dim=0)
        assert sd["embedding.weight"].shape[0] == n
        return sd

    def load_full_state_dict(self, sd, strict: bool = True):
        """inverse of full_state_dict: takes a reference-layout state_dict (`embedding.weight [N, D]`) and keeps this
        rank's rows of the categorical embedding"""
        if self.sharded_embedding:
            lo, hi, n = self._rows
            full = sd["embedding.weight"]
            if full.shape[0] != n:
                raise ValueError(f"embedding.weight has {full.shape[0]} rows, the graph has {n} nodes")
            sd = dict(sd)
            own = self.embedding.weight.detach().clone()
            own[: hi - lo] = full[lo:hi].to(own)
            sd["embedding.weight"] = own
        return self.load_state_dict(sd, strict=strict)

    def sync_gradients(self):
        """one flat all-reduce (sum) of every parameter gradient: 216 KB at default dims.  The gradients are gathered into
        ONE persistent buffer by a single multi-tensor copy, reduced there, and the parameters' `.grad`s become views of
        it (no copy back): 2 launches + the collective per step instead of a concatenation and one copy per parameter."""
        local = self.embedding.weight if self.sharded_embedding else None      # rank-local rows: nothing to reduce
        params = [p for p in self.parameters() if p.grad is not None and p is not local]
        if not params:
            return
        key = tuple((id(p), p.grad.dtype) for p in params)
        cache = self.__dict__.get("_flat_grads")
        if cache is None or cache[0] != key:
            if len({p.grad.dtype for p in params}) != 1:
                raise RuntimeError("gradient all-reduce: mixed gradient dtypes")
            flat = torch.empty(sum(p.numel() for p in params), dtype=params[0].grad.dtype, device=params[0].device)
            views, off = [], 0
            for p in params:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            cache = self.__dict__["_flat_grads"] = (key, flat, views)
        _, flat, views = cache
        torch._foreach_copy_(views, [p.grad for p in params])
        _all_reduce_sum_(flat, self.group)
        for p, v in zip(params, views):
            p.grad = v


def train_step(model: DistAlternateGCN, optimizer, shard, labels, pos_weight):
    """pangnn.py:194-216 on a shard: the loss is the GLOBAL mean, so every rank divides its local sum
    by the global edge count and the gradient all-reduce is a plain sum."""
    optimizer.zero_grad(set_to_none=True)
    loss, out = model.loss_and_logits(shard, labels, pos_weight)
    loss.backward(PF.unit_grad(loss.device))
    model.sync_gradients()
    optimizer.step()
    return loss.detach(), out.detach()


def gather_logits(out_local: torch.Tensor, shard, group=None) -> torch.Tensor:
    """logits of all ranks back in the whole graph's edge order (evaluation / tests)."""
    world = dist.get_world_size(group)
    counts = torch.zeros(world, dtype=torch.long, device=out_local.device)
    counts[shard.rank] = out_local.shape[0]
    _all_reduce_sum_(counts, group)
    mx = int(counts.max())
    buf = torch.zeros(mx, dtype=out_local.dtype, device=out_local.device)
    buf[: out_local.shape[0]] = out_local
    allb = _all_gather_rows(buf, group).view(world, mx)
    masks = _all_gather_rows(shard.owned_mask.to(torch.uint8), group).view(world, -1).bool()
    full = torch.empty(shard.e_sim_total, dtype=out_local.dtype, device=out_local.device)
    for r in range(world):
        full[masks[r]] = allb[r, : int(counts[r])]
    return full
