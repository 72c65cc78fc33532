"""Device-resident graph structure for the message-passing kernels.

The reference hands a COO `edge_index [2,E] int64` to `GCNConv` on every call
(/root/reference/src/gnn.py:158,165) and PyG walks it with index_select / scatter_add.  The HIP
kernels instead want the edges grouped by destination (forward) and by source (backward), with
int32 neighbour ids.  `EdgeStructure` builds both groupings once per `edge_index` tensor on the
GPU (stable radix sort => deterministic per-row order) and memoises the GCN normalisation per
`edge_weight` tensor.  The caller's edge order is never changed: per-edge outputs (logits) stay in
`edge_index` order, `perm` maps sorted positions back.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import _lib


# mini-batch sized square structures are built by one launch (EdgeStructure._small_build); False: always the general
# radix-sort build + index-op plans (what the parity tests compare the small build with)
SMALL_STRUCTURE = os.environ.get("PANGNN_SMALL_STRUCTURE", "1") != "0"


# components of a structure as the native registry (csrc/graph_ops.cpp) names them: bits of `push_native(need)` and of
# pangnn::_prepare_structure(..., need)
(NEED_BY_DST, NEED_BY_SRC, NEED_BAND, NEED_RUNSUM, NEED_PLAN_DST, NEED_PLAN_SRC, NEED_NORM, NEED_NORM_SRC, NEED_ACTIONS,
 NEED_ENTRY) = (1 << i for i in range(10))
_STRUCT_BITS = NEED_BY_DST | NEED_BY_SRC | NEED_BAND | NEED_RUNSUM | NEED_ENTRY
_NORM_BITS = NEED_NORM | NEED_NORM_SRC | NEED_ACTIONS


# Long rows of the propagate kernels (one wave walks one row: spmm_row_kernel): a row of more than LONG_ROW entries — a hub of
# the similarity graph — would keep ONE wave busy for its whole length while the rest of the chip has finished (1e5 entries of
# 256-byte rows: 25 MB through one wave, ~0.5 ms).  A CSR whose longest row exceeds LONG_ROW is therefore run as SEGMENTS of at
# most ~sqrt(longest row) entries (long_segment): the same kernel over the segment table (a row pointer over virtual rows) writes one partial row per
# segment, and the contiguous part sum that the decoder's run sums already use adds a row's partials in order — fixed order,
# no atomics, independent of the grid.  SURVEY.md §7 step 3 ("long rows split across wavefronts with a second-pass reduce").
LONG_ROW = int(os.environ.get("PANGNN_LONG_ROW", "8192"))
LONG_SEG = int(os.environ.get("PANGNN_LONG_SEG", "0"))       # 0: about the square root of the longest row (below)


def long_segment(max_len: int) -> int:
    """entries per segment for a CSR whose longest row has `max_len` entries: the wave that walks a segment and the wave that
    adds the hub's partial rows both work serially (~25 ns per entry / per part row), so the critical path seg + max_len / seg
    is shortest at seg = sqrt(max_len); a power of two in [256, 4096]"""
    if LONG_SEG > 0:
        return LONG_SEG
    seg = 256
    while seg < 4096 and seg * seg < max_len:
        seg *= 2
    return seg


@dataclass
class CSR:
    rowptr: torch.Tensor   # int64 [N+1]
    other: torch.Tensor    # int32 [E]  opposite endpoint of each sorted edge
    perm: torch.Tensor     # int32 [E]  original edge id of each sorted edge

    def long_rows(self, known_short: bool = False):
        """None, or (seg_ptr int64 [V + 1], parts_rowptr int64 [N + 1]) when the longest row exceeds LONG_ROW: segment v covers
        entries [seg_ptr[v], seg_ptr[v + 1]), the segments of row r are [parts_rowptr[r], parts_rowptr[r + 1]) (every row has
        at least one).  Decided once per CSR (one host read-back of the longest row, none for structures whose producer
        vouches for short rows: `known_short`)."""
        hit = self.__dict__.get("_long")
        if hit is None:
            hit = False
            n = self.rowptr.shape[0] - 1
            if not known_short and n > 0 and self.other.shape[0] > LONG_ROW:
                lens = self.rowptr[1:] - self.rowptr[:-1]
                longest = int(lens.max())
                if longest > LONG_ROW:
                    seg = long_segment(longest)
                    nseg = torch.div(lens + (seg - 1), seg, rounding_mode="floor").clamp_(min=1)
                    parts_rowptr = torch.zeros(n + 1, dtype=torch.int64, device=lens.device)
                    torch.cumsum(nseg, 0, out=parts_rowptr[1:])
                    row_of = torch.repeat_interleave(torch.arange(n, device=lens.device), nseg)
                    j = torch.arange(row_of.shape[0], device=lens.device) - parts_rowptr[row_of]
                    seg_ptr = torch.cat([self.rowptr[row_of] + j * seg, self.rowptr[-1:]]).contiguous()
                    hit = (seg_ptr, parts_rowptr)
            self.__dict__["_long"] = hit
        return hit or None


def build_csr(edge_index: torch.Tensor, num_nodes: int, group_by: int, validate: bool = True,
              num_rows: Optional[int] = None) -> CSR:
    """pangnn_csr_build: group_by=1 rows are targets (edge_index[1]), 0 rows are sources.
    `num_nodes` bounds both endpoints; `num_rows` (<= num_nodes) trims rowptr for a rectangular
    (partitioned) graph whose grouped endpoint only takes ids < num_rows."""
    _lib.require_device(edge_index)
    if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError(f"edge_index must be int64 [2,E], got {edge_index.dtype} {tuple(edge_index.shape)}")
    lib = _lib.load()
    ei = edge_index if edge_index.is_contiguous() else edge_index.contiguous()
    e = ei.shape[1]
    dev = ei.device
    rowptr = torch.empty(num_nodes + 1, dtype=torch.int64, device=dev)
    other = torch.empty(e, dtype=torch.int32, device=dev)
    perm = torch.empty(e, dtype=torch.int32, device=dev)
    with _lib.device_guard(dev):
        ws_bytes = lib.pangnn_csr_build_workspace_bytes(e, num_nodes)
        if ws_bytes == 0:
            raise _lib.PangnnHipError("pangnn_csr_build_workspace_bytes failed")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.pangnn_csr_build(ei.data_ptr(), e, e, num_nodes, group_by, rowptr.data_ptr(),
                                        _lib.ptr(other), _lib.ptr(perm), ws.data_ptr(), ws_bytes,
                                        _lib.stream_ptr()), "pangnn_csr_build")
        if validate and e > 0:
            # the flag lives in the workspace; view it as a tensor slice (device -> host 4 bytes)
            off = lib.pangnn_csr_build_flag_ptr(ws.data_ptr(), e) - ws.data_ptr()
            if int(ws[off:off + 4].view(torch.int32).item()) != 0:
                raise ValueError(f"edge_index contains node ids outside [0, {num_nodes})")
    if num_rows is not None and num_rows < num_nodes:
        rowptr = rowptr[: num_rows + 1]
    return CSR(rowptr, other, perm)


class EdgeStructure:
    """Both groupings of one edge_index plus memoised GCN normalisations."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, num_src: Optional[int] = None, hints: Optional[dict] = None):
        """`num_nodes` = number of TARGET rows (= all nodes for a whole graph).  `num_src` (default
        the same) = number of SOURCE rows: a destination-partitioned shard keeps local target ids
        and global source ids (pangnn_amd/dist.py).
        `hints` (optional; set by producers that KNOW them, e.g. SubGraphDataset.batch): {"valid_ids": True} — every id
        is inside [0, num_nodes), skip the range check and its host read-back; {"sorted_by_src": bool} — whether the
        list is source-sorted, skip that test's host read-back; {"band_width": k} — whether the list is the whole-graph
        positional-neighbour pattern (0: it is not), skip that comparison's host read-back.  With all of them a structure
        is built without any device -> host synchronisation (a fresh mini-batch per step, pangnn.py:152-216)."""
        self.hints = dict(hints or {})
        self._key_tensor = edge_index        # the caches are keyed on THIS tensor's address: keep it alive
        self.edge_index = edge_index if edge_index.is_contiguous() else edge_index.contiguous()
        self.num_nodes = int(num_nodes)
        self.num_src = int(num_nodes if num_src is None else num_src)
        self.num_edges = int(edge_index.shape[1])
        self._by_dst: Optional[CSR] = None
        self._by_src: Optional[CSR] = None
        self._norm: Dict[Tuple, Tuple["GcnNorm", Optional[torch.Tensor]]] = {}
        self._runsum = None
        self._band: Optional[int] = None
        self._small_built = False
        self._native = 0                     # NEED_* bits already pushed to the native registry (push_native)

    def _small_build(self) -> bool:
        """A small square structure (a mini-batch of sub-graphs): both CSR orders and the S / T kernels' chunk plans of both
        orders from ONE launch (pangnn_structure_small) instead of ~10 launches per CSR order and ~18 per plan — the same
        tables, entry for entry, as build_csr / _plan_of_sorted_keys (tests/test_hip_parity.py).  False: not applicable."""
        if not SMALL_STRUCTURE or self._by_dst is not None or self._by_src is not None or self.num_src != self.num_nodes:
            return False
        lib = _lib.load()
        e, n = self.num_edges, self.num_nodes
        if not lib.pangnn_structure_small_supported(e, n):
            return False
        ct = int(lib.pangnn_decoder_chunk_tiles_for(e))
        span = 32 * ct
        nc = (e + span - 1) // span
        dev = self.edge_index.device
        ea, na, ca = -(-e // 4) * 4, -(-(n + 1) // 2) * 2, -(-nc // 4) * 4             # 16-byte aligned segments
        i32 = torch.empty(2 * (3 * ea + ca) + 4, dtype=torch.int32, device=dev)
        i64 = torch.empty(2 * (2 * na + 2), dtype=torch.int64, device=dev)
        seg32 = [i32[k * ea:k * ea + e] for k in range(6)] + [i32[6 * ea + k * ca:6 * ea + k * ca + nc] for k in range(2)]
        other_d, perm_d, keys_d, other_s, perm_s, keys_s, poff_d, poff_s = seg32
        bad = i32[6 * ea + 2 * ca:6 * ea + 2 * ca + 1]
        rp_d, prp_d, rp_s, prp_s = (i64[k * na:k * na + n + 1] for k in range(4))
        last_d, last_s = i64[4 * na:4 * na + 1], i64[4 * na + 2:4 * na + 3]
        with _lib.device_guard(dev):
            _lib.check(lib.pangnn_structure_small(
                self.edge_index.data_ptr(), e, e, n, span, rp_d.data_ptr(), other_d.data_ptr(), perm_d.data_ptr(),
                keys_d.data_ptr(), poff_d.data_ptr(), prp_d.data_ptr(), last_d.data_ptr(), rp_s.data_ptr(),
                other_s.data_ptr(), perm_s.data_ptr(), keys_s.data_ptr(), poff_s.data_ptr(), prp_s.data_ptr(),
                last_s.data_ptr(), bad.data_ptr(), _lib.stream_ptr()), "pangnn_structure_small")
            if not self.hints.get("valid_ids", False) and int(bad.item()) != 0:
                raise ValueError(f"edge_index contains node ids outside [0, {n})")
        self._by_dst, self._by_src = CSR(rp_d, other_d, perm_d), CSR(rp_s, other_s, perm_s)
        self._by_dst.__dict__["_long"] = self._by_src.__dict__["_long"] = False      # <= 16384 edges in all: no split, no read-back
        self._small_built = True
        from types import SimpleNamespace
        n_bound = nc + min(n, e)
        plans = self.__dict__.setdefault("_csr_plans", {})
        for by, poff, prp, keys, last in (("dst", poff_d, prp_d, keys_d, last_d), ("src", poff_s, prp_s, keys_s, last_s)):
            plan = SimpleNamespace(n_parts=n_bound, part_off=poff, part_rowptr=prp, keys=keys, chunk_tiles=ct, _last=last)
            plan.n_parts_exact = (lambda pl: (lambda: int(pl._last) + 1))(plan)
            plans[(by, ct)] = plan
        return True

    @property
    def by_dst(self) -> CSR:
        if self._by_dst is None and self._small_build():
            return self._by_dst
        if self._by_dst is None:
            nmax = max(self.num_nodes, self.num_src)
            self._by_dst = build_csr(self.edge_index, nmax, 1, validate=not self.hints.get("valid_ids", False),
                                     num_rows=self.num_nodes)
            if self.num_nodes < nmax and self.num_edges and int(self.edge_index[1].max()) >= self.num_nodes:
                raise ValueError("target id outside the local row range")
            if self.hints.get("valid_ids", False):       # a producer-vouched list (a collated batch): built without any
                self._by_dst.__dict__["_long"] = False   # read-back, so no read-back of the longest row either
        return self._by_dst

    @property
    def by_src(self) -> CSR:
        if self._by_src is None and self._small_build():
            return self._by_src
        if self._by_src is None:
            nmax = max(self.num_nodes, self.num_src)
            self._by_src = build_csr(self.edge_index, nmax, 0, validate=False, num_rows=self.num_src)
            if self.hints.get("valid_ids", False):
                self._by_src.__dict__["_long"] = False
        return self._by_src

    def band_width(self) -> int:
        """k > 0 if this edge list IS the positional-neighbour graph of a whole genome set as the reference builds it
        (dataset.py:356-361: edges (i, j), j in [i - k, i + k] within [0, N), self loops included, in nested-loop order) —
        a band matrix, propagated without index arrays (pangnn_band_propagate); 0 for any other list.  Decided once per
        structure by comparing with the generated pattern (one host sync)."""
        if self._band is None and "band_width" in self.hints:      # the producer knows (a collated batch: 0), no host sync
            self._band = int(self.hints["band_width"])
        if self._band is None:
            self._band = 0
            n, e = self.num_nodes, self.num_edges
            if self.num_src == n and n > 0 and e > 0:
                for k in range(1, 9):
                    if k < n and e == n * (2 * k + 1) - k * (k + 1):
                        dev = self.edge_index.device
                        i = torch.arange(n, dtype=torch.int64, device=dev).repeat_interleave(2 * k + 1)
                        j = i + torch.arange(-k, k + 1, dtype=torch.int64, device=dev).repeat(n)
                        keep = (j >= 0) & (j < n)
                        if torch.equal(self.edge_index, torch.stack([i[keep], j[keep]])):
                            self._band = k
                        break
        return self._band

    @staticmethod
    def _plan_of_sorted_keys(keys: torch.Tensor, n_rows: int, chunk_tiles: int = 1):
        """Layout of the per-(chunk, key) partial rows for an edge order whose `keys` are non-decreasing; a chunk is
        `chunk_tiles` consecutive 32-edge tiles (1: the strict-fp32 kernels of decoder.hip; pangnn_decoder_chunk_tiles()
        = 16: the S / T kernels of decoder16.hip, whose waves carry an open run from tile to tile inside a chunk):
          part_off[c]    index of chunk c's first part
          part_rowptr[r] parts of row r are [part_rowptr[r], part_rowptr[r+1])   (consecutive: sorted)
          keys           int32 copy handed to the kernel"""
        from types import SimpleNamespace
        e = keys.shape[0]
        span = 32 * int(chunk_tiles)
        flags = (torch.arange(e, device=keys.device) % span) == 0
        flags[1:] |= keys[1:] != keys[:-1]
        part_id = torch.cumsum(flags, 0) - 1
        # Everything stays on the device (no host read-back): the part buffer is sized by an upper bound — one part per
        # chunk start plus one per key change — and a row's first part is the part of its first entry (a key change
        # always starts a part).  `n_parts_exact()` reads the true count when somebody needs it (bench accounting).
        n_bound = (e + span - 1) // span + min(int(n_rows), e)
        first = torch.searchsorted(keys, torch.arange(n_rows + 1, device=keys.device, dtype=keys.dtype))
        pid_ext = torch.cat([part_id, part_id[-1:] + 1])
        plan = SimpleNamespace(n_parts=n_bound, part_off=part_id[::span].to(torch.int32).contiguous(),
                               part_rowptr=pid_ext[first].contiguous(), keys=keys.to(torch.int32).contiguous(),
                               chunk_tiles=int(chunk_tiles), _last=part_id[-1:])
        plan.n_parts_exact = lambda: int(plan._last) + 1
        return plan

    def csr_plan(self, by: str, chunk_tiles: int = 1):
        """run-sum plan of the CSR order `by` in {"dst", "src"} (pangnn_decoder_dgrad_f32: perm = that CSR's perm)"""
        cache = self.__dict__.setdefault("_csr_plans", {})
        key = (by, int(chunk_tiles))
        if key not in cache:
            csr = self.by_dst if by == "dst" else self.by_src
            n_rows = self.num_nodes if by == "dst" else self.num_src
            keys = self.edge_index[1 if by == "dst" else 0][csr.perm.long()] if self.num_edges else \
                self.edge_index.new_empty(0)
            cache[key] = self._plan_of_sorted_keys(keys, n_rows, chunk_tiles) if self.num_edges else None
        return cache[key]

    def runsum_plan(self, chunk_tiles: int = 1):
        """If the caller's edge order is sorted by source: the layout of the per-(chunk, source) partial rows the decoder
        training kernels can emit (include/pangnn_hip.h, `part_buf` / `part_off`), else None.
          part_off[c]    index of chunk c's first part
          part_rowptr[s] parts of source s are [part_rowptr[s], part_rowptr[s+1])   (consecutive: sorted)"""
        if self._runsum is None:
            src, e = self.edge_index[0], self.num_edges
            if e == 0:
                self._runsum = False
            elif "sorted_by_src" in self.hints:
                self._runsum = {} if self.hints["sorted_by_src"] else False
            else:
                self._runsum = {} if bool((src[1:] >= src[:-1]).all()) else False
        if self._runsum is False:
            return None
        ct = int(chunk_tiles)
        if ct not in self._runsum:
            # a source-sorted list IS its by-source order (stable sort of a sorted list): the small build's plan of that
            # order is this plan
            self._small_build()              # no-op unless applicable and nothing is built yet
            small = self.__dict__.get("_csr_plans", {}).get(("src", ct)) if self._small_built else None
            self._runsum[ct] = small if small is not None else \
                self._plan_of_sorted_keys(self.edge_index[0], self.num_src, ct)
        return self._runsum[ct]

    def native_has(self, need: int, norm=None, x=None) -> bool:
        """everything `need` names is in the native registry already (as far as this object knows: the registry may have
        evicted it, in which case the op that misses it calls pangnn::_prepare_structure, i.e. push_native(force=True))"""
        if (self._native & need & (_STRUCT_BITS | NEED_PLAN_DST | NEED_PLAN_SRC)) != (need & (_STRUCT_BITS | NEED_PLAN_DST | NEED_PLAN_SRC)) \
                or not (self._native & NEED_ENTRY):
            return False
        nb = need & _NORM_BITS
        if not nb:
            return True
        if norm is None or (norm._native & nb & ~NEED_ACTIONS) != (nb & ~NEED_ACTIONS):
            return False
        return not (nb & NEED_ACTIONS) or (x is not None and norm._native_actions == (x.data_ptr(), x._version))

    def push_native(self, need: int, edge_weight: Optional[torch.Tensor] = None, x: Optional[torch.Tensor] = None,
                    force: bool = False):
        """Build (lazily, with the C entry points this class already uses) and push to the native structure registry
        (csrc/graph_ops.cpp) the components `need` names — NEED_* bits — so that the per-step ops `torch.ops.pangnn.
        {gcn_propagate, embed_conv_in[_linear], decoder_loss, decoder_mlp}` and their backward ops find them by the identity of
        `edge_index` (/ `edge_weight` / `x`) without entering Python.  Square structures only (a partitioned shard's
        rectangular structures take the ctypes route).  `force`: push again what this object believes is there (the
        registry evicts least-recently-used entries)."""
        if self.num_src != self.num_nodes:
            raise ValueError("the native registry holds square structures (whole graphs, collated batches)")
        from .torch_ops import ops
        ei, n, e = self._key_tensor, self.num_nodes, self.num_edges
        have = 0 if force else self._native
        ct = int(_lib.load().pangnn_decoder_chunk_tiles_for(e))
        todo = need & _STRUCT_BITS & ~have
        if todo or not (have & NEED_ENTRY):
            by_dst, by_src, band, srt, plan = [], [], -1, -1, None
            short = self._small_built or bool(self.hints.get("valid_ids", False))
            if todo & NEED_BY_DST:
                c = self.by_dst
                by_dst = [c.rowptr, c.other, c.perm] + list(c.long_rows(short) or ())
            if todo & NEED_BY_SRC:
                c = self.by_src
                by_src = [c.rowptr, c.other, c.perm] + list(c.long_rows(short) or ())
            if todo & NEED_BAND:
                band = self.band_width()
            if todo & NEED_RUNSUM:
                plan = self.runsum_plan(ct)
                srt = 0 if plan is None else 1
            ops._register_structure(ei, n, self.num_src, self.edge_index, by_dst, by_src, band, srt)
            if plan is not None:
                ops._register_plan(ei, n, 0, ct, plan.part_off, plan.part_rowptr, plan.keys, int(plan.n_parts))
            have |= todo | NEED_ENTRY
        for bit, by, kind in ((NEED_PLAN_DST, "dst", 1), (NEED_PLAN_SRC, "src", 2)):
            if need & bit & ~have:
                if e:
                    plan = self.csr_plan(by, ct)
                    ops._register_plan(ei, n, kind, ct, plan.part_off, plan.part_rowptr, plan.keys, int(plan.n_parts))
                have |= bit
        self._native = have
        nb = need & _NORM_BITS
        if nb:
            norm = self.gcn_norm(edge_weight)
            nh = 0 if force else norm._native
            want_src = bool(nb & NEED_NORM_SRC)
            if (NEED_NORM & ~nh) or (want_src and (NEED_NORM_SRC & ~nh)):
                by_src = norm.by_src if want_src else norm._by_src
                ops._register_norm(ei, n, norm.weight_ref, norm.deg_inv_sqrt, norm.by_dst, norm.orig, by_src)
                nh |= NEED_NORM | (NEED_NORM_SRC if by_src is not None else 0)
            if nb & NEED_ACTIONS:
                if x is None:
                    raise ValueError("NEED_ACTIONS needs the feature tensor")
                key = (x.data_ptr(), x._version)
                if force or norm._native_actions != key:
                    from .functional import _node_actions
                    r, s_ = _node_actions(x, self, norm)
                    ops._register_actions(ei, n, norm.weight_ref, x, r, s_)
                    norm._native_actions = key
            norm._native = nh

    def gcn_norm(self, edge_weight: Optional[torch.Tensor], gather_dis=None) -> "GcnNorm":
        """norm for this edge_weight tensor (None = unit weights); cached on tensor identity.
        `gather_dis(dis_local [n_dst]) -> dis_src [n_src]` supplies the source nodes' deg^-1/2 for a
        partitioned shard (an all-gather); omitted for a whole graph."""
        key = None if edge_weight is None else (edge_weight.data_ptr(), edge_weight._version,
                                                tuple(edge_weight.shape))
        hit = self._norm.get(key)
        if hit is None:
            # the entry keeps `edge_weight` alive: a freed tensor's address can be handed to a new same-shape
            # tensor (fresh tensors all have _version 0), which would then match this key
            hit = (GcnNorm(self, edge_weight, gather_dis), edge_weight)
            hit[0].weight_ref = edge_weight              # the caller's tensor (cache key): the dispatcher ops pass it on
            if len(self._norm) >= 4:
                self._norm.pop(next(iter(self._norm)))
            self._norm[key] = hit
        return hit[0]


class GcnNorm:
    """k1-k3 of SURVEY.md §2.2: deg^-1/2[src] * w * deg^-1/2[dst], in both CSR orders."""

    def __init__(self, st: EdgeStructure, edge_weight: Optional[torch.Tensor], gather_dis=None):
        lib = _lib.load()
        dev = st.edge_index.device
        e, n = st.num_edges, st.num_nodes
        if edge_weight is not None:
            _lib.require_device(edge_weight)
            if edge_weight.dim() != 1 or edge_weight.shape[0] < e:
                raise ValueError(f"edge_weight must be [E>={e}], got {tuple(edge_weight.shape)}")
            edge_weight = edge_weight.detach().to(torch.float32).contiguous()
        d = st.by_dst
        ea, na = -(-e // 4) * 4, -(-n // 4) * 4                          # one allocation, 16-byte aligned segments
        small = e <= (1 << 20)                  # small graphs: the by-source table too (a launch-bound step counts allocations)
        buf = torch.empty(na + (3 if small else 2) * ea, dtype=torch.float32, device=dev)
        self.deg_inv_sqrt = buf[:n]
        self.by_dst = buf[na:na + e]                                     # CSR(dst) order
        self.orig = buf[na + ea:na + ea + e]                             # caller's edge order
        self._by_src_buf = buf[na + 2 * ea:na + 2 * ea + e] if small else None      # CSR(src) order, filled on first use
        with _lib.device_guard(dev):
            if gather_dis is None:
                if st.num_src != st.num_nodes:
                    raise ValueError("a rectangular structure needs gather_dis= for gcn_norm")
                _lib.check(lib.pangnn_gcn_norm_f32(d.rowptr.data_ptr(), _lib.ptr(d.other), _lib.ptr(d.perm),
                                                   _lib.ptr(edge_weight), n, e, self.deg_inv_sqrt.data_ptr(),
                                                   _lib.ptr(self.by_dst), _lib.ptr(self.orig),
                                                   _lib.stream_ptr()), "pangnn_gcn_norm_f32")
            else:
                _lib.check(lib.pangnn_gcn_degree_f32(d.rowptr.data_ptr(), _lib.ptr(d.perm), _lib.ptr(edge_weight),
                                                     n, self.deg_inv_sqrt.data_ptr(), _lib.stream_ptr()),
                           "pangnn_gcn_degree_f32")
                dis_src = gather_dis(self.deg_inv_sqrt).contiguous()
                assert dis_src.shape[0] >= st.num_src and dis_src.dtype == torch.float32
                _lib.check(lib.pangnn_gcn_edge_norm_f32(d.rowptr.data_ptr(), _lib.ptr(d.other), _lib.ptr(d.perm),
                                                        _lib.ptr(edge_weight), dis_src.data_ptr(),
                                                        self.deg_inv_sqrt.data_ptr(), n, e, _lib.ptr(self.by_dst),
                                                        _lib.ptr(self.orig), _lib.stream_ptr()),
                           "pangnn_gcn_edge_norm_f32")
        self._st = st
        self._by_src: Optional[torch.Tensor] = None
        self._native, self._native_actions = 0, None          # what the native registry holds of this normalisation
        self.weight_ref = None                                # the caller's tensor (set by EdgeStructure.gcn_norm)

    @property
    def by_src(self) -> torch.Tensor:
        """norm re-ordered into CSR(src) order (needed by the transposed propagate only)."""
        if self._by_src is None:
            st, lib = self._st, _lib.load()
            s = st.by_src
            out = self._by_src_buf if self._by_src_buf is not None else torch.empty_like(self.orig)
            with _lib.device_guard(out.device):
                _lib.check(lib.pangnn_permute_f32(_lib.ptr(self.orig), _lib.ptr(s.perm), _lib.ptr(out),
                                                  st.num_edges, _lib.stream_ptr()), "pangnn_permute_f32")
            self._by_src = out
        return self._by_src


# --------------------------------------------------------------------------------------
# structure cache: the reference passes raw tensors on every call, so memoise on identity
# --------------------------------------------------------------------------------------
_CACHE: Dict[Tuple, EdgeStructure] = {}
_CACHE_MAX = 16


def structure_key(edge_index: torch.Tensor, num_nodes: int):
    """identity of (edge_index tensor, node count): what the structure caches are keyed on"""
    return (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), int(num_nodes), edge_index.device.index)


def structure_of(edge_index: torch.Tensor, num_nodes: int, holder=None, name: str = "") -> EdgeStructure:
    """EdgeStructure for `edge_index`.  If `holder` (a Data/Batch object) is given the structure is
    kept on it (`holder._pangnn_structs[name]`), otherwise in a small identity-keyed cache."""
    if torch.compiler.is_compiling():
        return TracedStructure(edge_index, num_nodes)
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), int(num_nodes),
           edge_index.device.index)
    if holder is not None:
        d = getattr(holder, "_pangnn_structs", None)
        if d is None:
            d = {}
            try:
                setattr(holder, "_pangnn_structs", d)
            except Exception:
                d = None
        if d is not None:
            hit = d.get(name)
            if hit is not None and hit[0] == key:
                return hit[1]
            hints = getattr(holder, "_pangnn_hints", None)
            st = EdgeStructure(edge_index, num_nodes, hints=None if hints is None else hints.get(name))
            d[name] = (key, st)
            register(st, key)
            return st
    hit = _CACHE.get(key)
    if hit is not None:
        return hit
    st = EdgeStructure(edge_index, num_nodes)
    if len(_CACHE) >= _CACHE_MAX:
        _CACHE.pop(next(iter(_CACHE)))
    _CACHE[key] = st
    return st


class TracedStructure:
    """What torch.compile sees of a graph while it traces the model: the tensors and the sizes, nothing built.  The
    dispatcher ops (torch_ops.py) take `edge_index` / `edge_weight` as tensors and look the real EdgeStructure / GcnNorm up
    by identity when the compiled graph RUNS (structure_of without a holder: the small global cache)."""
    traced = True

    def __init__(self, edge_index: torch.Tensor, num_nodes: int):
        self._key_tensor = self.edge_index = edge_index
        self.num_nodes = self.num_src = int(num_nodes)
        self.num_edges = edge_index.shape[1]

    def gcn_norm(self, edge_weight, gather_dis=None):
        return TracedNorm(edge_weight)


class TracedNorm:
    def __init__(self, edge_weight):
        self.weight_ref = edge_weight


def register(st, key=None):
    """make `st` the answer of structure_of(st.edge_index, st.num_nodes) without a holder (the dispatcher ops receive the
    raw tensors and look the structure up by identity)"""
    if getattr(st, "traced", False):
        return
    if key is None:
        ei = st._key_tensor
        key = (ei.data_ptr(), ei._version, tuple(ei.shape), int(st.num_nodes), ei.device.index)
    if _CACHE.get(key) is not st:
        if len(_CACHE) >= _CACHE_MAX:
            _CACHE.pop(next(iter(_CACHE)))
        _CACHE[key] = st


def clear_cache():
    for st in _CACHE.values():
        st._native = 0
    _CACHE.clear()
    from .torch_ops import ops
    ops._registry_clear(torch.empty(0))


def forget(key, edge_index: Optional[torch.Tensor] = None):
    """drop one entry of the identity-keyed cache (a buffer whose content was rewritten in place) — and, given the buffer,
    its entry of the native registry (same address, same version counter, new content)"""
    st = _CACHE.pop(key, None)
    if st is not None:
        st._native = 0
    if edge_index is not None and edge_index.is_cuda:
        from .torch_ops import ops
        ops._registry_forget(edge_index, int(key[3]))
