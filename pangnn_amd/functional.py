"""autograd glue over the C ABI: each Function enqueues HIP kernels from libpangnn_hip.so on torch's
current stream.  No torch-native arithmetic stands in for a kernel here."""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib
from .graph import CSR, EdgeStructure, GcnNorm


ROWS16 = (torch.bfloat16, torch.float16)        # the two 2-byte row formats (PANGNN_DTYPE_BF16 / _F16)

def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


# Optional kernel timing for bench.py: when KERNEL_TIMER is a dict, every spmm launch whose `tag` is
# a key gets a (start, end) torch.cuda.Event pair recorded on the launch stream (= torch's current
# stream, the one passed to the C ABI) around the launch.
KERNEL_TIMER = None

# Matrix-pipe mode of the decoder backward / training kernel (include/pangnn_hip.h, `precision`):
# 1 (default) = every product on the bf16 matrix pipe with operands split into bf16 terms (hi + mid + lo carries
#     the 24 significand bits; partial products accumulated in fp32) — fp32-level error (tests: 2e-5 on logits
#     against fp64), 0.78x the time of mode 0 because f32 MFMA shares the SIMD's vector lanes;
# 0 = v_mfma_f32_32x32x2_f32 everywhere (bit-exact fp32 FMA chains).  Environment: PANGNN_DECODER_PRECISION.
DECODER_PRECISION = int(os.environ.get("PANGNN_DECODER_PRECISION", "1"))


def _timer_start(tag):
    if KERNEL_TIMER is None or tag not in KERNEL_TIMER:
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


def _timer_stop(tag, ev0):
    if ev0 is not None:
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        KERNEL_TIMER[tag].append((ev0, ev1))


def spmm_csr(csr: CSR, val: Optional[torch.Tensor], x: torch.Tensor, n_rows: int,
             bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
             accumulate: bool = False, tag: Optional[str] = None) -> torch.Tensor:
    """out[r] (+)= bias + sum_e val[e] * x[other[e]]  — pangnn_spmm_csr_f32, or pangnn_spmm_csr_bf16 / _f16 when the
    gathered rows are stored as bfloat16 / float16 (fp32 weights / accumulation / result either way)."""
    lib = _lib.load()
    _lib.require_device(x, csr.rowptr, val, bias)
    f = x.shape[1]
    long = csr.long_rows() if csr.other is not None and f in (16, 32, 64, 128, 256) else None
    if long is not None:
        # a hub row: the same kernel over segments of at most LONG_SEG entries (one partial row each), then the contiguous
        # part sum adds a row's partials in order, with the bias / accumulation of the call (graph.CSR.long_rows)
        seg_ptr, parts_rowptr = long
        parts = spmm_csr(_SegmentRows(seg_ptr, csr.other), val, x, seg_ptr.shape[0] - 1, tag=tag)
        if out is None:
            out = torch.empty(n_rows, f, dtype=torch.float32, device=x.device)
            accumulate = False
        with _lib.device_guard(x.device):
            _lib.check(lib.pangnn_spmm_csr_f32(parts_rowptr.data_ptr(), None, None, parts.data_ptr(), parts.stride(0),
                                               parts.shape[0], _lib.ptr(None if bias is None else _f32c(bias)), out.data_ptr(),
                                               out.stride(0), n_rows, parts.shape[0], f, int(accumulate), _lib.stream_ptr()),
                       "pangnn_spmm_csr_f32(long-row parts)")
        return out
    if x.dtype in ROWS16 and f in (32, 64, 128, 256):
        if x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 8:
            x = x.contiguous()
        fn, name = (lib.pangnn_spmm_csr_bf16, "pangnn_spmm_csr_bf16") if x.dtype == torch.bfloat16 \
            else (lib.pangnn_spmm_csr_f16, "pangnn_spmm_csr_f16")
    else:
        x = _f32c(x)
        fn, name = lib.pangnn_spmm_csr_f32, "pangnn_spmm_csr_f32"
    if out is None:
        out = torch.empty(n_rows, f, dtype=torch.float32, device=x.device)
        accumulate = False
    timed = KERNEL_TIMER is not None and tag in KERNEL_TIMER
    with _lib.device_guard(x.device):
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        _lib.check(fn(csr.rowptr.data_ptr(), _lib.ptr(csr.other), _lib.ptr(val),
                      x.data_ptr(), x.stride(0), x.shape[0], _lib.ptr(bias),
                      out.data_ptr(), out.stride(0), n_rows, int(csr.other.shape[0]), f,
                      int(accumulate), _lib.stream_ptr()), name)
        if timed:
            ev1.record()
            KERNEL_TIMER[tag].append((ev0, ev1))
    return out


class _SegmentRows:
    """the segment table of a CSR with long rows, shaped like a CSR for spmm_csr (virtual rows = segments)"""

    def __init__(self, seg_ptr, other):
        self.rowptr, self.other = seg_ptr, other

    def long_rows(self):
        return None


def segment_sum_rows(csr: CSR, m: torch.Tensor, col_off: int, f: int, n_rows: int,
                     out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """out[r] (+)= sum_{k in row r} m[perm[k], col_off:col_off+f] — pangnn_segment_sum_rows_f32."""
    lib = _lib.load()
    _lib.require_device(m)
    assert m.dtype == torch.float32 and m.stride(1) == 1
    if out is None:
        out = torch.empty(n_rows, f, dtype=torch.float32, device=m.device)
        accumulate = False
    with _lib.device_guard(m.device):
        _lib.check(lib.pangnn_segment_sum_rows_f32(csr.rowptr.data_ptr(), _lib.ptr(csr.perm), m.data_ptr(),
                                                   m.stride(0), m.shape[0], col_off, out.data_ptr(),
                                                   out.stride(0), n_rows, f, int(accumulate),
                                                   _lib.stream_ptr()), "pangnn_segment_sum_rows_f32")
    return out


COLSUM_SMALL_ROWS = 4096


def colsum(g: torch.Tensor) -> torch.Tensor:
    """column sums of dL/dout (GCNConv's bias gradient) in fp32: one launch for the short matrices of a mini-batch
    (torch's column reduction is a zero-fill + a reduce kernel there), torch's two-stage sum for long ones"""
    if g.is_cuda and g.dim() == 2 and g.shape[0] <= COLSUM_SMALL_ROWS and g.stride(1) == 1 and 0 < g.shape[1] <= 1024 \
            and (g.dtype == torch.float32 or g.dtype in ROWS16):
        lib = _lib.load()
        out = torch.empty(g.shape[1], dtype=torch.float32, device=g.device)
        with _lib.device_guard(g.device):
            _lib.check(lib.pangnn_colsum_small(g.data_ptr(), _dt(g), g.stride(0), g.shape[0], g.shape[1], out.data_ptr(),
                                               _lib.stream_ptr()), "pangnn_colsum_small")
        return out
    return g.sum(dim=0, dtype=torch.float32)


class _Propagate(torch.autograd.Function):
    """out = A_hat @ x + bias with A_hat given by (structure, norm).  Backward is the transposed
    propagate over the source-grouped CSR (k5^T); edge weights are not differentiated (they are a
    leaf without grad in the reference: SURVEY.md §8 a6)."""

    @staticmethod
    def forward(ctx, x, bias, st: EdgeStructure, norm: GcnNorm, tag=None, out_dtype=None):
        """`out_dtype` = torch.bfloat16 / torch.float16 (autocast, a propagate-first GCNConv): the result is what an autocast
        Linear consumes, i.e. it is cast to the autocast type first thing — done here once and stored, so the Linear reads
        2-byte rows, its dL/dx comes back in that type (autograd's dtype rule; in the reference the backward of that cast hands
        fp32 copies of the 16-bit values on) and the transposed propagate gathers it as stored: the same values as the
        reference's, half the gather bytes."""
        ctx.st, ctx.norm, ctx.tag = st, norm, tag
        ctx.has_bias = bias is not None
        ctx.x_dtype = x.dtype            # bfloat16 / float16 rows are gathered as stored (half the bytes); result fp32
        out = spmm_csr(st.by_dst, norm.by_dst, x, st.num_nodes, bias=None if bias is None else _f32c(bias),
                       tag=None if tag is None else tag + ".fwd")
        return out.to(out_dtype) if out_dtype in ROWS16 else out

    @staticmethod
    def backward(ctx, g):
        st, norm = ctx.st, ctx.norm
        if not (g.dtype in ROWS16 and g.shape[1] in (32, 64, 128, 256)):     # 2-byte rows: gathered as stored
            g = _f32c(g)
        gx = spmm_csr(st.by_src, norm.by_src, g, st.num_src,
                      tag=None if ctx.tag is None else ctx.tag + ".bwd") if ctx.needs_input_grad[0] else None
        gb = colsum(g) if (ctx.has_bias and ctx.needs_input_grad[1]) else None
        if gx is not None and gx.dtype != ctx.x_dtype:
            gx = gx.to(ctx.x_dtype)
        return gx, gb, None, None, None, None


# The per-step operators (dense layer, GCN propagate, first layer by linearity, training / inference decoder, criterion)
# are registered dispatcher ops `torch.ops.pangnn.*` (torch_ops.py: fake kernels, autograd formulas built from registered
# ops) — what FakeTensor tracing / torch.compile / any TorchDispatchMode need to see them.  Same kernels and the same host
# code as the ctypes autograd.Functions in this file, bit-identical results.  USE_DISPATCHER_OPS (PANGNN_DISPATCHER_OPS):
#   True / "1" (default since round 4)  every per-step operator of a whole (square) graph goes through its registered op —
#                     eager, traced or captured alike: ONE route.  The registered-op wrapper (torch.library's autograd
#                     plumbing) costs ~25 us of host time per call: nothing on a whole-graph step (12.83 vs 12.88 ms), nothing
#                     when a mini-batch step is replayed from a captured HIP graph (train.GraphedTrainStep /
#                     ReplayedFreshStep, the product's mini-batch paths), +0.16 ms on an eagerly launched 0.47 ms mini-batch
#                     step, which is why round 3 defaulted to "auto";
#   "auto"            through the ops only when somebody can observe them — torch.compile is tracing, or a dispatch /
#                     function mode is active (FakeTensorMode, make_fx, a TorchDispatchMode) — and straight to the
#                     autograd.Functions otherwise;
#   False / "0"       never (the route a partitioned shard always takes: its rectangular structures have no tensor-only
#                     description).
_mode = os.environ.get("PANGNN_DISPATCHER_OPS", "1").lower()
USE_DISPATCHER_OPS = "auto" if _mode == "auto" else False if _mode in ("0", "false", "off") else True
_dispatch_modes = getattr(torch._C, "_len_torch_dispatch_stack", lambda: 0)
_function_modes = getattr(torch._C, "_len_torch_function_stack", lambda: 0)


def observed() -> bool:
    """a tracer or a dispatch / function mode is watching the calls of this thread"""
    return torch.compiler.is_compiling() or _dispatch_modes() > 0 or _function_modes() > 0


def _via_ops(st: Optional[EdgeStructure] = None, tag=None) -> bool:
    """route this call through torch.ops.pangnn.*: whole (square) graphs, no kernel timer, and the mode above"""
    use = USE_DISPATCHER_OPS
    if use == "auto":
        use = observed()
    return bool(use) and (st is None or st.num_src == st.num_nodes) and (KERNEL_TIMER is None or tag is None)


def propagate(x, bias, st: EdgeStructure, norm: GcnNorm, tag=None, out_dtype=None):
    if _via_ops(st, tag) and getattr(norm, "weight_ref", norm) is not norm:
        from . import torch_ops
        _lib.require_device(x, bias)
        return torch_ops.gcn_propagate(x, bias, st, norm, False, out_dtype)
    return _Propagate.apply(x, bias, st, norm, tag, out_dtype)


class _BandPropagate(torch.autograd.Function):
    """out = A_hat x + bias for the positional-neighbour band graph (EdgeStructure.band_width() = k > 0, unit weights):
    pangnn_band_propagate — the sums of the generic propagate in the same order, no index arrays.  The band is
    symmetric, so backward is the same kernel on the upstream gradient, which also emits the bias gradient (its column
    sums) in the same pass."""

    @staticmethod
    def forward(ctx, x, bias, dis, k):
        ctx.k, ctx.has_bias, ctx.x_dtype = int(k), bias is not None, x.dtype
        ctx.save_for_backward(dis)
        return _band_call(x, None if bias is None else _f32c(bias), dis, int(k), False)[0]

    @staticmethod
    def backward(ctx, g):
        (dis,) = ctx.saved_tensors
        want_b = ctx.has_bias and ctx.needs_input_grad[1]
        gx, gb = _band_call(_f32c(g), None, dis, ctx.k, want_b)
        if not ctx.needs_input_grad[0]:
            gx = None
        elif gx.dtype != ctx.x_dtype:
            gx = gx.to(ctx.x_dtype)
        return gx, gb, None, None


def _band_call(x, bias, dis, k, want_colsum):
    lib = _lib.load()
    _lib.require_device(x, bias, dis)
    x = _rows_any(x)
    n, f = x.shape
    out = torch.empty(n, f, dtype=torch.float32, device=x.device)
    cs = torch.empty(f, dtype=torch.float32, device=x.device) if want_colsum else None
    with _lib.device_guard(x.device):
        ws_bytes = lib.pangnn_band_propagate_workspace_bytes(f) if want_colsum else 0
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if want_colsum else None
        _lib.check(lib.pangnn_band_propagate(x.data_ptr(), _dt(x), x.stride(0), dis.data_ptr(), _lib.ptr(bias), out.data_ptr(),
                                             out.stride(0), n, f, k, _lib.ptr(cs), _lib.ptr(ws), ws_bytes,
                                             _lib.stream_ptr()), "pangnn_band_propagate")
    return out, cs


def band_propagate(x, bias, st: EdgeStructure, norm: GcnNorm):
    """propagate over a band structure (st.band_width() > 0, unit weights, F in {64, 128})"""
    return _BandPropagate.apply(x, bias, norm.deg_inv_sqrt, st.band_width())


def _band_ok(x, st: EdgeStructure, unit_weights: bool) -> bool:
    return bool(unit_weights) and x.dim() == 2 and x.shape[1] in (64, 128) and st.num_src == st.num_nodes \
        and st.band_width() > 0


def propagate_any(x, bias, st: EdgeStructure, norm: GcnNorm, unit_weights: bool, tag=None, out_dtype=None):
    """GCNConv's message passing: the band kernel when the structure is the positional-neighbour band with unit weights
    (whole-graph mode of the reference), the general CSR kernels otherwise.  `out_dtype`: see _Propagate.forward."""
    if _via_ops(st, tag) and getattr(norm, "weight_ref", norm) is not norm:
        from . import torch_ops
        _lib.require_device(x, bias)
        return torch_ops.gcn_propagate(x, bias, st, norm, unit_weights, out_dtype)   # the op makes the same band / CSR choice
    if _band_ok(x, st, unit_weights) and (KERNEL_TIMER is None or tag is None or (tag + ".fwd") not in KERNEL_TIMER):
        y = band_propagate(x, bias, st, norm)
        return y.to(out_dtype) if out_dtype in ROWS16 else y
    return propagate(x, bias, st, norm, tag, out_dtype)


class _EdgeGatherConcat(torch.autograd.Function):
    """cat(z[src], z[dst] [, extra]) per edge (gnn.py:173-175).  Backward = two segment sums."""

    @staticmethod
    def forward(ctx, z, st: EdgeStructure, extra):
        lib = _lib.load()
        _lib.require_device(z, extra)
        z = _f32c(z)
        e, d = st.num_edges, z.shape[1]
        width = 2 * d + (1 if extra is not None else 0)
        out = torch.empty(e, width, dtype=torch.float32, device=z.device)
        ex = None if extra is None else _f32c(extra)
        with _lib.device_guard(z.device):
            _lib.check(lib.pangnn_edge_gather_concat_f32(z.data_ptr(), z.stride(0), z.shape[0],
                                                         st.edge_index.data_ptr(), e, 0, e, _lib.ptr(ex),
                                                         out.data_ptr(), out.stride(0), d,
                                                         _lib.stream_ptr()), "pangnn_edge_gather_concat_f32")
        ctx.st, ctx.d = st, d
        return out

    @staticmethod
    def backward(ctx, g):
        st, d = ctx.st, ctx.d
        g = _f32c(g)
        assert st.num_src == st.num_nodes, "edge_gather_concat is defined on a whole (square) graph"
        gz = segment_sum_rows(st.by_src, g, 0, d, st.num_nodes)
        segment_sum_rows(st.by_dst, g, d, d, st.num_nodes, out=gz, accumulate=True)
        return gz, None, None


def edge_gather_concat(z, st: EdgeStructure, extra=None):
    return _EdgeGatherConcat.apply(z, st, extra)


class _EdgePairAdd(torch.autograd.Function):
    """h[e] = p[src_e] + q[dst_e] (+ extra[e] * cvec): the decoder's first Linear re-associated onto
    nodes.  Backward = segment sums of g over out-edges (p) and in-edges (q)."""

    @staticmethod
    def forward(ctx, p, q, st: EdgeStructure, extra, cvec):
        lib = _lib.load()
        _lib.require_device(p, q, extra, cvec)
        p, q = _f32c(p), _f32c(q)
        assert p.shape[1] == q.shape[1] and p.stride(0) == q.stride(0)
        e, d = st.num_edges, p.shape[1]
        out = torch.empty(e, d, dtype=torch.float32, device=p.device)
        ex = None if extra is None else _f32c(extra)
        cv = None if cvec is None else _f32c(cvec)
        with _lib.device_guard(p.device):
            _lib.check(lib.pangnn_edge_pair_add_f32(p.data_ptr(), q.data_ptr(), p.stride(0), p.shape[0],
                                                    st.edge_index.data_ptr(), e, 0, e, _lib.ptr(ex),
                                                    _lib.ptr(cv), out.data_ptr(), out.stride(0), d,
                                                    _lib.stream_ptr()), "pangnn_edge_pair_add_f32")
        ctx.st, ctx.d = st, d
        ctx.save_for_backward(ex)
        return out

    @staticmethod
    def backward(ctx, g):
        st, d = ctx.st, ctx.d
        (ex,) = ctx.saved_tensors
        g = _f32c(g)
        gp = segment_sum_rows(st.by_src, g, 0, d, st.num_src) if ctx.needs_input_grad[0] else None
        gq = segment_sum_rows(st.by_dst, g, 0, d, st.num_nodes) if ctx.needs_input_grad[1] else None
        gc = (ex[: g.shape[0]].unsqueeze(0) @ g).squeeze(0) if (ex is not None and ctx.needs_input_grad[4]) else None
        return gp, gq, None, None, gc


def edge_pair_add(p, q, st: EdgeStructure, extra=None, cvec=None):
    return _EdgePairAdd.apply(p, q, st, extra, cvec)


class _SegmentMax(torch.autograd.Function):
    """'max' aggregation of per-edge messages at the target node (convolution.py:7).  Rows without
    in-edges are 0.  The gradient goes to the arg-max edge (first maximum on ties)."""

    @staticmethod
    def forward(ctx, msg, st: EdgeStructure):
        lib = _lib.load()
        _lib.require_device(msg)
        msg = _f32c(msg)
        n, f = st.num_nodes, msg.shape[1]
        d = st.by_dst
        out = torch.empty(n, f, dtype=torch.float32, device=msg.device)
        arg = torch.empty(n, f, dtype=torch.int32, device=msg.device)
        with _lib.device_guard(msg.device):
            _lib.check(lib.pangnn_segment_max_rows_f32(d.rowptr.data_ptr(), _lib.ptr(d.perm), msg.data_ptr(),
                                                       msg.stride(0), out.data_ptr(), arg.data_ptr(),
                                                       out.stride(0), n, f, _lib.stream_ptr()),
                       "pangnn_segment_max_rows_f32")
        ctx.st, ctx.e = st, msg.shape[0]
        ctx.save_for_backward(arg)
        ctx.mark_non_differentiable(arg)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (arg,) = ctx.saved_tensors
        st = ctx.st
        g = _f32c(g)
        n, f = g.shape
        gm = torch.zeros(ctx.e, f, dtype=torch.float32, device=g.device)
        with _lib.device_guard(g.device):
            _lib.check(lib.pangnn_segment_max_bwd_f32(g.data_ptr(), arg.data_ptr(), st.by_dst.rowptr.data_ptr(),
                                                      gm.data_ptr(), gm.stride(0), g.stride(0), n, f,
                                                      _lib.stream_ptr()), "pangnn_segment_max_bwd_f32")
        return gm, None


def segment_max(msg, st: EdgeStructure):
    return _SegmentMax.apply(msg, st)


class _SegmentSum(torch.autograd.Function):
    """'add' aggregation of arbitrary per-edge messages at the target node (generic MessagePassing)."""

    @staticmethod
    def forward(ctx, msg, st: EdgeStructure):
        msg = _f32c(msg)
        ctx.st = st
        return segment_sum_rows(st.by_dst, msg, 0, msg.shape[1], st.num_nodes)

    @staticmethod
    def backward(ctx, g):
        # d out[i] / d msg[e] = [dst_e == i]  =>  gmsg[e] = g[dst_e]: a row gather
        st = ctx.st
        g = _f32c(g)
        cat = _EdgeGatherConcat.apply(g, st, None)      # [E, 2F]: (g[src], g[dst])
        return cat[:, g.shape[1]:], None


def segment_sum(msg, st: EdgeStructure):
    return _SegmentSum.apply(msg, st)


def d16_chunk(num_edges: Optional[int] = None) -> int:
    """tiles per run-sum chunk of the S / T kernels for a list of `num_edges` edges (pangnn_decoder_chunk_tiles_for: 16 for
    E >= 1e6, fewer for short lists so that a mini-batch still spreads over the chip); None: the maximum"""
    lib = _lib.load()
    return int(lib.pangnn_decoder_chunk_tiles() if num_edges is None else lib.pangnn_decoder_chunk_tiles_for(int(num_edges)))


def _sum_parts(plan, part_buf: torch.Tensor, n_rows: int, out: torch.Tensor, accumulate: bool = False,
               row_lo: int = 0) -> torch.Tensor:
    """out[s - row_lo] (+)= sum of the consecutive part rows of source s, s in [row_lo, row_lo + n_rows)
    (pangnn_spmm_csr_f32, idx = NULL; the row pointer's entries are absolute part positions, so a window of rows is the same
    call on the shifted pointer)"""
    lib = _lib.load()
    if n_rows <= 0:
        return out
    with _lib.device_guard(part_buf.device):
        _lib.check(lib.pangnn_spmm_csr_f32(plan.part_rowptr.data_ptr() + 8 * int(row_lo), None, None, _lib.ptr(part_buf),
                                           part_buf.stride(0), part_buf.shape[0], None, out.data_ptr(),
                                           out.stride(0), n_rows, part_buf.shape[0], part_buf.shape[1], int(accumulate),
                                           _lib.stream_ptr()), "pangnn_spmm_csr_f32(parts)")
    return out


def _dgrad_sum(rec, st: EdgeStructure, by: Optional[str], w2, w3, n_rows: int = 0, out=None, g_b2=None, live=None,
               accumulate: bool = False):
    """dL/dh1 summed over the rows of CSR order `by` ("src" / "dst") from the per-edge records of the training
    kernel: pangnn_decoder_dgrad_f32 (run parts) + the short contiguous part sum.  `g_b2`: also filled with dL/db2
    (one call per step asks for it); by = None: the parameter sums alone."""
    lib = _lib.load()
    dev = rec.device
    plan = st.csr_plan(by, d16_chunk(st.num_edges)) if by else None
    csr = None if not by else (st.by_dst if by == "dst" else st.by_src)
    parts = None if plan is None else torch.empty(plan.n_parts, 64, dtype=torch.float32, device=dev)
    with _lib.device_guard(dev):
        ws_bytes = lib.pangnn_decoder_dgrad_workspace_bytes() if g_b2 is not None else 0
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
        ev = _timer_start("dec.dgrad")
        _lib.check(lib.pangnn_decoder_dgrad_f32(rec.data_ptr(), None if csr is None else _lib.ptr(csr.perm),
                                                None if plan is None else plan.keys.data_ptr(),
                                                w2.data_ptr(), w3.data_ptr(), st.num_edges, _lib.ptr(parts),
                                                None if plan is None else plan.part_off.data_ptr(),
                                                _lib.ptr(g_b2), _lib.ptr(live), _lib.ptr(ws), ws_bytes,
                                                _lib.stream_ptr()),
                   "pangnn_decoder_dgrad_f32")
        _timer_stop("dec.dgrad", ev)
    if plan is None:
        return None
    if out is None:
        out, accumulate = torch.empty(n_rows, 64, device=dev), False
    return _sum_parts(plan, parts, n_rows, out, accumulate)


def _decoder_train16(p, q, st: EdgeStructure, ex, cv, w2, b2, w3, b3, y=None, pw=None, denom=0, g_logits=None,
                     out_p=None, out_q=None, need_p=True, need_q=True, after_p=None, live=None, accumulate_q=False,
                     p_windows=None, out_logits=None):
    """Two-wave-per-SIMD training decoder (csrc/decoder16.hip).  One pass over the edges in the caller's order (S):
    logits, loss (y given) or the given dL/dlogits, every parameter gradient, per-source run sums when the list is
    source-sorted, and a 32-byte record per edge; then dL/dQ (and dL/dP for unsorted lists) from the records in CSR
    order (T).  No [E, 64] tensor exists.  Returns (loss, logits, gp, gq, g_cv, g_w2, g_b2, g_w3, g_b3).
    `after_p(gp)` is called once dL/dP is enqueued and before the T pass (the partitioned model starts the return
    exchange of the halo rows' gradients there, so that it runs under T).
    `live` (device int64[1], optional): the list is a fixed-shape padded batch whose first live[0] edges are real
    (include/pangnn_hip.h, live_edges): the padding enters no sum and the fused loss is the mean over live[0] edges.
    `accumulate_q` (with out_q): dL/dQ is ADDED to out_q (a second edge range of the same targets: the partitioned decoder).
    `p_windows` (source-sorted lists only): [(row_lo, row_hi, out [row_hi - row_lo, 64]), ...] — dL/dP is wanted for these row
    ranges only, each written to its own tensor (an edge range whose sources lie in known row ranges: nothing is written for
    the rows in between); the returned gp is then the list of those tensors.  `out_logits`: where the fused-loss logits go."""
    lib = _lib.load()
    dev = p.device
    e, d = st.num_edges, p.shape[1]
    fused = y is not None
    logits = (out_logits if out_logits is not None else torch.empty(e, dtype=torch.float32, device=dev)) if fused else None
    loss = torch.empty(1, dtype=torch.float32, device=dev) if fused else None
    g_w2 = torch.empty_like(w2)
    g_b2, g_w3, g_b3 = torch.empty_like(b2), torch.empty_like(w3), torch.empty_like(b3)
    g_cv = None if cv is None else torch.empty_like(cv)
    rec = torch.empty(max(e, 1), 8, dtype=torch.int32, device=dev)
    plan = st.runsum_plan(d16_chunk(e)) if (need_p and e > 0) else None
    parts = None if plan is None else torch.empty(plan.n_parts, d, dtype=torch.float32, device=dev)
    with _lib.device_guard(dev):
        ws_bytes = lib.pangnn_decoder_train_workspace_bytes()
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        ev = _timer_start("dec.bwd")
        if p.dtype != q.dtype:
            raise ValueError("p and q must be stored alike (both float32, both bfloat16 or both float16)")
        _lib.check(lib.pangnn_decoder_train_mixed(
            p.data_ptr(), p.stride(0), q.data_ptr(), q.stride(0), _dt(p), max(p.shape[0], q.shape[0]),
            st.edge_index.data_ptr(), e, e, _lib.ptr(ex), _lib.ptr(cv), w2.data_ptr(), b2.data_ptr(), w3.data_ptr(),
            b3.data_ptr(), d, _lib.ptr(y), _lib.ptr(pw), int(denom), _lib.ptr(g_logits), _lib.ptr(logits),
            _lib.ptr(loss), rec.data_ptr(), _lib.ptr(parts), None if plan is None else plan.part_off.data_ptr(),
            g_w2.data_ptr(), g_w3.data_ptr(), g_b3.data_ptr(), _lib.ptr(g_cv), _lib.ptr(live), ws.data_ptr(),
            ws_bytes, _lib.stream_ptr()), "pangnn_decoder_train_mixed")
        _timer_stop("dec.bwd", ev)
    gp = gq = None
    b2_out = g_b2          # dL/db2 comes out of exactly one dgrad call
    if need_p:
        if e == 0 and p_windows is not None:
            gp = [o.zero_() for _, _, o in p_windows]
        elif e == 0:
            gp = (out_p if out_p is not None else torch.empty(p.shape[0], d, device=dev)).zero_()
        elif plan is not None and p_windows is not None:
            gp = [_sum_parts(plan, parts, hi - lo, o, row_lo=lo) for lo, hi, o in p_windows]
        elif plan is not None:
            gp = _sum_parts(plan, parts, p.shape[0], out_p if out_p is not None else torch.empty(p.shape[0], d, device=dev))
        else:
            gp = _dgrad_sum(rec, st, "src", w2, w3, p.shape[0], out_p, g_b2=b2_out, live=live)
            b2_out = None
            if p_windows is not None:                       # an unsorted list: all rows were summed, hand out the windows
                gp = [o.copy_(gp[lo:hi]) for lo, hi, o in p_windows]
        if after_p is not None:
            after_p(gp)
    if need_q:
        acc_q = bool(accumulate_q) and out_q is not None
        if e == 0:
            gq = out_q if acc_q else (out_q if out_q is not None else torch.empty(q.shape[0], d, device=dev)).zero_()
        else:
            gq = _dgrad_sum(rec, st, "dst", w2, w3, q.shape[0], out_q, g_b2=b2_out, live=live, accumulate=acc_q)
            b2_out = None
    if b2_out is not None:
        if e == 0:
            g_b2.zero_()
        else:
            _dgrad_sum(rec, st, None, w2, w3, g_b2=b2_out, live=live)
    return loss, logits, gp, gq, g_cv, g_w2, g_b2, g_w3, g_b3


def _rows_f32(t: torch.Tensor) -> torch.Tensor:
    """fp32 with unit column stride and a 16-byte friendly row stride; column windows of a wider
    row-major matrix pass through without a copy"""
    if t.dtype != torch.float32:
        t = t.float()
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.stride(0) >= t.shape[1] \
            and t.data_ptr() % 16 == 0:
        return t
    return t.contiguous()


def _rows_any(t: torch.Tensor) -> torch.Tensor:
    """_rows_f32 that lets bfloat16 / float16 storage through (the *_mixed entry points read it as stored)"""
    if t.dtype not in ROWS16:
        return _rows_f32(t)
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.stride(0) >= t.shape[1] \
            and t.data_ptr() % 16 == 0:
        return t
    return t.contiguous()


def autocast_rows_dtype(t: torch.Tensor):
    """torch.bfloat16 / torch.float16 when that mixed precision is on for `t`'s device (accelerate's autocast, pangnn.py:25,
    src/setup.py:50), else None: the type the reference's autocast Linear outputs — and hence the rows PyG's propagate
    gathers — are stored in"""
    if t.is_cuda and torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype("cuda")
        return dt if dt in ROWS16 else None
    return None


def _rows_dec(t: torch.Tensor) -> torch.Tensor:
    """decoder tables as the decoder kernels read them: float32, or bfloat16 / float16 as stored"""
    return _rows_any(t)


def autocast_bf16(t: torch.Tensor) -> bool:
    """bf16 mixed precision is on for `t`'s device (config 5)"""
    return autocast_rows_dtype(t) == torch.bfloat16


def dtype_code(dtype) -> int:
    """PANGNN_DTYPE_* of a torch dtype (None / anything that is not a 2-byte row format: f32)"""
    return 1 if dtype == torch.bfloat16 else 2 if dtype == torch.float16 else 0


def _dt(t: torch.Tensor) -> int:
    """PANGNN_DTYPE_* of a tensor's storage"""
    return dtype_code(t.dtype)


class _DecoderMLP(torch.autograd.Function):
    """Fused link decoder (node_dim 64): logits[e] = w3 . relu(W2 relu(p[src]+q[dst] (+w_e c)) + b2) + b3.
    Forward keeps every [E, 64] intermediate on chip; backward recomputes per tile, emits
    dL/dh1pre [E,64] once and segment-sums it by source (-> g_p) and by target (-> g_q).
    `pq_joint=True`: p and q are the two column halves of one [N, 128] matrix and the gradient comes
    back as one [N, 128] matrix (no padding / adding of two partial gradients)."""

    @staticmethod
    def forward(ctx, p, q, st: EdgeStructure, extra, cvec, w2, b2, w3, b3, pq_joint=False):
        lib = _lib.load()
        _lib.require_device(p, q, extra, cvec, w2, b2, w3, b3)
        rows = _rows_dec if DECODER_PRECISION == 1 else _rows_f32      # bf16-stored tables: gathered as stored
        if pq_joint:
            pq = rows(p)
            d = pq.shape[1] // 2
            p, q = pq[:, :d], pq[:, d:]
        else:
            p, q = rows(p), rows(q)
            if p.dtype != q.dtype:
                p, q = p.float(), q.float()
        w2, b2, w3, b3 = (_f32c(t) for t in (w2, b2, w3, b3))
        ex = None if extra is None else _f32c(extra)
        cv = None if cvec is None else _f32c(cvec)
        e, d = st.num_edges, p.shape[1]
        logits = torch.empty(e, dtype=torch.float32, device=p.device)
        with _lib.device_guard(p.device):
            ev = _timer_start("dec.fwd")
            if p.dtype in ROWS16:
                _lib.check(lib.pangnn_decoder_mlp_infer_mixed(p.data_ptr(), p.stride(0), q.data_ptr(), q.stride(0), _dt(p),
                                                              max(p.shape[0], q.shape[0]), st.edge_index.data_ptr(), e,
                                                              e, _lib.ptr(ex), _lib.ptr(cv), w2.data_ptr(),
                                                              b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), d,
                                                              _lib.ptr(logits), _lib.stream_ptr()),
                           "pangnn_decoder_mlp_infer_mixed")
            else:
                _lib.check(lib.pangnn_decoder_mlp_infer_f32(p.data_ptr(), p.stride(0), q.data_ptr(), q.stride(0),
                                                            max(p.shape[0], q.shape[0]), st.edge_index.data_ptr(), e, e,
                                                            _lib.ptr(ex), _lib.ptr(cv), w2.data_ptr(), b2.data_ptr(),
                                                            w3.data_ptr(), b3.data_ptr(), d, _lib.ptr(logits),
                                                            DECODER_PRECISION, _lib.stream_ptr()),
                           "pangnn_decoder_mlp_infer_f32")
            _timer_stop("dec.fwd", ev)
        ctx.st, ctx.joint = st, pq_joint
        ctx.save_for_backward(p, q, ex, cv, w2, b2, w3, b3)
        return logits

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        st = ctx.st
        p, q, ex, cv, w2, b2, w3, b3 = ctx.saved_tensors
        g = _f32c(g)
        e, d = st.num_edges, p.shape[1]
        dev = p.device
        if DECODER_PRECISION == 1:
            if ctx.joint:
                g_pq = torch.empty(p.shape[0], 2 * d, dtype=torch.float32, device=dev)
                _, _, _, _, g_cv, g_w2, g_b2, g_w3, g_b3 = _decoder_train16(
                    p, q, st, ex, cv, w2, b2, w3, b3, g_logits=g, out_p=g_pq[:, :d], out_q=g_pq[:, d:])
                return g_pq, None, None, None, g_cv, g_w2, g_b2, g_w3, g_b3, None
            _, _, gp, gq, g_cv, g_w2, g_b2, g_w3, g_b3 = _decoder_train16(
                p, q, st, ex, cv, w2, b2, w3, b3, g_logits=g, need_p=ctx.needs_input_grad[0],
                need_q=ctx.needs_input_grad[1])
            return gp, gq, None, None, g_cv, g_w2, g_b2, g_w3, g_b3, None
        g_h1 = torch.empty(e, d, dtype=torch.float32, device=dev)
        g_w2 = torch.empty_like(w2)
        g_b2, g_w3, g_b3 = torch.empty_like(b2), torch.empty_like(w3), torch.empty_like(b3)
        g_cv = None if cv is None else torch.empty_like(cv)
        plan = st.runsum_plan()
        parts = None if plan is None else torch.empty(plan.n_parts, d, dtype=torch.float32, device=dev)
        with _lib.device_guard(dev):
            ws_bytes = lib.pangnn_decoder_mlp_bwd_workspace_bytes(e)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            ev = _timer_start("dec.bwd")
            _lib.check(lib.pangnn_decoder_mlp_bwd_f32(p.data_ptr(), p.stride(0), q.data_ptr(), q.stride(0),
                                                      max(p.shape[0], q.shape[0]), st.edge_index.data_ptr(), e, e,
                                                      _lib.ptr(ex), _lib.ptr(cv), w2.data_ptr(), b2.data_ptr(),
                                                      w3.data_ptr(), b3.data_ptr(), d, _lib.ptr(g), _lib.ptr(g_h1),
                                                      g_w2.data_ptr(), g_b2.data_ptr(), g_w3.data_ptr(),
                                                      g_b3.data_ptr(), _lib.ptr(g_cv), _lib.ptr(parts),
                                                      None if plan is None else plan.part_off.data_ptr(),
                                                      DECODER_PRECISION, ws.data_ptr(), ws_bytes,
                                                      _lib.stream_ptr()), "pangnn_decoder_mlp_bwd_f32")
            _timer_stop("dec.bwd", ev)

        def by_source(out=None):
            if plan is not None:
                return _sum_parts(plan, parts, p.shape[0],
                                  out if out is not None else torch.empty(p.shape[0], d, device=dev))
            return segment_sum_rows(st.by_src, g_h1, 0, d, p.shape[0], out=out)

        if ctx.joint:
            g_pq = torch.empty(p.shape[0], 2 * d, dtype=torch.float32, device=dev)
            by_source(g_pq[:, :d])
            segment_sum_rows(st.by_dst, g_h1, 0, d, p.shape[0], out=g_pq[:, d:])
            return g_pq, None, None, None, g_cv, g_w2, g_b2, g_w3, g_b3, None
        gp = by_source() if ctx.needs_input_grad[0] else None
        gq = segment_sum_rows(st.by_dst, g_h1, 0, d, q.shape[0]) if ctx.needs_input_grad[1] else None
        return gp, gq, None, None, g_cv, g_w2, g_b2, g_w3, g_b3, None


def _timed_decoder():
    """a kernel-timer tag when bench.py asked for per-launch times of the decoder kernels (KERNEL_TIMER holds their tags): the
    event pairs are recorded by the ctypes route (same kernels, same order), so such a call takes it — like a tagged propagate"""
    t = KERNEL_TIMER
    return "dec" if (t is not None and ("dec.bwd" in t or "dec.fwd" in t or "dec.dgrad" in t)) else None


def decoder_mlp(p, q, st: EdgeStructure, extra, cvec, w2, b2, w3, b3):
    return _DecoderMLP.apply(p, q, st, extra, cvec, w2, b2, w3, b3, False)


def decoder_mlp_pq(pq, st: EdgeStructure, extra, cvec, w2, b2, w3, b3):
    """p = pq[:, :D], q = pq[:, D:] (one [N, 2D] node-level product)"""
    _lib.require_device(pq, extra, cvec, w2, b2, w3, b3)
    if _via_ops(st, _timed_decoder()) and DECODER_PRECISION == 1:
        from . import torch_ops
        return torch_ops.decoder_mlp_pq(pq, st, extra, cvec, w2, b2, w3, b3)
    return _DecoderMLP.apply(pq, None, st, extra, cvec, w2, b2, w3, b3, True)


# The training decoder computes every gradient in its forward pass; its backward only has to scale them by the
# upstream gradient of the loss, which for `loss.backward()` is 1 — but a device tensor cannot be compared with 1
# without a sync.  A training loop that owns the call passes this cached tensor as the root gradient instead
# (`loss.backward(unit_grad(device))`), recognised by identity: no [N,128] multiply per step.
_UNIT_GRAD = {}


def unit_grad(device) -> torch.Tensor:
    key = torch.device(device)
    t = _UNIT_GRAD.get(key)
    if t is None:
        t = _UNIT_GRAD[key] = torch.ones((), dtype=torch.float32, device=key)
        if key.type == "cuda":             # the C++ autograd formula of pangnn::decoder_loss recognises it by address too
            from . import torch_ops
            torch_ops.ops._set_unit_grad(t)
    return t


def is_unit_grad(g: torch.Tensor) -> bool:
    if type(g) is not torch.Tensor:          # a FakeTensor / FunctionalTensor while a compiler traces the backward formula
        return False
    t = _UNIT_GRAD.get(g.device)
    return t is not None and g.data_ptr() == t.data_ptr() and g.dim() == 0


def scale_by_loss_grad_(ctx, tensors, go):
    """The gradients the one-pass training decoder left in memory, times the upstream gradient `go` of its loss.
    `loss.backward()` and accelerate's `(loss / gradient_accumulation_steps).backward()` (pangnn.py:207) hand over EXACTLY 1,
    which the host cannot know without a synchronisation: pangnn_scale_unless_one_f32 reads the device scalar and returns
    when it is 1.0, so the usual step costs one launch and no pass over the [N, 128] gradient; any other value (a
    GradScaler's scale under fp16, gradient accumulation) is applied in place.  In place means the stored gradients are
    consumed: a second backward through the same node (retain_graph=True) raises instead of scaling twice."""
    if type(go) is not torch.Tensor or not go.is_cuda:          # a FakeTensor while a compiler traces the formula
        return [None if t is None else t * go for t in tensors]
    if getattr(ctx, "_pangnn_scaled", False):
        raise RuntimeError("pangnn_amd: the fused decoder loss computes its gradients in its forward pass and hands them over "
                           "once; backward through it a second time is not supported (call the model again)")
    ctx._pangnn_scaled = True
    live = [t for t in tensors if t is not None and t.numel() > 0]
    for t in live:
        if t.dtype != torch.float32 or not t.is_contiguous():
            return [None if u is None else u * go for u in tensors]
    lib = _lib.load()
    n = len(live)
    if n:
        import ctypes as C
        g32 = go if go.dtype == torch.float32 else go.float()
        with _lib.device_guard(g32.device):
            _lib.check(lib.pangnn_scale_unless_one_f32((C.c_void_p * n)(*[t.data_ptr() for t in live]),
                                                       (C.c_int64 * n)(*[t.numel() for t in live]), n,
                                                       g32.data_ptr(), _lib.stream_ptr()), "pangnn_scale_unless_one_f32")
    return list(tensors)


class _DecoderLoss(torch.autograd.Function):
    """Training form of the fused decoder: mean BCEWithLogits(pos_weight) loss, logits and ALL gradients in
    one pass over the edges (pangnn_decoder_mlp_loss_f32).  Everything is computed in forward(); backward()
    only scales the stored gradients by the upstream gradient of the loss."""

    @staticmethod
    def forward(ctx, p, q, st: EdgeStructure, extra, cvec, w2, b2, w3, b3, y, pos_weight, denom, pq_joint, live=None):
        lib = _lib.load()
        _lib.require_device(p, q, extra, cvec, w2, b2, w3, b3, y, pos_weight)
        if live is not None and (DECODER_PRECISION != 1 or live.dtype != torch.int64 or not live.is_cuda):
            raise ValueError("live= (a padded fixed-shape batch) needs the default decoder mode and a device int64 tensor")
        rows = _rows_dec if DECODER_PRECISION == 1 else _rows_f32      # bf16-stored tables: gathered as stored
        if pq_joint:
            pq = rows(p)
            d = pq.shape[1] // 2
            p, q = pq[:, :d], pq[:, d:]
        else:
            p, q = rows(p), rows(q)
            if p.dtype != q.dtype:
                p, q = p.float(), q.float()
        w2, b2, w3, b3, y = (_f32c(t) for t in (w2, b2, w3, b3, y))
        ex = None if extra is None else _f32c(extra)
        cv = None if cvec is None else _f32c(cvec)
        pw = None if pos_weight is None else _f32c(pos_weight).reshape(-1)
        e, d = st.num_edges, p.shape[1]
        dev = p.device
        if DECODER_PRECISION == 1:
            if pq_joint:
                g_pq = torch.empty(p.shape[0], 2 * d, dtype=torch.float32, device=dev)
                loss, logits, _, _, g_cv, g_w2, g_b2, g_w3, g_b3 = _decoder_train16(
                    p, q, st, ex, cv, w2, b2, w3, b3, y=y, pw=pw, denom=denom, out_p=g_pq[:, :d], out_q=g_pq[:, d:],
                    live=live)
                gp, gq = g_pq, None
            else:
                loss, logits, gp, gq, g_cv, g_w2, g_b2, g_w3, g_b3 = _decoder_train16(
                    p, q, st, ex, cv, w2, b2, w3, b3, y=y, pw=pw, denom=denom, live=live)
            ctx.has_cv, ctx.has_q = g_cv is not None, gq is not None
            ctx.save_for_backward(gp, gq if gq is not None else gp.new_empty(0),
                                  g_cv if g_cv is not None else gp.new_empty(0), g_w2, g_b2, g_w3, g_b3)
            ctx.mark_non_differentiable(logits)
            return loss.view(()), logits
        logits = torch.empty(e, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        g_h1 = torch.empty(e, d, dtype=torch.float32, device=dev)
        g_w2 = torch.empty_like(w2)
        g_b2, g_w3, g_b3 = torch.empty_like(b2), torch.empty_like(w3), torch.empty_like(b3)
        g_cv = None if cv is None else torch.empty_like(cv)
        plan = st.runsum_plan()
        parts = None if plan is None else torch.empty(plan.n_parts, d, dtype=torch.float32, device=dev)
        with _lib.device_guard(dev):
            ws_bytes = lib.pangnn_decoder_mlp_bwd_workspace_bytes(e)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            ev = _timer_start("dec.bwd")
            _lib.check(lib.pangnn_decoder_mlp_loss_f32(
                p.data_ptr(), p.stride(0), q.data_ptr(), q.stride(0), max(p.shape[0], q.shape[0]),
                st.edge_index.data_ptr(), e, e, _lib.ptr(ex), _lib.ptr(cv), w2.data_ptr(), b2.data_ptr(),
                w3.data_ptr(), b3.data_ptr(), d, _lib.ptr(y), _lib.ptr(pw), int(denom), _lib.ptr(logits),
                loss.data_ptr(), _lib.ptr(g_h1), g_w2.data_ptr(), g_b2.data_ptr(), g_w3.data_ptr(), g_b3.data_ptr(),
                _lib.ptr(g_cv), _lib.ptr(parts), None if plan is None else plan.part_off.data_ptr(),
                DECODER_PRECISION, ws.data_ptr(), ws_bytes, _lib.stream_ptr()), "pangnn_decoder_mlp_loss_f32")
            _timer_stop("dec.bwd", ev)

        def by_source(out=None):
            if plan is not None:       # per-run partial rows came out of the kernel: short contiguous sum
                return _sum_parts(plan, parts, p.shape[0],
                                  out if out is not None else torch.empty(p.shape[0], d, device=dev))
            return segment_sum_rows(st.by_src, g_h1, 0, d, p.shape[0], out=out)

        if pq_joint:
            g_pq = torch.empty(p.shape[0], 2 * d, dtype=torch.float32, device=dev)
            by_source(g_pq[:, :d])
            segment_sum_rows(st.by_dst, g_h1, 0, d, p.shape[0], out=g_pq[:, d:])
            gp, gq = g_pq, None
        else:
            gp = by_source()
            gq = segment_sum_rows(st.by_dst, g_h1, 0, d, q.shape[0])
        del g_h1
        ctx.has_cv, ctx.has_q = g_cv is not None, gq is not None
        ctx.save_for_backward(gp, gq if gq is not None else gp.new_empty(0),
                              g_cv if g_cv is not None else gp.new_empty(0), g_w2, g_b2, g_w3, g_b3)
        ctx.mark_non_differentiable(logits)
        return loss.view(()), logits

    @staticmethod
    def backward(ctx, go, _go_logits):
        gp, gq, g_cv, g_w2, g_b2, g_w3, g_b3 = ctx.saved_tensors
        if is_unit_grad(go):         # `loss.backward(unit_grad(device))` (train.train_step): nothing to scale
            return (gp, gq if ctx.has_q else None, None, None, g_cv if ctx.has_cv else None,
                    g_w2, g_b2, g_w3, g_b3, None, None, None, None, None)
        gp, gq, g_cv, g_w2, g_b2, g_w3, g_b3 = scale_by_loss_grad_(
            ctx, [gp, gq if ctx.has_q else None, g_cv if ctx.has_cv else None, g_w2, g_b2, g_w3, g_b3], go)
        return (gp, gq, None, None, g_cv, g_w2, g_b2, g_w3, g_b3, None, None, None, None, None)


def decoder_loss(p, q, st, extra, cvec, w2, b2, w3, b3, y, pos_weight, denom):
    return _DecoderLoss.apply(p, q, st, extra, cvec, w2, b2, w3, b3, y, pos_weight, denom, False, None)


def decoder_loss_pq(pq, st, extra, cvec, w2, b2, w3, b3, y, pos_weight, denom, live=None):
    """`live` (device int64[1]): a padded fixed-shape batch whose first live[0] edges are real (train.ReplayedFreshStep):
    the loss is their mean and the padding contributes to no gradient"""
    _lib.require_device(pq, extra, cvec, w2, b2, w3, b3, y, pos_weight, live)
    if _via_ops(st, _timed_decoder()) and DECODER_PRECISION == 1:
        from . import torch_ops
        return torch_ops.decoder_loss_pq(pq, st, extra, cvec, w2, b2, w3, b3, y, pos_weight, denom, live)
    return _DecoderLoss.apply(pq, None, st, extra, cvec, w2, b2, w3, b3, y, pos_weight, denom, True, live)


@torch.compiler.assume_constant_result
def _linear_supported(k: int, m: int) -> bool:
    """pangnn_linear_supported(k, m, wgrad=1); a constant of the shapes, baked in when torch.compile traces"""
    return bool(_lib.load().pangnn_linear_supported(k, m, 1))


def linear(x, w, bias=None, in_act: int = 0, out_dtype=None):
    """torch.nn.functional.linear for node-level layers; shapes the HIP kernels do not cover
    (K or M outside {64,128}) go to hipBLASLt via torch.
    in_act = 1: linear(ELU(x), w, bias) with the activation folded into the kernels — forward: ELU applied to the rows on their
    way into LDS; backward: dL/dx comes out already multiplied by ELU'(x), the weight gradient re-applies ELU (no activation
    kernel, no activated tensor in HBM; pangnn::linear / linear_backward, csrc/torch_ops.cpp).
    `x` may be stored as bfloat16 / float16 and `out_dtype=torch.bfloat16` / `torch.float16` stores the result in that type
    (the autocast Linear outputs of config 5 / of `--mixed_precision fp16`): fp32 products and sums either way, one rounding
    on store; gradients of 16-bit tensors are stored alike."""
    _lib.require_device(x, w, bias)                  # no CPU path: the torch branch below is hipBLASLt on the GPU
    k, m = w.shape[1], w.shape[0]
    if x.dim() == 2 and _linear_supported(int(k), int(m)) and (out_dtype in (None, torch.float32) or out_dtype in ROWS16) \
            and not (x.dtype in ROWS16 and out_dtype in ROWS16 and x.dtype != out_dtype):
        # ONE route (round 4): the dispatcher op, whose HIP implementation and autograd formula are C++
        # (csrc/torch_ops.cpp) — the ctypes autograd.Function twin of rounds 1-3 is gone
        from . import torch_ops
        return torch_ops.ops.linear(x, w, bias, int(in_act), dtype_code(out_dtype))
    if in_act:
        x = torch.nn.functional.elu(x)
    y = torch.nn.functional.linear(x.float(), w, bias)
    return y if out_dtype is None else y.to(out_dtype)


def bce_with_logits(logits, y, pos_weight=None, denom=None):
    """mean BCEWithLogitsLoss(pos_weight) over `denom` edges (default: all of them; a partitioned shard passes the job's
    edge count), loss and dL/dlogits from one pass: pangnn::bce_with_logits — implementation and autograd formula in C++
    (csrc/torch_ops.cpp), the only route"""
    _lib.require_device(logits, y, pos_weight)
    from . import torch_ops
    return torch_ops.ops.bce_with_logits(logits, y, pos_weight, int(logits.shape[0] if denom is None else denom))[0]



def _node_actions(x_tab, st, norm):
    """(r, s) = (A_hat x, A_hat 1): the two node vectors through which a scalar-feature embedding acts after one
    propagate.  Computed once per (graph, normalisation, feature tensor) with the real propagate kernel on a 16-column
    table and cached on the norm object (like the normalisation itself: static for a static graph)."""
    xv = x_tab.detach().to(torch.float32).reshape(-1)
    cache = norm.__dict__.setdefault("_node_actions", {})
    key = (x_tab.data_ptr(), x_tab._version, tuple(x_tab.shape))
    if key not in cache:
        # ONE launch (pangnn_node_actions_f32: a wave per row of the by-target CSR) — a fresh mini-batch pays this every step
        # (rounds 2-4: the generic propagate on a 16-column table (x, 1, 0, ...): table build + propagate + a transposing copy)
        lib = _lib.load()
        xv = xv.contiguous()
        rs = torch.empty(2, st.num_nodes, dtype=torch.float32, device=xv.device)
        csr, val = st.by_dst, norm.by_dst
        _lib.require_device(xv, csr.rowptr, val)
        with _lib.device_guard(xv.device):
            _lib.check(lib.pangnn_node_actions_f32(csr.rowptr.data_ptr(), _lib.ptr(csr.other), _lib.ptr(val), xv.data_ptr(),
                                                   st.num_nodes, rs[0].data_ptr(), rs[1].data_ptr(), _lib.stream_ptr()),
                       "pangnn_node_actions_f32")
        cache.clear()
        cache[key] = (rs[0], rs[1], x_tab)                                   # x_tab kept alive: key is its address
    r, s, _ = cache[key]
    return r, s


class _EmbedPropagate(torch.autograd.Function):
    """agg = A_hat (x w^T + 1 b^T): the scalar-feature embedding (nn.Linear(1, D), gnn.py:97) followed by the
    first GCN layer's propagate (gnn.py:158; GCNConv with in < out propagates before its dense layer).
    `x_tab` [n_src] holds the scalar feature of every row the edges read (on a partitioned graph: own + halo
    rows, exchanged once since x never changes — this layer then needs no per-step exchange at all).
    Forward runs the real propagate kernel on h0.  Backward needs only the two parameter gradients, which by
    linearity are (A_hat x)^T g and (A_hat 1)^T g; the transposed propagate is skipped exactly as autograd
    skips gradients of inputs that require none.  A_hat x and A_hat 1 are cached on the norm object.
    (Round 2's form, `fuse_embedding="propagate"`; the default is _EmbedConvIn below.)"""

    @staticmethod
    def forward(ctx, x_tab, w, b, st, norm, tag):
        _lib.require_device(x_tab, w, b)
        xv = x_tab.detach().to(torch.float32).reshape(-1)
        h0 = torch.addcmul(b.detach().reshape(1, -1), xv.unsqueeze(1), w.detach().reshape(1, -1))
        agg = spmm_csr(st.by_dst, norm.by_dst, h0, st.num_nodes, tag=None if tag is None else tag + ".fwd")
        r, s = _node_actions(x_tab, st, norm)
        ctx.save_for_backward(r, s)
        ctx.d = w.shape[0]
        return agg

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        r, s = ctx.saved_tensors
        g = _rows_f32(g)
        n, f = g.shape
        out = torch.empty(2, f, dtype=torch.float32, device=g.device)
        with _lib.device_guard(g.device):
            ws_bytes = lib.pangnn_weighted_colsum_workspace_bytes(f)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=g.device)
            _lib.check(lib.pangnn_weighted_colsum_f32(g.data_ptr(), g.stride(0), r.data_ptr(), s.data_ptr(), n, f,
                                                      out.data_ptr(), ws.data_ptr(), ws_bytes, _lib.stream_ptr()),
                       "pangnn_weighted_colsum_f32")
        return None, out[0].reshape(ctx.d, 1), out[1], None, None, None


def embed_propagate(x_tab, w, b, st, norm, tag=None):
    _lib.require_device(x_tab, w, b)
    if _via_ops(st, tag) and getattr(norm, "weight_ref", norm) is not norm:
        from . import torch_ops
        return torch_ops.embed_propagate(x_tab, w, b, st, norm)
    return _EmbedPropagate.apply(x_tab, w, b, st, norm, tag)


class _EmbedConvIn(torch.autograd.Function):
    """conv_in(embedding(x)) for the scalar-feature model (gnn.py:97,125 + GCNConv at :131 / :146 / :158) by linearity:

        A_hat (x w^T + 1 b^T) W^T + b_in  =  r a^T + s c^T + b_in,     r = A_hat x, s = A_hat 1, a = W w, c = W b.

    r and s are node vectors computed once per graph by the propagate kernel (`_node_actions`, cached like gcn_norm);
    per step the layer is ONE launch that writes the [N, H] rows (pangnn_embed_conv_in_rows: a, c formed in the kernel) —
    no propagate, no dense product — and its backward ONE pass over dL/dout ([r s 1]^T g) plus a one-workgroup kernel
    forming dL/dW = (r^T g) w^T + (s^T g) b^T, dL/dw = W^T (r^T g), dL/db = W^T (s^T g), dL/db_in = 1^T g
    (pangnn_embed_conv_in_grads).
    The propagate and the dense layer commute, so this is also GCNConv's linear-then-propagate order (in >= out).
    Exact algebra (no approximation): the result differs from the layer-by-layer evaluation by fp32 re-association
    only.  `x_tab` as in _EmbedPropagate (own + halo rows on a shard)."""

    @staticmethod
    def forward(ctx, x_tab, w, b, w_in, b_in, st, norm, out_dtype):
        lib = _lib.load()
        _lib.require_device(x_tab, w, b, w_in, b_in)
        r, s = _node_actions(x_tab, st, norm)
        wv, bv, win = _f32c(w.detach().reshape(-1)), _f32c(b.detach().reshape(-1)), _f32c(w_in.detach())
        n, h, d = st.num_nodes, win.shape[0], win.shape[1]
        bias = None if b_in is None else _f32c(b_in.detach())
        out = torch.empty(n, h, dtype=out_dtype or torch.float32, device=r.device)
        with _lib.device_guard(r.device):
            # a = W w and c = W b are formed inside the kernel: the whole layer is this one launch
            _lib.check(lib.pangnn_embed_conv_in_rows(r.data_ptr(), s.data_ptr(), wv.data_ptr(), bv.data_ptr(), win.data_ptr(),
                                                     _lib.ptr(bias), d, out.data_ptr(), _dt(out), out.stride(0), n, h,
                                                     _lib.stream_ptr()), "pangnn_embed_conv_in_rows")
        ctx.save_for_backward(r, s, wv, bv, win)
        ctx.has_bias = b_in is not None
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        r, s, wv, bv, win = ctx.saved_tensors
        g = _rows_any(g)
        n, h = g.shape
        d = win.shape[1]
        dev = g.device
        g_w, g_b = torch.empty(d, 1, dtype=torch.float32, device=dev), torch.empty(d, dtype=torch.float32, device=dev)
        g_win = torch.empty(h, d, dtype=torch.float32, device=dev)
        g_bin = torch.empty(h, dtype=torch.float32, device=dev) if ctx.has_bias else None
        with _lib.device_guard(dev):
            ws_bytes = lib.pangnn_embed_conv_in_grads_workspace_bytes(h)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.pangnn_embed_conv_in_grads(g.data_ptr(), _dt(g), g.stride(0), r.data_ptr(), s.data_ptr(), n,
                                                      wv.data_ptr(), bv.data_ptr(), win.data_ptr(), d, h, g_w.data_ptr(),
                                                      g_b.data_ptr(), g_win.data_ptr(), _lib.ptr(g_bin), ws.data_ptr(), ws_bytes,
                                                      _lib.stream_ptr()), "pangnn_embed_conv_in_grads")
        return None, g_w, g_b, g_win, g_bin, None, None, None


def embed_conv_in(x_tab, w, b, w_in, b_in, st, norm, out_dtype=None):
    _lib.require_device(x_tab, w, b, w_in, b_in)
    if _via_ops(st) and getattr(norm, "weight_ref", norm) is not norm and (out_dtype in (None, torch.float32) or out_dtype in ROWS16):
        from . import torch_ops
        return torch_ops.embed_conv_in(x_tab, w, b, w_in, b_in, st, norm, out_dtype)
    return _EmbedConvIn.apply(x_tab, w, b, w_in, b_in, st, norm, out_dtype)


class _EmbedConvInLinear(torch.autograd.Function):
    """linear(ELU(conv_in(embedding(x))), w_out, bias_out) for the scalar-feature model — the first layer by linearity
    (_EmbedConvIn) FUSED into the dense layer that consumes it (conv_out's GCNConv.lin, gnn.py:164-166; linear_out,
    gnn.py:147): the [N, H] rows are generated inside the kernels (pangnn_embed_linear_fwd / _bwd), never written or read.
    Backward: one pass over g = dL/dy gives dL/dW_out (+ dL/dbias_out), a second one the three weighted column sums of
    dL/dh = (g W_out) * ELU'(h) — all the first layer's parameters need — without writing dL/dh.
    Forward equals linear(embed_conv_in(...), in_act=1) bit for bit (same generated values, same product); the parameter
    gradients differ from that form by fp32 re-association."""

    @staticmethod
    def forward(ctx, x_tab, w, b, w_in, b_in, w_out, bias_out, st, norm):
        lib = _lib.load()
        _lib.require_device(x_tab, w, b, w_in, b_in, w_out, bias_out)
        r, s = _node_actions(x_tab, st, norm)
        wv, bv, win = _f32c(w.detach().reshape(-1)), _f32c(b.detach().reshape(-1)), _f32c(w_in.detach())
        bin_ = None if b_in is None else _f32c(b_in.detach())
        wout = _f32c(w_out.detach())
        bout = None if bias_out is None else _f32c(bias_out.detach())
        n, h, d, m = st.num_nodes, win.shape[0], win.shape[1], wout.shape[0]
        if wout.shape[1] != h or not lib.pangnn_embed_linear_supported(h, m):
            raise ValueError(f"embed_conv_in_linear: unsupported widths H={h}, M={m}")
        y = torch.empty(n, m, dtype=torch.float32, device=r.device)
        with _lib.device_guard(r.device):
            _lib.check(lib.pangnn_embed_linear_fwd(r.data_ptr(), s.data_ptr(), n, wv.data_ptr(), bv.data_ptr(), win.data_ptr(),
                                                   _lib.ptr(bin_), d, h, wout.data_ptr(), _lib.ptr(bout), m, y.data_ptr(),
                                                   y.stride(0), _lib.stream_ptr()), "pangnn_embed_linear_fwd")
        ctx.save_for_backward(r, s, wv, bv, win, wout, bin_ if bin_ is not None else wv.new_empty(0))
        ctx.has_bin, ctx.has_bout = b_in is not None, bias_out is not None
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        r, s, wv, bv, win, wout, bin_ = ctx.saved_tensors
        bin_ = bin_ if ctx.has_bin else None
        g = _rows_f32(g)
        n, m = g.shape
        h, d = win.shape
        dev = g.device
        out = torch.empty(m * h + m + 3 * h + d + d + h * d + h, dtype=torch.float32, device=dev)     # one allocation
        o = 0
        g_wout, o = out[o:o + m * h].view(m, h), o + m * h
        g_bout, o = out[o:o + m], o + m
        sums, o = out[o:o + 3 * h], o + 3 * h
        g_w, o = out[o:o + d].view(d, 1), o + d
        g_b, o = out[o:o + d], o + d
        g_win, o = out[o:o + h * d].view(h, d), o + h * d
        g_bin = out[o:o + h]
        with _lib.device_guard(dev):
            ws_bytes = lib.pangnn_embed_linear_bwd_workspace_bytes(h, m)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.pangnn_embed_linear_bwd(g.data_ptr(), g.stride(0), r.data_ptr(), s.data_ptr(), n, wv.data_ptr(),
                                                   bv.data_ptr(), win.data_ptr(), _lib.ptr(bin_), d, h, wout.data_ptr(), m,
                                                   g_wout.data_ptr(), g_bout.data_ptr() if ctx.has_bout else None,
                                                   sums.data_ptr(), ws.data_ptr(), ws_bytes, _lib.stream_ptr()),
                       "pangnn_embed_linear_bwd")
            _lib.check(lib.pangnn_embed_conv_in_grads_from_sums(sums.data_ptr(), wv.data_ptr(), bv.data_ptr(), win.data_ptr(),
                                                                d, h, g_w.data_ptr(), g_b.data_ptr(), g_win.data_ptr(),
                                                                g_bin.data_ptr() if ctx.has_bin else None,
                                                                _lib.stream_ptr()), "pangnn_embed_conv_in_grads_from_sums")
        return (None, g_w, g_b, g_win, g_bin if ctx.has_bin else None, g_wout, g_bout if ctx.has_bout else None, None, None)


@torch.compiler.assume_constant_result
def embed_linear_supported(h: int, m: int) -> bool:
    return bool(_lib.load().pangnn_embed_linear_supported(int(h), int(m)))     # (64, 64), (64, 128), (128, 64)


def embed_conv_in_linear(x_tab, w, b, w_in, b_in, w_out, bias_out, st, norm):
    _lib.require_device(x_tab, w, b, w_in, b_in, w_out, bias_out)
    if _via_ops(st) and getattr(norm, "weight_ref", norm) is not norm:
        from . import torch_ops
        return torch_ops.embed_conv_in_linear(x_tab, w, b, w_in, b_in, w_out, bias_out, st, norm)
    return _EmbedConvInLinear.apply(x_tab, w, b, w_in, b_in, w_out, bias_out, st, norm)


class _PQOperands(torch.autograd.Function):
    """mlp[0] = Linear(2D (+1), D) of the decoder (gnn.py:110,173-175) re-arranged for the node-level P | Q product:
    w_pq = [W[:, :D] ; W[:, D:2D]]  ([2D, D]),  b_pq = [0 ; b]  ([2D]),  cvec = W[:, 2D]  (skip connections only).
    Forward is two or three small copies; backward assembles dL/dW with ONE concatenation (autograd's own rules for the three
    slices cost two zero-fills, two strided copies and two additions per step: a fifth of a replayed mini-batch step's
    launches).  Plain data movement: the values are exactly those of the sliced form."""

    @staticmethod
    def forward(ctx, w, b, d, skip):
        ctx.d, ctx.skip = int(d), bool(skip)
        d = int(d)
        if w.is_cuda and not observed() and w.dtype == torch.float32 and b.dtype == torch.float32 and w.dim() == 2 \
                and w.stride(1) == 1 and b.is_contiguous() and w.shape[1] >= 2 * d + int(bool(skip)):
            # one launch (a fresh or replayed mini-batch step is made of launches: cat + pad + copy were three)
            lib = _lib.load()
            out = torch.empty(2 * d * d + 2 * d + (d if skip else 0), dtype=torch.float32, device=w.device)
            w_pq, b_pq = out[:2 * d * d].view(2 * d, d), out[2 * d * d:2 * d * d + 2 * d]
            cvec = out[2 * d * d + 2 * d:] if skip else None
            with _lib.device_guard(w.device):
                _lib.check(lib.pangnn_pq_operands_f32(w.data_ptr(), w.stride(0), b.data_ptr(), d, int(bool(skip)),
                                                      w_pq.data_ptr(), b_pq.data_ptr(), _lib.ptr(cvec), _lib.stream_ptr()),
                           "pangnn_pq_operands_f32")
            return w_pq, b_pq, cvec
        w_pq = torch.cat([w[:, :d], w[:, d:2 * d]], dim=0)
        b_pq = torch.nn.functional.pad(b, (d, 0))
        cvec = w[:, 2 * d].contiguous() if skip else None
        return w_pq, b_pq, cvec

    @staticmethod
    def backward(ctx, g_wpq, g_bpq, g_cvec):
        d = ctx.d
        g_w = g_b = None
        if g_wpq is not None or g_cvec is not None:
            cols = [g_wpq[:d] if g_wpq is not None else g_cvec.new_zeros(d, d),
                    g_wpq[d:] if g_wpq is not None else g_cvec.new_zeros(d, d)]
            if ctx.skip:
                cols.append((g_cvec if g_cvec is not None else g_wpq.new_zeros(d)).reshape(d, 1))
            g_w = torch.cat(cols, dim=1)
        if g_bpq is not None:
            g_b = g_bpq[d:]
        return g_w, g_b, None, None


def pq_operands(w, b, d: int, skip: bool):
    """(w_pq, b_pq, cvec) of the decoder's first layer (see _PQOperands); sliced with plain autograd while a compiler traces"""
    if torch.compiler.is_compiling():
        return (torch.cat([w[:, :d], w[:, d:2 * d]], dim=0), torch.cat([torch.zeros_like(b), b]),
                w[:, 2 * d].contiguous() if skip else None)
    return _PQOperands.apply(w, b, d, skip)

