"""torch dispatcher view of the hot-path operators: `torch.ops.pangnn.*`.

The schemas and their HIP implementations are registered in C++ (csrc/torch_ops.cpp: TORCH_LIBRARY(pangnn, ...) +
TORCH_LIBRARY_IMPL(pangnn, CUDA, ...) over the C ABI of libpangnn_hip.so); this module loads that library and
registers, on the same ops,
  * fake (meta) kernels — shapes / dtypes only, so FakeTensor tracing and torch.compile work without a GPU launch,
  * autograd formulas  — edge_gather_concat: two segment sums over the cached CSR orders (propagate — the transposed
    propagate over the by-source CSR, edge weights not differentiated, SURVEY.md §8 a6 — and segment_max_rows — scatter to
    the arg-max — have theirs in C++, csrc/torch_ops.cpp),
  * the autocast policy — what PyG does under `accelerate`'s mixed precision: the propagate gathers bfloat16 rows as
    stored (half the bytes) with fp32 weights / accumulation / result; float16 rows are promoted to fp32 (the kernels
    have no fp16 row format); everything else runs in fp32.
The second half of this module registers the per-step operators of the train step (linear, gcn_propagate, embed_conv_in,
decoder_loss, decoder_mlp and their backward ops) from Python on the same library; `functional`'s public functions go
through them by default — eager, traced or captured alike (`functional.USE_DISPATCHER_OPS`; PANGNN_DISPATCHER_OPS=auto: only
when a tracer / dispatch mode observes the calls, =0: never — the ctypes autograd.Functions: same kernels, same results).
There is no CPU implementation: the ops raise on CPU tensors.
"""
from __future__ import annotations

import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
TLIB_PATH = os.environ.get("PANGNN_TORCH_LIB") or os.path.join(_HERE, "libpangnn_torch.so")   # override: the host-sanitizer build

if not os.path.exists(TLIB_PATH):
    raise ImportError(f"pangnn_amd: {TLIB_PATH} is missing — build it with `make -C pangnn_amd/csrc` "
                      f"(or `python -c 'import __graft_entry__ as g; g.build()'`)")
torch.ops.load_library(TLIB_PATH)
ops = torch.ops.pangnn


# ---------------------------------------------------------------------------------------------- fake kernels
@torch.library.register_fake("pangnn::csr_from_coo")
def _(edge_index, num_nodes, group_by):
    e = edge_index.shape[1]
    return (edge_index.new_empty(num_nodes + 1), edge_index.new_empty(e, dtype=torch.int32),
            edge_index.new_empty(e, dtype=torch.int32))


@torch.library.register_fake("pangnn::gcn_norm")
def _(rowptr, other, perm, edge_weight):
    n, e = rowptr.shape[0] - 1, other.shape[0]
    f = lambda *s: rowptr.new_empty(*s, dtype=torch.float32)           # noqa: E731
    return f(n), f(e), f(e)


@torch.library.register_fake("pangnn::spmm")
def _(rowptr, other, val, x, bias, n_rows):
    return x.new_empty(n_rows, x.shape[1], dtype=torch.float32)


@torch.library.register_fake("pangnn::propagate")
def _(rowptr, other, val, rowptr_t, other_t, val_t, x, bias):
    return x.new_empty(rowptr.shape[0] - 1, x.shape[1], dtype=torch.float32)


@torch.library.register_fake("pangnn::edge_gather_concat")
def _(z, edge_index, extra):
    return z.new_empty(edge_index.shape[1], 2 * z.shape[1] + (0 if extra is None else 1), dtype=torch.float32)


@torch.library.register_fake("pangnn::segment_sum_rows")
def _(rowptr, perm, m, col_off, f, n_rows):
    return m.new_empty(n_rows, f, dtype=torch.float32)


@torch.library.register_fake("pangnn::segment_max_rows")
def _(rowptr, perm, m, n_rows):
    return m.new_empty(n_rows, m.shape[1], dtype=torch.float32), m.new_empty(n_rows, m.shape[1], dtype=torch.int32)


@torch.library.register_fake("pangnn::segment_max_bwd")
def _(g, arg, rowptr, num_edges):
    return g.new_empty(num_edges, g.shape[1], dtype=torch.float32)


# ---------------------------------------------------------------------------------------------- autograd
# pangnn::propagate and pangnn::segment_max_rows: autograd formulas in C++ (csrc/torch_ops.cpp, Autograd key)
def _gather_setup(ctx, inputs, output):
    z, edge_index, extra = inputs
    ctx.edge_index, ctx.n, ctx.d = edge_index, z.shape[0], z.shape[1]


def _gather_backward(ctx, g):
    from .graph import structure_of
    st = structure_of(ctx.edge_index, ctx.n)
    gz = ops.segment_sum_rows(st.by_src.rowptr, st.by_src.perm, g, 0, ctx.d, ctx.n) + \
        ops.segment_sum_rows(st.by_dst.rowptr, st.by_dst.perm, g, ctx.d, ctx.d, ctx.n)
    return gz, None, None


torch.library.register_autograd("pangnn::edge_gather_concat", _gather_backward, setup_context=_gather_setup)


# ---------------------------------------------------------------------------------------------- autocast policy
_impl = torch.library.Library("pangnn", "IMPL")
_NO_AUTOCAST = torch._C.DispatchKeySet(torch._C.DispatchKey.AutocastCUDA)


def _autocast_propagate(rowptr, other, val, rowptr_t, other_t, val_t, x, bias):
    """PyG under autocast: the message gather runs on rows of the autocast dtype, weights and sums stay fp32.
    bfloat16: rows are gathered as stored (pangnn_spmm_csr_bf16); float16 has no row format here -> fp32."""
    dt = torch.get_autocast_dtype("cuda")
    if x.is_floating_point():
        x = x.to(torch.bfloat16) if dt == torch.bfloat16 else x.float()
    with torch._C._ExcludeDispatchKeyGuard(_NO_AUTOCAST):
        return ops.propagate(rowptr, other, val, rowptr_t, other_t, val_t, x, bias)


_impl.impl("propagate", _autocast_propagate, "AutocastCUDA")
for _name in ("edge_gather_concat", "segment_sum_rows", "segment_max_rows"):
    torch.library.register_autocast(f"pangnn::{_name}", "cuda", torch.float32)


def propagate(x, bias, st, norm):
    """`A_hat x + bias` through the dispatcher op (st: graph.EdgeStructure, norm: graph.GcnNorm)"""
    return ops.propagate(st.by_dst.rowptr, st.by_dst.other, norm.by_dst, st.by_src.rowptr, st.by_src.other, norm.by_src,
                         x, bias)


# ==============================================================================================================
# The per-step operators of the train step as dispatcher ops (round 3): dense layer, GCN propagate, first layer by
# linearity, training / inference decoder.  Schemas take TENSORS only — a graph enters as its `edge_index` (and
# `edge_weight`) tensor and the implementation looks the cached EdgeStructure / GcnNorm up by identity
# (graph.structure_of; the wrappers below register the caller's structure first) — so FakeTensor tracing and
# torch.compile see ordinary ops with fake kernels and autograd formulas, and the formulas themselves only call
# registered ops (the backward graph is traceable too).  The implementations are the SAME code as the ctypes
# autograd.Functions of functional.py, called with a plain context object: identical kernels, identical results.
# No CPU implementation: the ops raise on CPU tensors.
# ==============================================================================================================
from . import functional as _PF          # noqa: E402
from . import graph as _G                # noqa: E402

_lib2 = torch.library.Library("pangnn", "FRAGMENT")
# (pangnn::linear / pangnn::linear_backward: schema, HIP implementation and autograd formula live in csrc/torch_ops.cpp)
_lib2.define("gcn_propagate(Tensor x, Tensor? bias, Tensor edge_index, Tensor? edge_weight, bool allow_band, bool out_bf16) -> Tensor")
_lib2.define("gcn_propagate_backward(Tensor g, Tensor edge_index, Tensor? edge_weight, bool allow_band, bool has_bias, bool x_bf16) -> (Tensor, Tensor)")
_lib2.define("embed_conv_in(Tensor x, Tensor w, Tensor b, Tensor w_in, Tensor? b_in, Tensor edge_index, Tensor? edge_weight, "
             "bool out_bf16) -> Tensor")
_lib2.define("embed_conv_in_backward(Tensor g, Tensor x, Tensor w, Tensor b, Tensor w_in, Tensor edge_index, "
             "Tensor? edge_weight, bool has_bias) -> (Tensor, Tensor, Tensor, Tensor)")
_lib2.define("decoder_loss(Tensor pq, Tensor edge_index, Tensor? extra, Tensor? cvec, Tensor w2, Tensor b2, Tensor w3, "
             "Tensor b3, Tensor y, Tensor? pos_weight, int denom, Tensor? live) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)")
_lib2.define("decoder_mlp(Tensor pq, Tensor edge_index, Tensor? extra, Tensor? cvec, Tensor w2, Tensor b2, Tensor w3, "
             "Tensor b3) -> Tensor")
_lib2.define("decoder_mlp_backward(Tensor g, Tensor pq, Tensor edge_index, Tensor? extra, Tensor? cvec, Tensor w2, Tensor b2, "
             "Tensor w3, Tensor b3) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)")


_lib2.define("embed_propagate(Tensor x, Tensor w, Tensor b, Tensor edge_index, Tensor? edge_weight) -> Tensor")
_lib2.define("embed_propagate_backward(Tensor g, Tensor x, Tensor edge_index, Tensor? edge_weight) -> (Tensor, Tensor)")


_lib2.define("embed_conv_in_linear(Tensor x, Tensor w, Tensor b, Tensor w_in, Tensor? b_in, Tensor w_out, Tensor? bias_out, "
             "Tensor edge_index, Tensor? edge_weight) -> Tensor")
_lib2.define("embed_conv_in_linear_backward(Tensor g, Tensor x, Tensor w, Tensor b, Tensor w_in, Tensor? b_in, Tensor w_out, "
             "Tensor edge_index, Tensor? edge_weight, bool has_bias_out) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)")


class _Ctx:
    """what an autograd.Function's static forward / backward needs from its context, outside autograd"""

    def __init__(self, needs=()):
        self.needs_input_grad = tuple(needs)
        self.saved_tensors = ()

    def save_for_backward(self, *t):
        self.saved_tensors = t

    def mark_non_differentiable(self, *a):
        pass


def _struct(edge_index, n):
    _PF._lib.require_device(edge_index)
    return _G.structure_of(edge_index, int(n))


def _e(ref, *shape):
    return ref.new_empty(shape, dtype=torch.float32)


# ---------------------------------------------------------------------------------------------- dense layer
# implementation (TORCH_LIBRARY_IMPL(pangnn, CUDA)) and autograd formula (TORCH_LIBRARY_IMPL(pangnn, Autograd): a
# torch::autograd::Function whose backward is ONE pangnn::linear_backward call) are C++ — csrc/torch_ops.cpp; only the
# fake kernels are registered here
@torch.library.register_fake("pangnn::linear")
def _(x, w, bias, in_act, out_bf16):
    return x.new_empty(x.shape[0], w.shape[0], dtype=torch.bfloat16 if out_bf16 else torch.float32)


@torch.library.register_fake("pangnn::linear_backward")
def _(g, x, w, in_act, has_bias, need_dx):
    return (x.new_empty(x.shape if need_dx else (0,)), w.new_empty(w.shape, dtype=torch.float32),
            w.new_empty(w.shape[0] if has_bias else 0, dtype=torch.float32))


# ---------------------------------------------------------------------------------------------- GCN propagate
def _gcn_propagate_impl(x, bias, edge_index, edge_weight, allow_band, out_bf16):
    st = _struct(edge_index, x.shape[0])
    norm = st.gcn_norm(edge_weight)
    if allow_band and _PF._band_ok(x, st, edge_weight is None):
        y = _PF._BandPropagate.forward(_Ctx(), x, bias, norm.deg_inv_sqrt, st.band_width())
        return y.to(torch.bfloat16) if out_bf16 else y
    return _PF._Propagate.forward(_Ctx(), x, bias, st, norm, None, bool(out_bf16))


def _gcn_propagate_backward_impl(g, edge_index, edge_weight, allow_band, has_bias, x_bf16):
    st = _struct(edge_index, g.shape[0])
    norm = st.gcn_norm(edge_weight)
    ctx = _Ctx((True, has_bias, False, False, False, False))
    ctx.has_bias, ctx.x_dtype = bool(has_bias), torch.bfloat16 if x_bf16 else torch.float32
    if allow_band and _PF._band_ok(g, st, edge_weight is None):
        ctx.k = st.band_width()
        ctx.save_for_backward(norm.deg_inv_sqrt)
        gx, gb = _PF._BandPropagate.backward(ctx, g)[:2]
    else:
        ctx.st, ctx.norm, ctx.tag = st, norm, None
        gx, gb = _PF._Propagate.backward(ctx, g)[:2]
    return gx, (gb if gb is not None else _e(g, 0))


_lib2.impl("gcn_propagate", _gcn_propagate_impl, "CUDA")
_lib2.impl("gcn_propagate_backward", _gcn_propagate_backward_impl, "CUDA")


@torch.library.register_fake("pangnn::gcn_propagate")
def _(x, bias, edge_index, edge_weight, allow_band, out_bf16):
    return x.new_empty(x.shape, dtype=torch.bfloat16 if out_bf16 else torch.float32)


@torch.library.register_fake("pangnn::gcn_propagate_backward")
def _(g, edge_index, edge_weight, allow_band, has_bias, x_bf16):
    return (g.new_empty(g.shape, dtype=torch.bfloat16 if x_bf16 else torch.float32),
            g.new_empty(g.shape[1] if has_bias else 0, dtype=torch.float32))


def _gcn_setup(ctx, inputs, output):
    x, bias, edge_index, edge_weight, allow_band, _out_bf16 = inputs
    ctx.save_for_backward(edge_index, edge_weight)
    ctx.allow_band = allow_band
    ctx.has_bias, ctx.x_bf16 = bias is not None, x.dtype == torch.bfloat16


def _gcn_bwd(ctx, g):
    edge_index, edge_weight = ctx.saved_tensors
    gx, gb = ops.gcn_propagate_backward(g, edge_index, edge_weight, ctx.allow_band, ctx.has_bias, ctx.x_bf16)
    return (gx if ctx.needs_input_grad[0] else None, gb if ctx.has_bias else None, None, None, None, None)


torch.library.register_autograd("pangnn::gcn_propagate", _gcn_bwd, setup_context=_gcn_setup)


# ---------------------------------------------------------------------------------------------- first layer by linearity
def _embed_conv_in_impl(x, w, b, w_in, b_in, edge_index, edge_weight, out_bf16):
    st = _struct(edge_index, x.shape[0])
    return _PF._EmbedConvIn.forward(_Ctx(), x, w, b, w_in, b_in, st, st.gcn_norm(edge_weight),
                                    torch.bfloat16 if out_bf16 else None)


def _embed_conv_in_backward_impl(g, x, w, b, w_in, edge_index, edge_weight, has_bias):
    st = _struct(edge_index, x.shape[0])
    r, s = _PF._node_actions(x, st, st.gcn_norm(edge_weight))
    ctx = _Ctx()
    ctx.save_for_backward(r, s, _PF._f32c(w.reshape(-1)), _PF._f32c(b.reshape(-1)), _PF._f32c(w_in))
    ctx.has_bias = bool(has_bias)
    out = _PF._EmbedConvIn.backward(ctx, g)
    return out[1], out[2], out[3], (out[4] if out[4] is not None else _e(g, 0))


_lib2.impl("embed_conv_in", _embed_conv_in_impl, "CUDA")
_lib2.impl("embed_conv_in_backward", _embed_conv_in_backward_impl, "CUDA")


@torch.library.register_fake("pangnn::embed_conv_in")
def _(x, w, b, w_in, b_in, edge_index, edge_weight, out_bf16):
    return w_in.new_empty(x.shape[0], w_in.shape[0], dtype=torch.bfloat16 if out_bf16 else torch.float32)


@torch.library.register_fake("pangnn::embed_conv_in_backward")
def _(g, x, w, b, w_in, edge_index, edge_weight, has_bias):
    f = lambda *s: w_in.new_empty(s, dtype=torch.float32)        # noqa: E731
    return f(w_in.shape[1], 1), f(w_in.shape[1]), f(*w_in.shape), f(w_in.shape[0] if has_bias else 0)


def _eci_setup(ctx, inputs, output):
    x, w, b, w_in, b_in, edge_index, edge_weight, _ = inputs
    ctx.save_for_backward(x, w, b, w_in, edge_index, edge_weight)
    ctx.has_bias = b_in is not None


def _eci_bwd(ctx, g):
    x, w, b, w_in, edge_index, edge_weight = ctx.saved_tensors
    g_w, g_b, g_win, g_bin = ops.embed_conv_in_backward(g, x, w, b, w_in, edge_index, edge_weight, ctx.has_bias)
    return None, g_w.reshape(w.shape), g_b, g_win, (g_bin if ctx.has_bias else None), None, None, None


torch.library.register_autograd("pangnn::embed_conv_in", _eci_bwd, setup_context=_eci_setup)


# ---------------------------------------------------------------------------------------------- first layer + next dense layer
def _ecil_impl(x, w, b, w_in, b_in, w_out, bias_out, edge_index, edge_weight):
    st = _struct(edge_index, x.shape[0])
    return _PF._EmbedConvInLinear.forward(_Ctx(), x, w, b, w_in, b_in, w_out, bias_out, st, st.gcn_norm(edge_weight))


def _ecil_backward_impl(g, x, w, b, w_in, b_in, w_out, edge_index, edge_weight, has_bias_out):
    st = _struct(edge_index, x.shape[0])
    r, s = _PF._node_actions(x, st, st.gcn_norm(edge_weight))
    f = _PF._f32c
    ctx = _Ctx()
    ctx.save_for_backward(r, s, f(w.reshape(-1)), f(b.reshape(-1)), f(w_in), f(w_out), f(b_in) if b_in is not None else r.new_empty(0))
    ctx.has_bin, ctx.has_bout = b_in is not None, bool(has_bias_out)
    out = _PF._EmbedConvInLinear.backward(ctx, g)
    return (out[1], out[2], out[3], out[4] if out[4] is not None else _e(g, 0), out[5],
            out[6] if out[6] is not None else _e(g, 0))


_lib2.impl("embed_conv_in_linear", _ecil_impl, "CUDA")
_lib2.impl("embed_conv_in_linear_backward", _ecil_backward_impl, "CUDA")


@torch.library.register_fake("pangnn::embed_conv_in_linear")
def _(x, w, b, w_in, b_in, w_out, bias_out, edge_index, edge_weight):
    return w_out.new_empty(x.shape[0], w_out.shape[0], dtype=torch.float32)


@torch.library.register_fake("pangnn::embed_conv_in_linear_backward")
def _(g, x, w, b, w_in, b_in, w_out, edge_index, edge_weight, has_bias_out):
    f = lambda *s: w_in.new_empty(s, dtype=torch.float32)        # noqa: E731
    h, d = w_in.shape
    return (f(d, 1), f(d), f(h, d), f(h if b_in is not None else 0), f(*w_out.shape), f(w_out.shape[0] if has_bias_out else 0))


def _ecil_setup(ctx, inputs, output):
    x, w, b, w_in, b_in, w_out, bias_out, edge_index, edge_weight = inputs
    ctx.save_for_backward(x, w, b, w_in, b_in, w_out, edge_index, edge_weight)
    ctx.has_bout = bias_out is not None


def _ecil_bwd(ctx, g):
    x, w, b, w_in, b_in, w_out, edge_index, edge_weight = ctx.saved_tensors
    g_w, g_b, g_win, g_bin, g_wout, g_bout = ops.embed_conv_in_linear_backward(g, x, w, b, w_in, b_in, w_out, edge_index,
                                                                               edge_weight, ctx.has_bout)
    return (None, g_w.reshape(w.shape), g_b, g_win, g_bin if b_in is not None else None, g_wout,
            g_bout if ctx.has_bout else None, None, None)


torch.library.register_autograd("pangnn::embed_conv_in_linear", _ecil_bwd, setup_context=_ecil_setup)


# ---------------------------------------------------------------------------------------------- round 2's first layer
def _embed_propagate_impl(x, w, b, edge_index, edge_weight):
    st = _struct(edge_index, x.shape[0])
    return _PF._EmbedPropagate.forward(_Ctx(), x, w, b, st, st.gcn_norm(edge_weight), None)


def _embed_propagate_backward_impl(g, x, edge_index, edge_weight):
    st = _struct(edge_index, x.shape[0])
    ctx = _Ctx()
    ctx.save_for_backward(*_PF._node_actions(x, st, st.gcn_norm(edge_weight)))
    ctx.d = g.shape[1]
    out = _PF._EmbedPropagate.backward(ctx, g)
    return out[1], out[2]


_lib2.impl("embed_propagate", _embed_propagate_impl, "CUDA")
_lib2.impl("embed_propagate_backward", _embed_propagate_backward_impl, "CUDA")


@torch.library.register_fake("pangnn::embed_propagate")
def _(x, w, b, edge_index, edge_weight):
    return w.new_empty(x.shape[0], w.shape[0], dtype=torch.float32)


@torch.library.register_fake("pangnn::embed_propagate_backward")
def _(g, x, edge_index, edge_weight):
    return g.new_empty(g.shape[1], 1, dtype=torch.float32), g.new_empty(g.shape[1], dtype=torch.float32)


def _ep_setup(ctx, inputs, output):
    x, w, b, edge_index, edge_weight = inputs
    ctx.save_for_backward(x, edge_index, edge_weight)


def _ep_bwd(ctx, g):
    x, edge_index, edge_weight = ctx.saved_tensors
    g_w, g_b = ops.embed_propagate_backward(g, x, edge_index, edge_weight)
    return None, g_w, g_b, None, None


torch.library.register_autograd("pangnn::embed_propagate", _ep_bwd, setup_context=_ep_setup)


# ---------------------------------------------------------------------------------------------- criterion
# pangnn::bce_with_logits: schema, HIP implementation and autograd formula in C++ (csrc/torch_ops.cpp); fake kernel here
@torch.library.register_fake("pangnn::bce_with_logits")
def _(logits, y, pos_weight, denom):
    return logits.new_empty((), dtype=torch.float32), logits.new_empty(logits.shape, dtype=torch.float32)


# ---------------------------------------------------------------------------------------------- decoder
def _decoder_loss_impl(pq, edge_index, extra, cvec, w2, b2, w3, b3, y, pos_weight, denom, live):
    """`live` (device int64[1], optional): a padded fixed-shape batch whose first live[0] edges are real"""
    st = _struct(edge_index, pq.shape[0])
    ctx = _Ctx()
    loss, logits = _PF._DecoderLoss.forward(ctx, pq, None, st, extra, cvec, w2, b2, w3, b3, y, pos_weight, int(denom), True,
                                            live)
    g_pq, _, g_cv, g_w2, g_b2, g_w3, g_b3 = ctx.saved_tensors
    return loss, logits, g_pq, g_cv, g_w2, g_b2, g_w3, g_b3


def _decoder_mlp_impl(pq, edge_index, extra, cvec, w2, b2, w3, b3):
    st = _struct(edge_index, pq.shape[0])
    return _PF._DecoderMLP.forward(_Ctx(), pq, None, st, extra, cvec, w2, b2, w3, b3, True)


def _decoder_mlp_backward_impl(g, pq, edge_index, extra, cvec, w2, b2, w3, b3):
    st = _struct(edge_index, pq.shape[0])
    if _PF.DECODER_PRECISION != 1:
        raise RuntimeError("pangnn::decoder_mlp_backward: only the default decoder precision is registered")
    rows = _PF._rows_any(pq)
    d = rows.shape[1] // 2
    f = _PF._f32c
    ctx = _Ctx()
    ctx.st, ctx.joint = st, True
    ctx.save_for_backward(rows[:, :d], rows[:, d:], None if extra is None else f(extra), None if cvec is None else f(cvec),
                          f(w2), f(b2), f(w3), f(b3))
    out = _PF._DecoderMLP.backward(ctx, g)
    g_cv = out[4] if out[4] is not None else _e(pq, 0)
    return out[0], g_cv, out[5], out[6], out[7], out[8]


_lib2.impl("decoder_loss", _decoder_loss_impl, "CUDA")
_lib2.impl("decoder_mlp", _decoder_mlp_impl, "CUDA")
_lib2.impl("decoder_mlp_backward", _decoder_mlp_backward_impl, "CUDA")


@torch.library.register_fake("pangnn::decoder_loss")
def _(pq, edge_index, extra, cvec, w2, b2, w3, b3, y, pos_weight, denom, live):
    f = lambda *s: w2.new_empty(s, dtype=torch.float32)           # noqa: E731
    return (f(), f(edge_index.shape[1]), f(*pq.shape), f(*(cvec.shape if cvec is not None else (0,))), f(*w2.shape),
            f(*b2.shape), f(*w3.shape), f(*b3.shape))


@torch.library.register_fake("pangnn::decoder_mlp")
def _(pq, edge_index, extra, cvec, w2, b2, w3, b3):
    return w2.new_empty(edge_index.shape[1], dtype=torch.float32)


@torch.library.register_fake("pangnn::decoder_mlp_backward")
def _(g, pq, edge_index, extra, cvec, w2, b2, w3, b3):
    f = lambda *s: w2.new_empty(s, dtype=torch.float32)           # noqa: E731
    return (f(*pq.shape), f(*(cvec.shape if cvec is not None else (0,))), f(*w2.shape), f(*b2.shape), f(*w3.shape),
            f(*b3.shape))


def _dloss_setup(ctx, inputs, output):
    ctx.save_for_backward(*output[2:])
    ctx.has_cv = inputs[3] is not None
    ctx.mark_non_differentiable(*output[1:])
    ctx.set_materialize_grads(False)


def _dloss_bwd(ctx, go, *_unused):
    if go is None:
        return (None,) * 12
    g_pq, g_cv, g_w2, g_b2, g_w3, g_b3 = ctx.saved_tensors
    if not _PF.is_unit_grad(go):
        # loss.backward() / accelerate's (loss / 1).backward(): a device scalar that is 1 — one launch that finds that out
        # on the device and leaves (functional.scale_by_loss_grad_)
        g_pq, g_cv, g_w2, g_b2, g_w3, g_b3 = _PF.scale_by_loss_grad_(
            ctx, [g_pq, g_cv if ctx.has_cv else None, g_w2, g_b2, g_w3, g_b3], go)
    return (g_pq, None, None, g_cv if ctx.has_cv else None, g_w2, g_b2, g_w3, g_b3, None, None, None, None)


torch.library.register_autograd("pangnn::decoder_loss", _dloss_bwd, setup_context=_dloss_setup)


def _dmlp_setup(ctx, inputs, output):
    pq, edge_index, extra, cvec, w2, b2, w3, b3 = inputs
    ctx.save_for_backward(pq, w2, b2, w3, b3, edge_index, extra, cvec)


def _dmlp_bwd(ctx, g):
    pq, w2, b2, w3, b3, edge_index, extra, cvec = ctx.saved_tensors
    g_pq, g_cv, g_w2, g_b2, g_w3, g_b3 = ops.decoder_mlp_backward(g, pq, edge_index, extra, cvec, w2, b2, w3, b3)
    return g_pq, None, None, (g_cv if cvec is not None else None), g_w2, g_b2, g_w3, g_b3


torch.library.register_autograd("pangnn::decoder_mlp", _dmlp_bwd, setup_context=_dmlp_setup)


# ---------------------------------------------------------------------------------------------- wrappers taking structures
def gcn_propagate(x, bias, st, norm, allow_band=False, out_bf16=False):
    _G.register(st)
    return ops.gcn_propagate(x, bias, st._key_tensor, getattr(norm, "weight_ref", None), bool(allow_band), bool(out_bf16))


def embed_conv_in(x_tab, w, b, w_in, b_in, st, norm, out_dtype=None):
    _G.register(st)
    return ops.embed_conv_in(x_tab, w, b, w_in, b_in, st._key_tensor, getattr(norm, "weight_ref", None),
                             out_dtype == torch.bfloat16)


def embed_conv_in_linear(x_tab, w, b, w_in, b_in, w_out, bias_out, st, norm):
    _G.register(st)
    return ops.embed_conv_in_linear(x_tab, w, b, w_in, b_in, w_out, bias_out, st._key_tensor, getattr(norm, "weight_ref", None))


def embed_propagate(x_tab, w, b, st, norm):
    _G.register(st)
    return ops.embed_propagate(x_tab, w, b, st._key_tensor, getattr(norm, "weight_ref", None))


def decoder_loss_pq(pq, st, extra, cvec, w2, b2, w3, b3, y, pos_weight, denom, live=None):
    _G.register(st)
    out = ops.decoder_loss(pq, st._key_tensor, extra, cvec, w2, b2, w3, b3, y, pos_weight, int(denom), live)
    return out[0], out[1]


def decoder_mlp_pq(pq, st, extra, cvec, w2, b2, w3, b3):
    _G.register(st)
    return ops.decoder_mlp(pq, st._key_tensor, extra, cvec, w2, b2, w3, b3)
