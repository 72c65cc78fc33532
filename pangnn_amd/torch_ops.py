"""torch dispatcher view of the hot-path operators: `torch.ops.pangnn.*`.

The schemas and their HIP implementations are registered in C++ (csrc/torch_ops.cpp: TORCH_LIBRARY(pangnn, ...) +
TORCH_LIBRARY_IMPL(pangnn, CUDA, ...) over the C ABI of libpangnn_hip.so); this module loads that library and
registers, on the same ops,
  * fake (meta) kernels — shapes / dtypes only, so FakeTensor tracing and torch.compile work without a GPU launch,
  * autograd formulas  — propagate: the transposed propagate over the by-source CSR (edge weights are not
    differentiated); edge_gather_concat: two segment sums; segment_max_rows: scatter to the arg-max,
  * the autocast policy — what PyG does under `accelerate`'s mixed precision: the propagate gathers bfloat16 rows as
    stored (half the bytes) with fp32 weights / accumulation / result; float16 rows are promoted to fp32 (the kernels
    have no fp16 row format); everything else runs in fp32.
`functional.propagate` goes through `torch.ops.pangnn.propagate` when `functional.USE_DISPATCHER_OPS` is set
(PANGNN_DISPATCHER_OPS=1), otherwise through the ctypes autograd.Function of round 1 (same kernels, same results).
There is no CPU implementation: the ops raise on CPU tensors.
"""
from __future__ import annotations

import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
TLIB_PATH = os.path.join(_HERE, "libpangnn_torch.so")

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
def _propagate_setup(ctx, inputs, output):
    rowptr, other, val, rowptr_t, other_t, val_t, x, bias = inputs
    ctx.save_for_backward(rowptr_t, other_t, val_t)
    ctx.n_src, ctx.x_dtype, ctx.has_bias = x.shape[0], x.dtype, bias is not None


def _propagate_backward(ctx, g):
    rowptr_t, other_t, val_t = ctx.saved_tensors
    gx = ops.spmm(rowptr_t, other_t, val_t, g, None, ctx.n_src) if ctx.needs_input_grad[6] else None
    if gx is not None and gx.dtype != ctx.x_dtype:
        gx = gx.to(ctx.x_dtype)
    gb = g.sum(dim=0) if (ctx.has_bias and ctx.needs_input_grad[7]) else None
    return None, None, None, None, None, None, gx, gb


torch.library.register_autograd("pangnn::propagate", _propagate_backward, setup_context=_propagate_setup)


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


def _segmax_setup(ctx, inputs, output):
    rowptr, perm, m, n_rows = inputs
    ctx.save_for_backward(output[1], rowptr)
    ctx.e = m.shape[0]
    ctx.mark_non_differentiable(output[1])


def _segmax_backward(ctx, g, _g_arg):
    arg, rowptr = ctx.saved_tensors
    return None, None, ops.segment_max_bwd(g, arg, rowptr, ctx.e), None


torch.library.register_autograd("pangnn::segment_max_rows", _segmax_backward, setup_context=_segmax_setup)


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
