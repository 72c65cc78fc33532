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
The per-step operators of the train step that read a graph (gcn_propagate, embed_conv_in[_linear], embed_propagate,
decoder_loss, decoder_mlp and their backward ops) are C++ too since round 5 (csrc/graph_ops.cpp: schema, HIP implementation,
autograd formula, and the structure registry they look a graph up in by the identity of its `edge_index` tensor); the second
half of this module holds their fake kernels, the registry's build-on-miss hook and the wrappers `functional` calls with this
package's structure objects (`functional.USE_DISPATCHER_OPS`; PANGNN_DISPATCHER_OPS=auto: only when a tracer / dispatch mode
observes the calls, =0: never — the ctypes autograd.Functions: same kernels, same results).
There is no CPU implementation: the ops raise on CPU tensors.
"""
from __future__ import annotations

import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROW_DTYPES = (torch.float32, torch.bfloat16, torch.float16)       # indexed by PANGNN_DTYPE_*


def _code(dtype) -> int:
    """PANGNN_DTYPE_* of a torch dtype (None: f32)"""
    return 1 if dtype == torch.bfloat16 else 2 if dtype == torch.float16 else 0

TLIB_PATH = os.environ.get("PANGNN_TORCH_LIB") or os.path.join(_HERE, "libpangnn_torch.so")   # override: the host-sanitizer build

if not os.path.exists(TLIB_PATH):
    raise ImportError(f"pangnn_amd: {TLIB_PATH} is missing — build it with `make -C pangnn_amd/csrc` "
                      f"(or `python -c 'import __graft_entry__ as g; g.build()'`)")
from . import _lib                                     # noqa: E402
_lib.load()     # first: libpangnn_torch.so needs "libpangnn_hip.so" by SONAME, which the library already in the process (the in-tree
#                 one, or a diagnostic build chosen through PANGNN_HIP_LIB and linked with the same -soname) then satisfies
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
    bfloat16 / float16: rows are gathered as stored (pangnn_spmm_csr_bf16 / _f16)."""
    dt = torch.get_autocast_dtype("cuda")
    if x.is_floating_point():
        x = x.to(dt) if dt in (torch.bfloat16, torch.float16) else x.float()
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
# The per-step operators of the train step that READ A GRAPH: gcn_propagate, embed_conv_in, embed_conv_in_linear,
# embed_propagate, decoder_loss, decoder_mlp and their backward ops.  Round 5: schema, HIP implementation and autograd
# formula are C++ (csrc/graph_ops.cpp), like pangnn::linear — rounds 3-4 registered them from here over ctypes.  Schemas take
# TENSORS only: a graph enters as its `edge_index` (and `edge_weight`) tensor and the implementation finds what was built for
# it in the native structure registry by identity.  What stays here: the fake (meta) kernels — FakeTensor tracing and
# torch.compile see ordinary ops —, the ONE hook the registry calls on a miss (pangnn::_prepare_structure: graph.py builds
# and pushes what is missing), and thin wrappers that take this package's structure objects and make sure their tables are
# pushed before the op is called (a few integer tests per call in the steady state).
# There is no CPU implementation: the ops raise on CPU tensors.
# ==============================================================================================================
from . import functional as _PF          # noqa: E402
from . import graph as _G                # noqa: E402

_hooks = torch.library.Library("pangnn", "IMPL")


def _prepare_structure(edge_index, n_dst, edge_weight, x, need):
    """registry miss: build (or find) the structure of `edge_index` and push the components `need` names"""
    _PF._lib.require_device(edge_index)
    st = _G.structure_of(edge_index, int(n_dst))
    _G.register(st)
    st.push_native(int(need) | _G.NEED_ENTRY, edge_weight, x, force=True)


_hooks.impl("_prepare_structure", _prepare_structure, "CompositeExplicitAutograd")


# ---------------------------------------------------------------------------------------------- fake kernels
@torch.library.register_fake("pangnn::linear")
def _(x, w, bias, in_act, out_dtype):
    return x.new_empty(x.shape[0], w.shape[0], dtype=_ROW_DTYPES[out_dtype])


@torch.library.register_fake("pangnn::linear_backward")
def _(g, x, w, in_act, has_bias, need_dx):
    return (x.new_empty(x.shape if need_dx else (0,)), w.new_empty(w.shape, dtype=torch.float32),
            w.new_empty(w.shape[0] if has_bias else 0, dtype=torch.float32))


@torch.library.register_fake("pangnn::bce_with_logits")
def _(logits, y, pos_weight, denom):
    return logits.new_empty((), dtype=torch.float32), logits.new_empty(logits.shape, dtype=torch.float32)


@torch.library.register_fake("pangnn::gcn_propagate")
def _(x, bias, edge_index, edge_weight, allow_band, out_dtype):
    return x.new_empty(x.shape, dtype=_ROW_DTYPES[out_dtype])


@torch.library.register_fake("pangnn::gcn_propagate_backward")
def _(g, edge_index, edge_weight, allow_band, has_bias, x_dtype):
    return (g.new_empty(g.shape, dtype=_ROW_DTYPES[x_dtype]),
            g.new_empty(g.shape[1] if has_bias else 0, dtype=torch.float32))


@torch.library.register_fake("pangnn::embed_conv_in")
def _(x, w, b, w_in, b_in, edge_index, edge_weight, out_dtype):
    return w_in.new_empty(x.shape[0], w_in.shape[0], dtype=_ROW_DTYPES[out_dtype])


@torch.library.register_fake("pangnn::embed_conv_in_backward")
def _(g, x, w, b, w_in, edge_index, edge_weight, has_bias):
    f = lambda *s: w_in.new_empty(s, dtype=torch.float32)        # noqa: E731
    return f(w_in.shape[1], 1), f(w_in.shape[1]), f(*w_in.shape), f(w_in.shape[0] if has_bias else 0)


@torch.library.register_fake("pangnn::embed_conv_in_linear")
def _(x, w, b, w_in, b_in, w_out, bias_out, edge_index, edge_weight):
    return w_out.new_empty(x.shape[0], w_out.shape[0], dtype=torch.float32)


@torch.library.register_fake("pangnn::embed_conv_in_linear_backward")
def _(g, x, w, b, w_in, b_in, w_out, edge_index, edge_weight, has_bias_out):
    f = lambda *s: w_in.new_empty(s, dtype=torch.float32)        # noqa: E731
    h, d = w_in.shape
    return (f(d, 1), f(d), f(h, d), f(h if b_in is not None else 0), f(*w_out.shape), f(w_out.shape[0] if has_bias_out else 0))


@torch.library.register_fake("pangnn::embed_propagate")
def _(x, w, b, edge_index, edge_weight):
    return w.new_empty(x.shape[0], w.shape[0], dtype=torch.float32)


@torch.library.register_fake("pangnn::embed_propagate_backward")
def _(g, x, edge_index, edge_weight):
    return g.new_empty(g.shape[1], 1, dtype=torch.float32), g.new_empty(g.shape[1], dtype=torch.float32)


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


# ---------------------------------------------------------------------------------------------- wrappers taking structures
_N = _G


def _ready(st, need, norm=None, x=None):
    """make `st` findable by identity (Python cache and native registry) with the tables `need` names; the weight tensor the
    op is keyed on.  A tensor-only stand-in (graph.TracedStructure, while torch.compile traces) passes through."""
    w = getattr(norm, "weight_ref", None)
    if not getattr(st, "traced", False):
        _G.register(st)
        if not st.native_has(need, norm, x):
            st.push_native(need, w, x)
    return w


def gcn_propagate(x, bias, st, norm, allow_band=False, out_dtype=None):
    band = bool(allow_band) and getattr(norm, "weight_ref", None) is None and x.dim() == 2 and x.shape[1] in (64, 128)
    w = _ready(st, _N.NEED_BY_DST | _N.NEED_NORM | (_N.NEED_BAND if band else 0), norm)
    return ops.gcn_propagate(x, bias, st._key_tensor, w, bool(allow_band), _code(out_dtype))


def embed_conv_in(x_tab, w, b, w_in, b_in, st, norm, out_dtype=None):
    wt = _ready(st, _N.NEED_BY_DST | _N.NEED_NORM | _N.NEED_ACTIONS, norm, x_tab)
    return ops.embed_conv_in(x_tab, w, b, w_in, b_in, st._key_tensor, wt, _code(out_dtype))


def embed_conv_in_linear(x_tab, w, b, w_in, b_in, w_out, bias_out, st, norm):
    wt = _ready(st, _N.NEED_BY_DST | _N.NEED_NORM | _N.NEED_ACTIONS, norm, x_tab)
    return ops.embed_conv_in_linear(x_tab, w, b, w_in, b_in, w_out, bias_out, st._key_tensor, wt)


def embed_propagate(x_tab, w, b, st, norm):
    wt = _ready(st, _N.NEED_BY_DST | _N.NEED_NORM | _N.NEED_ACTIONS, norm, x_tab)
    return ops.embed_propagate(x_tab, w, b, st._key_tensor, wt)


def decoder_loss_pq(pq, st, extra, cvec, w2, b2, w3, b3, y, pos_weight, denom, live=None):
    _ready(st, _N.NEED_BY_DST | _N.NEED_RUNSUM | _N.NEED_PLAN_DST)
    out = ops.decoder_loss(pq, st._key_tensor, extra, cvec, w2, b2, w3, b3, y, pos_weight, int(denom), live)
    return out[0], out[1]


def decoder_mlp_pq(pq, st, extra, cvec, w2, b2, w3, b3):
    _ready(st, _N.NEED_ENTRY)
    return ops.decoder_mlp(pq, st._key_tensor, extra, cvec, w2, b2, w3, b3)
