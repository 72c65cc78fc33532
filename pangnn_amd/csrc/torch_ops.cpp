// torch dispatcher registration of the hot-path operators: TORCH_LIBRARY(pangnn, ...) schemas with
// HIP ("CUDA" dispatch key on ROCm builds of torch) implementations that call the C ABI of libpangnn_hip.so on
// torch's current stream.  No arithmetic lives here: each implementation checks its operands (TORCH_CHECK ->
// Python RuntimeError), allocates the outputs with at::empty on the input's device and forwards raw pointers.
// The autograd formulas of the ops that are pure functions of their tensor operands (linear, bce_with_logits, propagate,
// segment_max_rows) are torch::autograd::Functions below, registered under the Autograd key.  Fake (meta) kernels, the autocast
// policy and the formulas of the ops that look a cached graph structure up by its edge_index tensor are registered on the same
// ops from Python (pangnn_amd/torch_ops.py: torch.library.register_fake / register_autograd), so `accelerate`'s autocast and
// torch.compile see ordinary dispatcher ops.  Built by csrc/Makefile into pangnn_amd/libpangnn_torch.so (g++ against the torch
// headers; contains no device code).
#include "torch_common.h"

using namespace pangnn_torch;

namespace {

// csr_from_coo(edge_index i64[2,E], num_nodes, group_by) -> (rowptr i64[N+1], other i32[E], perm i32[E])
std::tuple<at::Tensor, at::Tensor, at::Tensor> csr_from_coo(const at::Tensor& edge_index, int64_t num_nodes,
                                                           int64_t group_by) {
  on_gpu(edge_index, "edge_index");
  TORCH_CHECK(edge_index.scalar_type() == at::kLong && edge_index.dim() == 2 && edge_index.size(0) == 2,
              "pangnn::csr_from_coo: edge_index must be int64 [2, E]");
  TORCH_CHECK(num_nodes >= 0 && (group_by == 0 || group_by == 1), "pangnn::csr_from_coo: num_nodes >= 0, group_by in {0, 1}");
  const DeviceGuard guard(edge_index.device());
  const at::Tensor ei = edge_index.contiguous();
  const int64_t e = ei.size(1);
  auto rowptr = at::empty({num_nodes + 1}, ei.options());
  auto other = at::empty({e}, ei.options().dtype(at::kInt));
  auto perm = at::empty({e}, ei.options().dtype(at::kInt));
  const size_t wsb = pangnn_csr_build_workspace_bytes(e, num_nodes);
  TORCH_CHECK(wsb > 0, "pangnn::csr_from_coo: workspace query failed: ", pangnn_last_error());
  auto ws = at::empty({(int64_t)wsb}, ei.options().dtype(at::kByte));
  check_rc(pangnn_csr_build(ei.data_ptr<int64_t>(), e, e, num_nodes, (int)group_by, rowptr.data_ptr<int64_t>(),
                            other.data_ptr<int32_t>(), perm.data_ptr<int32_t>(), ws.data_ptr(), wsb, stream_of(ei)),
           "pangnn_csr_build");
  return {rowptr, other, perm};
}

// gcn_norm(rowptr, other, perm, w?) -> (deg^-1/2 [N], norm in CSR order [E], norm in the caller's edge order [E])
std::tuple<at::Tensor, at::Tensor, at::Tensor> gcn_norm(const at::Tensor& rowptr, const at::Tensor& other,
                                                       const at::Tensor& perm, const c10::optional<at::Tensor>& w) {
  on_gpu(rowptr, "rowptr");
  TORCH_CHECK(rowptr.dim() == 1 && rowptr.size(0) >= 1, "pangnn::gcn_norm: rowptr must be [N + 1]");
  const int64_t n = rowptr.size(0) - 1, e = other.size(0);
  csr_operands("gcn_norm", rowptr, other, "other", rowptr, n);
  operand("gcn_norm", "perm", perm, rowptr, at::kInt);
  TORCH_CHECK(perm.dim() == 1 && perm.size(0) == e, "pangnn::gcn_norm: perm has ", perm.size(0), " entries for ", e, " edges");
  operand_any_float("gcn_norm", "edge_weight", w, rowptr);
  const DeviceGuard guard(rowptr.device());
  c10::optional<at::Tensor> wc;
  if (w.has_value() && w->defined()) {
    TORCH_CHECK(w->dim() == 1 && w->size(0) >= e, "pangnn::gcn_norm: edge_weight must be [E]");
    wc = w->to(at::kFloat).contiguous();
  }
  auto dis = at::empty({n}, rowptr.options().dtype(at::kFloat));
  auto ns = at::empty({e}, dis.options());
  auto no = at::empty({e}, dis.options());
  check_rc(pangnn_gcn_norm_f32(rowptr.data_ptr<int64_t>(), other.data_ptr<int32_t>(), perm.data_ptr<int32_t>(),
                               opt_ptr<float>(wc), n, e, dis.data_ptr<float>(), ns.data_ptr<float>(),
                               no.data_ptr<float>(), stream_of(rowptr)),
           "pangnn_gcn_norm_f32");
  return {dis, ns, no};
}

// spmm(rowptr, other, val?, x f32|bf16|f16 [n_src, F], bias?, n_rows) -> f32 [n_rows, F]
//   out[r] = bias + sum_{k in row r} val[k] x[other[k]]      forward propagate (by-target CSR) and, with the by-source
//   CSR, its transpose (spmm_bwd of SURVEY.md §8b is this op on the other CSR)
at::Tensor spmm(const at::Tensor& rowptr, const at::Tensor& other, const c10::optional<at::Tensor>& val,
                const at::Tensor& x, const c10::optional<at::Tensor>& bias, int64_t n_rows) {
  on_gpu(x, "x");
  TORCH_CHECK(x.dim() == 2 && x.is_floating_point(), "pangnn::spmm: x must be a floating-point [n_src, F]");
  csr_operands("spmm", rowptr, other, "other", x, n_rows);
  operand_any_float("spmm", "val", val, x);
  operand_any_float("spmm", "bias", bias, x);
  TORCH_CHECK(!(val.has_value() && val->defined()) || (val->dim() == 1 && val->size(0) == other.size(0)),
              "pangnn::spmm: val must have one entry per CSR entry (", other.size(0), ")");
  TORCH_CHECK(!(bias.has_value() && bias->defined()) || bias->numel() == x.size(1), "pangnn::spmm: bias must be [F]");
  const DeviceGuard guard(x.device());
  const int64_t f = x.size(1);
  const bool rows16 = is_rows16(x) && (f == 32 || f == 64 || f == 128 || f == 256);
  at::Tensor xc = rows16 ? x.contiguous() : x.to(at::kFloat).contiguous();
  c10::optional<at::Tensor> vc, bc;
  if (val.has_value() && val->defined()) vc = val->to(at::kFloat).contiguous();
  if (bias.has_value() && bias->defined()) bc = bias->to(at::kFloat).contiguous();
  auto out = at::empty({n_rows, f}, x.options().dtype(at::kFloat));
  const auto fn16 = xc.scalar_type() == at::kHalf ? pangnn_spmm_csr_f16 : pangnn_spmm_csr_bf16;
  const int rc = rows16 ? fn16(rowptr.data_ptr<int64_t>(), other.data_ptr<int32_t>(), opt_ptr<float>(vc), xc.data_ptr(),
                               xc.stride(0), xc.size(0), opt_ptr<float>(bc), out.data_ptr<float>(), out.stride(0), n_rows,
                               other.size(0), (int32_t)f, 0, stream_of(x))
                        : pangnn_spmm_csr_f32(rowptr.data_ptr<int64_t>(), other.data_ptr<int32_t>(), opt_ptr<float>(vc),
                                              xc.data_ptr<float>(), xc.stride(0), xc.size(0), opt_ptr<float>(bc),
                                              out.data_ptr<float>(), out.stride(0), n_rows, other.size(0), (int32_t)f, 0,
                                              stream_of(x));
  check_rc(rc, "pangnn_spmm_csr");
  return out;
}

// propagate(rowptr, other, val, rowptr_t, other_t, val_t, x, bias?) -> f32 [N, F]: GCNConv's message passing
// (PyG MessagePassing.propagate + GCNConv.message, gnn.py:158,165 call sites).  Forward = spmm over the by-target
// CSR; the by-source CSR (rowptr_t, other_t, val_t) rides along for the autograd formula registered from Python
// (dL/dx = spmm over it; the normalised weights are not differentiated: SURVEY.md §8 a6).
at::Tensor propagate(const at::Tensor& rowptr, const at::Tensor& other, const at::Tensor& val, const at::Tensor& rowptr_t,
                     const at::Tensor& other_t, const at::Tensor& val_t, const at::Tensor& x,
                     const c10::optional<at::Tensor>& bias) {
  on_gpu(x, "x");
  TORCH_CHECK(rowptr.dim() == 1 && rowptr.size(0) >= 1 && rowptr_t.dim() == 1 && rowptr_t.size(0) >= 1,
              "pangnn::propagate: row pointers must be [N + 1]");
  // the transposed triple is only used by the backward formula, but it is saved from here: fail now, not there
  csr_operands("propagate", rowptr_t, other_t, "other_t", x, rowptr_t.size(0) - 1);
  operand("propagate", "val_t", val_t, x, at::kFloat);
  TORCH_CHECK(val_t.size(0) == other_t.size(0) && other_t.size(0) == other.size(0),
              "pangnn::propagate: the two CSR orders hold different edge counts");
  return spmm(rowptr, other, val, x, bias, rowptr.size(0) - 1);
}

// edge_gather_concat(z f32[N,D], edge_index i64[2,E], extra f32[E]?) -> f32 [E, 2D (+1)]   (src/gnn.py:173-175)
at::Tensor edge_gather_concat(const at::Tensor& z, const at::Tensor& edge_index, const c10::optional<at::Tensor>& extra) {
  on_gpu(z, "z");
  on_gpu(edge_index, "edge_index");
  TORCH_CHECK(z.dim() == 2 && edge_index.dim() == 2 && edge_index.size(0) == 2 && edge_index.scalar_type() == at::kLong,
              "pangnn::edge_gather_concat: z [N, D], edge_index int64 [2, E]");
  TORCH_CHECK(edge_index.device() == z.device(), "pangnn::edge_gather_concat: edge_index is on ", edge_index.device(),
              " but the kernel runs on ", z.device());
  operand_any_float("edge_gather_concat", "extra", extra, z);
  TORCH_CHECK(!(extra.has_value() && extra->defined()) || extra->numel() >= edge_index.size(1),
              "pangnn::edge_gather_concat: extra must have one entry per edge");
  const DeviceGuard guard(z.device());
  const at::Tensor zc = z.to(at::kFloat).contiguous(), ei = edge_index.contiguous();
  c10::optional<at::Tensor> ex;
  if (extra.has_value() && extra->defined()) ex = extra->to(at::kFloat).contiguous();
  const int64_t e = ei.size(1), d = zc.size(1);
  auto out = at::empty({e, 2 * d + (ex.has_value() ? 1 : 0)}, zc.options());
  check_rc(pangnn_edge_gather_concat_f32(zc.data_ptr<float>(), zc.stride(0), zc.size(0), ei.data_ptr<int64_t>(), e, 0, e,
                                         opt_ptr<float>(ex), out.data_ptr<float>(), out.stride(0), (int32_t)d,
                                         stream_of(z)),
           "pangnn_edge_gather_concat_f32");
  return out;
}

// segment_sum_rows(rowptr, perm, m f32[E, W], col_off, f, n_rows) -> f32 [n_rows, f]: out[r] = sum_{k in row r} m[perm[k], col_off : col_off + f]
at::Tensor segment_sum_rows(const at::Tensor& rowptr, const at::Tensor& perm, const at::Tensor& m, int64_t col_off,
                            int64_t f, int64_t n_rows) {
  on_gpu(m, "m");
  TORCH_CHECK(m.dim() == 2 && col_off >= 0 && col_off + f <= m.size(1), "pangnn::segment_sum_rows: bad column window");
  csr_operands("segment_sum_rows", rowptr, perm, "perm", m, n_rows);
  TORCH_CHECK(perm.size(0) <= m.size(0), "pangnn::segment_sum_rows: perm addresses ", perm.size(0), " rows, m has ", m.size(0));
  const DeviceGuard guard(m.device());
  const at::Tensor mc = m.to(at::kFloat).contiguous();
  auto out = at::empty({n_rows, f}, mc.options());
  check_rc(pangnn_segment_sum_rows_f32(rowptr.data_ptr<int64_t>(), perm.data_ptr<int32_t>(), mc.data_ptr<float>(),
                                       mc.stride(0), mc.size(0), col_off, out.data_ptr<float>(), out.stride(0), n_rows,
                                       (int32_t)f, 0, stream_of(m)),
           "pangnn_segment_sum_rows_f32");
  return out;
}

// segment_max_rows(rowptr, perm, m f32[E, F], n_rows) -> (f32 [n_rows, F], arg i32 [n_rows, F])   aggr = 'max' (convolution.py:7)
std::tuple<at::Tensor, at::Tensor> segment_max_rows(const at::Tensor& rowptr, const at::Tensor& perm, const at::Tensor& m,
                                                    int64_t n_rows) {
  on_gpu(m, "m");
  TORCH_CHECK(m.dim() == 2, "pangnn::segment_max_rows: m must be [E, F]");
  csr_operands("segment_max_rows", rowptr, perm, "perm", m, n_rows);
  TORCH_CHECK(perm.size(0) <= m.size(0), "pangnn::segment_max_rows: perm addresses ", perm.size(0), " rows, m has ", m.size(0));
  const DeviceGuard guard(m.device());
  const at::Tensor mc = m.to(at::kFloat).contiguous();
  auto out = at::empty({n_rows, mc.size(1)}, mc.options());
  auto arg = at::empty({n_rows, mc.size(1)}, mc.options().dtype(at::kInt));
  check_rc(pangnn_segment_max_rows_f32(rowptr.data_ptr<int64_t>(), perm.data_ptr<int32_t>(), mc.data_ptr<float>(),
                                       mc.stride(0), out.data_ptr<float>(), arg.data_ptr<int32_t>(), out.stride(0), n_rows,
                                       (int32_t)mc.size(1), stream_of(m)),
           "pangnn_segment_max_rows_f32");
  return {out, arg};
}

// segment_max_bwd(g f32[n_rows, F], arg i32[n_rows, F], rowptr, num_edges) -> f32 [E, F]
at::Tensor segment_max_bwd(const at::Tensor& g, const at::Tensor& arg, const at::Tensor& rowptr, int64_t num_edges) {
  on_gpu(g, "g");
  TORCH_CHECK(g.dim() == 2 && num_edges >= 0, "pangnn::segment_max_bwd: g must be [n_rows, F]");
  operand("segment_max_bwd", "arg", arg, g, at::kInt);
  operand("segment_max_bwd", "rowptr", rowptr, g, at::kLong);
  TORCH_CHECK(arg.sizes() == g.sizes() && rowptr.size(0) >= g.size(0) + 1,
              "pangnn::segment_max_bwd: arg must have g's shape and rowptr n_rows + 1 entries");
  const DeviceGuard guard(g.device());
  const at::Tensor gc = g.to(at::kFloat).contiguous();
  auto gm = at::zeros({num_edges, gc.size(1)}, gc.options());
  check_rc(pangnn_segment_max_bwd_f32(gc.data_ptr<float>(), arg.data_ptr<int32_t>(), rowptr.data_ptr<int64_t>(),
                                      gm.data_ptr<float>(), gm.stride(0), gc.stride(0), gc.size(0), (int32_t)gc.size(1),
                                      stream_of(g)),
           "pangnn_segment_max_bwd_f32");
  return gm;
}

// ---------------------------------------------------------------------------------------------------------------
// Node-level dense layer (round 4: implementation AND autograd formula in C++ — no Python between the dispatcher and the
// kernels).  linear(x f32|bf16|f16 [N, K], w f32 [M, K], bias?, in_act, out_dtype) -> [N, M] of storage type out_dtype (PANGNN_DTYPE_*):  y = act(x) w^T + bias with
// in_act = 1: act = ELU applied to the rows on their way into LDS (src/gnn.py:108 folded into the consumer).  K, M in
// {64, 128} (pangnn_linear_supported; the Python layer sends every other shape to hipBLASLt).
// linear_backward(g, x, w, in_act, has_bias, need_dx) -> (dx like x | empty, dw f32 [M, K], db f32 [M] | empty):
// dx = (g w) * ELU'(x) in the epilogue when in_act (gate = x), dw = g^T act(x), db = column sums of g.
// ---------------------------------------------------------------------------------------------------------------
void linear_shapes(const char* op, const at::Tensor& x, const at::Tensor& w) {
  on_gpu(x, "x");
  TORCH_CHECK(x.dim() == 2 && w.dim() == 2 && x.is_floating_point() && w.is_floating_point() && x.size(1) == w.size(1),
              "pangnn::", op, ": x [N, K] and w [M, K] floating point, got ", x.sizes(), " and ", w.sizes());
  TORCH_CHECK(w.is_cuda() && w.device() == x.device(), "pangnn::", op, ": w is on ", w.device(), " but the kernel runs on ",
              x.device());
  TORCH_CHECK(pangnn_linear_supported((int32_t)w.size(1), (int32_t)w.size(0), 0), "pangnn::", op,
              ": K and M must be 64 or 128 (got ", w.size(1), ", ", w.size(0), ")");
}

at::Tensor linear_fwd(const at::Tensor& x, const at::Tensor& w, const c10::optional<at::Tensor>& bias, int64_t in_act,
                      int64_t out_dtype) {
  linear_shapes("linear", x, w);
  operand_any_float("linear", "bias", bias, x);
  TORCH_CHECK(!(bias.has_value() && bias->defined()) || bias->numel() == w.size(0), "pangnn::linear: bias must be [M]");
  const DeviceGuard guard(x.device());
  const at::Tensor xr = rows_any(x), wc = w.to(at::kFloat).contiguous();
  c10::optional<at::Tensor> bc;
  if (bias.has_value() && bias->defined()) bc = bias->to(at::kFloat).contiguous();
  const int64_t n = xr.size(0), k = xr.size(1), m = wc.size(0);
  auto y = at::empty({n, m}, x.options().dtype(scalar_of(out_dtype)));
  check_rc(pangnn_linear_act_fwd_mixed(xr.data_ptr(), dtype_code(xr), xr.stride(0), wc.data_ptr<float>(), opt_ptr<float>(bc),
                                       y.data_ptr(), dtype_code(y), y.stride(0), n, (int32_t)k, (int32_t)m, (int32_t)in_act,
                                       nullptr, 0, 0, stream_of(x)),
           "pangnn_linear_act_fwd_mixed");
  return y;
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> linear_bwd(const at::Tensor& g, const at::Tensor& x, const at::Tensor& w,
                                                         int64_t in_act, bool has_bias, bool need_dx) {
  linear_shapes("linear_backward", x, w);
  TORCH_CHECK(g.dim() == 2 && g.is_cuda() && g.device() == x.device() && g.is_floating_point() && g.size(0) == x.size(0) &&
                  g.size(1) == w.size(0),
              "pangnn::linear_backward: g must be a floating-point [N, M] on ", x.device());
  TORCH_CHECK(pangnn_linear_supported((int32_t)w.size(1), (int32_t)w.size(0), 1),
              "pangnn::linear_backward: shape not covered (pangnn_linear_supported)");
  const DeviceGuard guard(x.device());
  const at::Tensor xr = rows_any(x), gr = rows_any(g), wc = w.to(at::kFloat).contiguous();
  const int64_t n = xr.size(0), k = xr.size(1), m = wc.size(0);
  at::Tensor gx = at::empty({0}, x.options());
  if (need_dx) {
    // dx = g w, gated by ELU'(x): the kernel stages w [M, K] through the strides of its transpose (no copy of w^T)
    gx = at::empty({n, k}, xr.options());                           // stored like x (it is x's gradient)
    check_rc(pangnn_linear_dgrad_mixed(gr.data_ptr(), dtype_code(gr), gr.stride(0), wc.data_ptr<float>(), gx.data_ptr(),
                                       dtype_code(gx), gx.stride(0), n, (int32_t)k, (int32_t)m,
                                       in_act ? xr.data_ptr() : nullptr, dtype_code(xr), in_act ? xr.stride(0) : 0,
                                       stream_of(x)),
             "pangnn_linear_dgrad_mixed");
  }
  auto gw = at::empty({m, k}, wc.options());
  auto gb = at::empty({has_bias ? m : 0}, wc.options());
  const size_t wsb = pangnn_linear_wgrad_workspace_bytes((int32_t)k, (int32_t)m);
  auto ws = at::empty({(int64_t)wsb}, wc.options().dtype(at::kByte));
  check_rc(pangnn_linear_act_wgrad_mixed(gr.data_ptr(), dtype_code(gr), gr.stride(0), xr.data_ptr(), dtype_code(xr),
                                         xr.stride(0), n, (int32_t)k, (int32_t)m, (int32_t)in_act, gw.data_ptr<float>(),
                                         has_bias ? gb.data_ptr<float>() : nullptr, ws.data_ptr(), wsb, stream_of(x)),
           "pangnn_linear_act_wgrad_mixed");
  return {gx, gw, gb};
}

// autograd formula of pangnn::linear, registered under the Autograd key: forward redispatches below autograd, backward is
// ONE call of pangnn::linear_backward (itself a dispatcher op, so a tracer sees it)
at::Tensor call_linear(const at::Tensor& x, const at::Tensor& w, const c10::optional<at::Tensor>& bias, int64_t in_act,
                       int64_t out_dtype) {
  static auto op = c10::Dispatcher::singleton()
                       .findSchemaOrThrow("pangnn::linear", "")
                       .typed<at::Tensor(const at::Tensor&, const at::Tensor&, const c10::optional<at::Tensor>&, int64_t, int64_t)>();
  return op.call(x, w, bias, in_act, out_dtype);
}
std::tuple<at::Tensor, at::Tensor, at::Tensor> call_linear_backward(const at::Tensor& g, const at::Tensor& x,
                                                                   const at::Tensor& w, int64_t in_act, bool has_bias,
                                                                   bool need_dx) {
  static auto op = c10::Dispatcher::singleton()
                       .findSchemaOrThrow("pangnn::linear_backward", "")
                       .typed<std::tuple<at::Tensor, at::Tensor, at::Tensor>(const at::Tensor&, const at::Tensor&,
                                                                             const at::Tensor&, int64_t, bool, bool)>();
  return op.call(g, x, w, in_act, has_bias, need_dx);
}

class LinearFunction : public torch::autograd::Function<LinearFunction> {
 public:
  static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& x, const at::Tensor& w,
                            const c10::optional<at::Tensor>& bias, int64_t in_act, int64_t out_dtype) {
    at::AutoDispatchBelowADInplaceOrView below;
    ctx->save_for_backward({x, w});
    ctx->saved_data["in_act"] = in_act;
    ctx->saved_data["has_bias"] = bias.has_value() && bias->defined();
    return call_linear(x, w, bias, in_act, out_dtype);
  }
  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const at::Tensor &x = saved[0], &w = saved[1];
    const bool has_bias = ctx->saved_data["has_bias"].toBool(), need_dx = ctx->needs_input_grad(0);
    auto [gx, gw, gb] = call_linear_backward(grads[0], x, w, ctx->saved_data["in_act"].toInt(), has_bias, need_dx);
    return {need_dx ? gx : at::Tensor(), gw, has_bias ? gb : at::Tensor(), at::Tensor(), at::Tensor()};
  }
};

at::Tensor linear_autograd(const at::Tensor& x, const at::Tensor& w, const c10::optional<at::Tensor>& bias, int64_t in_act,
                           int64_t out_dtype) {
  return LinearFunction::apply(x, w, bias, in_act, out_dtype);
}

// ---------------------------------------------------------------------------------------------------------------
// Criterion: torch.nn.BCEWithLogitsLoss(pos_weight=class_balance), mean over `denom` edges (pangnn.py:98,203) —
// bce_with_logits(logits [E], y [E], pos_weight? (device scalar), denom) -> (loss [], dL/dlogits [E]) in ONE pass
// (pangnn_bce_logits_f32); the autograd formula hands dL/dlogits on, scaled by the upstream gradient of the loss.
// ---------------------------------------------------------------------------------------------------------------
std::tuple<at::Tensor, at::Tensor> bce_fwd(const at::Tensor& logits, const at::Tensor& y, const c10::optional<at::Tensor>& pw,
                                           int64_t denom) {
  on_gpu(logits, "logits");
  TORCH_CHECK(logits.dim() == 1 && y.dim() == 1 && y.size(0) == logits.size(0) && logits.is_floating_point() &&
                  y.is_floating_point() && y.is_cuda() && y.device() == logits.device(),
              "pangnn::bce_with_logits: logits and y must be floating-point [E] on one GPU");
  operand_any_float("bce_with_logits", "pos_weight", pw, logits);
  TORCH_CHECK(denom >= 0, "pangnn::bce_with_logits: denom >= 0");
  const DeviceGuard guard(logits.device());
  const at::Tensor x = logits.to(at::kFloat).contiguous(), yy = y.to(at::kFloat).contiguous();
  c10::optional<at::Tensor> pc;
  if (pw.has_value() && pw->defined()) pc = pw->to(at::kFloat).contiguous().reshape({-1});
  auto loss = at::empty(at::IntArrayRef{}, x.options());          // 0-dim (NOT loss.view({}): `{}` also converts to a ScalarType)
  auto g = at::empty_like(x);
  const size_t wsb = pangnn_bce_logits_workspace_bytes();
  auto ws = at::empty({(int64_t)wsb}, x.options().dtype(at::kByte));
  check_rc(pangnn_bce_logits_f32(x.data_ptr<float>(), yy.data_ptr<float>(), opt_ptr<float>(pc), x.size(0), denom,
                                 loss.data_ptr<float>(), g.data_ptr<float>(), ws.data_ptr(), wsb, stream_of(logits)),
           "pangnn_bce_logits_f32");
  return {loss, g};
}

class BceFunction : public torch::autograd::Function<BceFunction> {
 public:
  static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const at::Tensor& logits,
                                                const at::Tensor& y, const c10::optional<at::Tensor>& pw, int64_t denom) {
    at::AutoDispatchBelowADInplaceOrView below;
    static auto op = c10::Dispatcher::singleton()
                         .findSchemaOrThrow("pangnn::bce_with_logits", "")
                         .typed<std::tuple<at::Tensor, at::Tensor>(const at::Tensor&, const at::Tensor&,
                                                                   const c10::optional<at::Tensor>&, int64_t)>();
    auto [loss, g] = op.call(logits, y, pw, denom);
    ctx->save_for_backward({g});
    ctx->mark_non_differentiable({g});
    return {loss, g};
  }
  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
    const at::Tensor g = ctx->get_saved_variables()[0];
    at::Tensor gl;
    if (grads[0].defined()) gl = g * grads[0];
    return {gl, at::Tensor(), at::Tensor(), at::Tensor()};
  }
};

std::tuple<at::Tensor, at::Tensor> bce_autograd(const at::Tensor& logits, const at::Tensor& y,
                                                const c10::optional<at::Tensor>& pw, int64_t denom) {
  auto out = BceFunction::apply(logits, y, pw, denom);
  return {out[0], out[1]};
}

// ---------------------------------------------------------------------------------------------------------------
// Autograd formulas of the two graph ops that are pure functions of their tensor operands (round 4: C++, like linear /
// bce_with_logits — what is left to Python are the formulas of ops that look a cached structure up by edge_index).
//   propagate:        dL/dx = spmm over the by-source CSR triple that rides along (the normalised weights are not
//                     differentiated: SURVEY.md §8 a6), dL/dbias = column sums of g
//   segment_max_rows: dL/dm = scatter of g to the arg-max entries (pangnn::segment_max_bwd)
// Both backward formulas call registered ops, so a tracer sees them.
// ---------------------------------------------------------------------------------------------------------------

class PropagateFunction : public torch::autograd::Function<PropagateFunction> {
 public:
  static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& rowptr, const at::Tensor& other,
                            const at::Tensor& val, const at::Tensor& rowptr_t, const at::Tensor& other_t,
                            const at::Tensor& val_t, const at::Tensor& x, const c10::optional<at::Tensor>& bias) {
    at::AutoDispatchBelowADInplaceOrView below;
    ctx->save_for_backward({rowptr_t, other_t, val_t});
    ctx->saved_data["n_src"] = x.size(0);
    ctx->saved_data["x_dtype"] = (int64_t)x.scalar_type();
    ctx->saved_data["has_bias"] = bias.has_value() && bias->defined();
    static auto op = typed_op<at::Tensor(const at::Tensor&, const at::Tensor&, const at::Tensor&, const at::Tensor&,
                                         const at::Tensor&, const at::Tensor&, const at::Tensor&,
                                         const c10::optional<at::Tensor>&)>("pangnn::propagate");
    return op.call(rowptr, other, val, rowptr_t, other_t, val_t, x, bias);
  }
  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const at::Tensor& g = grads[0];
    at::Tensor gx, gb;
    if (ctx->needs_input_grad(6)) {
      static auto op = typed_op<at::Tensor(const at::Tensor&, const at::Tensor&, const c10::optional<at::Tensor>&,
                                           const at::Tensor&, const c10::optional<at::Tensor>&, int64_t)>("pangnn::spmm");
      gx = op.call(saved[0], saved[1], saved[2], g, c10::nullopt, ctx->saved_data["n_src"].toInt());
      const auto dt = (at::ScalarType)ctx->saved_data["x_dtype"].toInt();
      if (gx.scalar_type() != dt) gx = gx.to(dt);
    }
    // fp32 column sums whatever g is stored as (a bfloat16 g under autocast: the ctypes route's colsum kernel sums in fp32 too)
    if (ctx->saved_data["has_bias"].toBool() && ctx->needs_input_grad(7)) gb = at::sum(g, {0}, false, at::kFloat);
    return {at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), gx, gb};
  }
};

at::Tensor propagate_autograd(const at::Tensor& rowptr, const at::Tensor& other, const at::Tensor& val,
                              const at::Tensor& rowptr_t, const at::Tensor& other_t, const at::Tensor& val_t,
                              const at::Tensor& x, const c10::optional<at::Tensor>& bias) {
  return PropagateFunction::apply(rowptr, other, val, rowptr_t, other_t, val_t, x, bias);
}

class SegmentMaxFunction : public torch::autograd::Function<SegmentMaxFunction> {
 public:
  static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const at::Tensor& rowptr,
                                                const at::Tensor& perm, const at::Tensor& m, int64_t n_rows) {
    at::AutoDispatchBelowADInplaceOrView below;
    static auto op = typed_op<std::tuple<at::Tensor, at::Tensor>(const at::Tensor&, const at::Tensor&, const at::Tensor&,
                                                                 int64_t)>("pangnn::segment_max_rows");
    auto [out, arg] = op.call(rowptr, perm, m, n_rows);
    ctx->save_for_backward({arg, rowptr});
    ctx->saved_data["e"] = m.size(0);
    ctx->mark_non_differentiable({arg});
    return {out, arg};
  }
  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    at::Tensor gm;
    if (grads[0].defined()) {
      static auto op = typed_op<at::Tensor(const at::Tensor&, const at::Tensor&, const at::Tensor&, int64_t)>(
          "pangnn::segment_max_bwd");
      gm = op.call(grads[0], saved[0], saved[1], ctx->saved_data["e"].toInt());
    }
    return {at::Tensor(), at::Tensor(), gm, at::Tensor()};
  }
};

std::tuple<at::Tensor, at::Tensor> segment_max_autograd(const at::Tensor& rowptr, const at::Tensor& perm, const at::Tensor& m,
                                                        int64_t n_rows) {
  auto out = SegmentMaxFunction::apply(rowptr, perm, m, n_rows);
  return {out[0], out[1]};
}

}  // namespace

TORCH_LIBRARY(pangnn, m) {
  // the C ABI this file was compiled against (include/pangnn_hip.h) must be the one the loaded libpangnn_hip.so implements:
  // signatures changed between versions, and a stale library selected by path would read pointers as sizes
  TORCH_CHECK(pangnn_abi_version() == PANGNN_ABI_VERSION, "libpangnn_torch.so was built against C ABI version ",
              PANGNN_ABI_VERSION, " but the loaded libpangnn_hip.so reports ", pangnn_abi_version(),
              " — rebuild both from one tree (make -C pangnn_amd/csrc)");
  m.def("csr_from_coo(Tensor edge_index, int num_nodes, int group_by) -> (Tensor, Tensor, Tensor)");
  m.def("gcn_norm(Tensor rowptr, Tensor other, Tensor perm, Tensor? edge_weight) -> (Tensor, Tensor, Tensor)");
  m.def("spmm(Tensor rowptr, Tensor other, Tensor? val, Tensor x, Tensor? bias, int n_rows) -> Tensor");
  m.def("propagate(Tensor rowptr, Tensor other, Tensor val, Tensor rowptr_t, Tensor other_t, Tensor val_t, Tensor x, "
        "Tensor? bias) -> Tensor");
  m.def("edge_gather_concat(Tensor z, Tensor edge_index, Tensor? extra) -> Tensor");
  m.def("segment_sum_rows(Tensor rowptr, Tensor perm, Tensor m, int col_off, int f, int n_rows) -> Tensor");
  m.def("segment_max_rows(Tensor rowptr, Tensor perm, Tensor m, int n_rows) -> (Tensor, Tensor)");
  m.def("segment_max_bwd(Tensor g, Tensor arg, Tensor rowptr, int num_edges) -> Tensor");
  m.def("linear(Tensor x, Tensor w, Tensor? bias, int in_act, int out_dtype) -> Tensor");
  m.def("linear_backward(Tensor g, Tensor x, Tensor w, int in_act, bool has_bias, bool need_dx) -> (Tensor, Tensor, Tensor)");
  m.def("bce_with_logits(Tensor logits, Tensor y, Tensor? pos_weight, int denom) -> (Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(pangnn, CUDA, m) {       // "CUDA" is the dispatch key of HIP tensors on ROCm builds of torch
  m.impl("csr_from_coo", &csr_from_coo);
  m.impl("gcn_norm", &gcn_norm);
  m.impl("spmm", &spmm);
  m.impl("propagate", &propagate);
  m.impl("edge_gather_concat", &edge_gather_concat);
  m.impl("segment_sum_rows", &segment_sum_rows);
  m.impl("segment_max_rows", &segment_max_rows);
  m.impl("segment_max_bwd", &segment_max_bwd);
  m.impl("linear", &linear_fwd);
  m.impl("linear_backward", &linear_bwd);
  m.impl("bce_with_logits", &bce_fwd);
}

TORCH_LIBRARY_IMPL(pangnn, Autograd, m) {   // autograd formulas that live in C++ (the others are registered from Python)
  m.impl("linear", &linear_autograd);
  m.impl("bce_with_logits", &bce_autograd);
  m.impl("propagate", &propagate_autograd);
  m.impl("segment_max_rows", &segment_max_autograd);
}
