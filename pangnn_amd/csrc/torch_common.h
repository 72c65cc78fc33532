// Helpers shared by the two translation units of libpangnn_torch.so (torch_ops.cpp: the tensor ops; graph_ops.cpp: the
// structure registry and the per-step ops that read a graph): stream / device guard / return-code / operand checks.
#pragma once
#include <ATen/ATen.h>
#include <ATen/core/dispatch/Dispatcher.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include "../../include/pangnn_hip.h"

namespace pangnn_torch {

inline void* stream_of(const at::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

inline void check_rc(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed (rc=", rc, "): ", pangnn_last_error()); }

inline const at::Tensor& on_gpu(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), "pangnn: ", name, " must be a GPU tensor (there is no CPU path), got ", t.device());
  return t;
}

// Every implementation forwards raw data_ptr()s to a kernel launched on `ref`'s device: each operand must live
// there (a host pointer would be a GPU memory fault, not an exception) and have the dtype the C ABI reads it as.
using DeviceGuard = c10::hip::OptionalHIPGuardMasqueradingAsCUDA;

inline void operand(const char* op, const char* name, const at::Tensor& t, const at::Tensor& ref, at::ScalarType dtype) {
  TORCH_CHECK(t.defined(), "pangnn::", op, ": ", name, " is undefined");
  TORCH_CHECK(t.is_cuda() && t.device() == ref.device(), "pangnn::", op, ": ", name, " is on ", t.device(),
              " but the kernel runs on ", ref.device(), " (every operand must be on that GPU)");
  TORCH_CHECK(t.scalar_type() == dtype, "pangnn::", op, ": ", name, " must be ", dtype, ", got ", t.scalar_type());
  TORCH_CHECK(t.is_contiguous(), "pangnn::", op, ": ", name, " must be contiguous");
}

inline void operand_any_float(const char* op, const char* name, const c10::optional<at::Tensor>& t, const at::Tensor& ref) {
  if (!t.has_value() || !t->defined()) return;
  TORCH_CHECK(t->is_cuda() && t->device() == ref.device(), "pangnn::", op, ": ", name, " is on ", t->device(),
              " but the kernel runs on ", ref.device(), " (every operand must be on that GPU)");
  TORCH_CHECK(t->is_floating_point(), "pangnn::", op, ": ", name, " must be a floating-point tensor");
}

// CSR triple of one order: rowptr int64 [>= n_rows + 1], ids int32 [E]; E and the row pointer's last entry are the
// caller's contract (graph.build_csr validates them once per graph; reading rowptr back here would be a host sync)
inline void csr_operands(const char* op, const at::Tensor& rowptr, const at::Tensor& ids, const char* ids_name,
                  const at::Tensor& ref, int64_t n_rows) {
  operand(op, "rowptr", rowptr, ref, at::kLong);
  operand(op, ids_name, ids, ref, at::kInt);
  TORCH_CHECK(rowptr.dim() == 1 && ids.dim() == 1, "pangnn::", op, ": rowptr and ", ids_name, " must be 1-D");
  TORCH_CHECK(n_rows >= 0 && rowptr.size(0) >= n_rows + 1, "pangnn::", op, ": rowptr has ", rowptr.size(0),
              " entries for ", n_rows, " rows");
}

template <typename T>
const T* opt_ptr(const c10::optional<at::Tensor>& t) {
  return (t.has_value() && t->defined()) ? t->data_ptr<T>() : nullptr;
}

// PANGNN_DTYPE_* of a row tensor: bfloat16 and float16 rows are read / written as stored, everything else is f32
inline bool is_rows16(const at::Tensor& t) { return t.scalar_type() == at::kBFloat16 || t.scalar_type() == at::kHalf; }
inline int32_t dtype_code(const at::Tensor& t) {
  return t.scalar_type() == at::kBFloat16 ? PANGNN_DTYPE_BF16 : t.scalar_type() == at::kHalf ? PANGNN_DTYPE_F16 : PANGNN_DTYPE_F32;
}
inline at::ScalarType scalar_of(int64_t code) {
  TORCH_CHECK(code == PANGNN_DTYPE_F32 || code == PANGNN_DTYPE_BF16 || code == PANGNN_DTYPE_F16,
              "pangnn: storage type code ", code, " (0 float32, 1 bfloat16, 2 float16)");
  return code == PANGNN_DTYPE_BF16 ? at::kBFloat16 : code == PANGNN_DTYPE_F16 ? at::kHalf : at::kFloat;
}

// rows as the kernels read them: f32 (any other float is converted) or bfloat16 / float16 as stored, unit column stride, a row stride
// that keeps 16-byte loads aligned — column windows of a wider matrix pass through without a copy
inline at::Tensor rows_any(const at::Tensor& t) {
  if (is_rows16(t)) {
    if (t.dim() == 2 && t.stride(1) == 1 && t.stride(0) % 8 == 0 && t.stride(0) >= t.size(1) &&
        reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0)
      return t;
    return t.contiguous();
  }
  const at::Tensor f = t.scalar_type() == at::kFloat ? t : t.to(at::kFloat);
  if (f.dim() == 2 && f.stride(1) == 1 && f.stride(0) % 4 == 0 && f.stride(0) >= f.size(1) &&
      reinterpret_cast<uintptr_t>(f.data_ptr()) % 16 == 0)
    return f;
  return f.contiguous();
}


template <typename Sig>
auto typed_op(const char* name) {
  return c10::Dispatcher::singleton().findSchemaOrThrow(name, "").typed<Sig>();
}

}  // namespace pangnn_torch
