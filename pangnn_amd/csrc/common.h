// Shared helpers for libpangnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/pangnn_hip.h"

namespace pangnn {

void set_error(const char* fmt, ...);

#define PG_CHECK_ARG(cond, code, ...)            \
  do {                                           \
    if (!(cond)) {                               \
      ::pangnn::set_error(__VA_ARGS__);          \
      return (code);                             \
    }                                            \
  } while (0)

#define PG_CHECK_LAUNCH(name)                                                          \
  do {                                                                                 \
    hipError_t e__ = hipGetLastError();                                                \
    if (e__ != hipSuccess) {                                                           \
      ::pangnn::set_error("%s: launch failed: %s", (name), hipGetErrorString(e__));    \
      return (int)e__;                                                                 \
    }                                                                                  \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int kWave = 64;          // CDNA4 wavefront
constexpr int kBlock = 256;        // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;

// Second stage of the two-stage reductions (per-workgroup partial rows -> one row), reproducible and short:
// a 1024-thread block owns 64 consecutive elements; wave g adds parts g, g+16, g+32, ... in that order, then the
// 16 wave sums are added in wave order.  (One thread walking all parts serially took 60-240 us per call.)
// Launch with kSumThreads threads and (len + 63) / 64 blocks; the result is valid in wave 0 (threadIdx.x < 64).
constexpr int kSumGroups = 16;
constexpr int kSumThreads = kSumGroups * kWave;
#ifdef __HIPCC__
// ---- 2-byte row formats (storage only; every sum and product is fp32).  A kernel templated on its storage type uses
// `unsigned short` for bfloat16 bits (bf16 -> f32 is a shift) and `_Float16` for IEEE half (`--mixed_precision fp16`,
// src/setup.py:50; v_cvt_f32_f16 / v_cvt_f16_f32); RowFmt<T>::value is the PANGNN_DTYPE_* code of T.
template <typename T> struct RowFmt { static constexpr int value = 0; };
template <> struct RowFmt<unsigned short> { static constexpr int value = 1; };
template <> struct RowFmt<_Float16> { static constexpr int value = 2; };
// four consecutive stored elements (8 bytes) -> f32, exact
__device__ __forceinline__ float4 rows16_to_f32(uint2 t, int fmt) {
  if (fmt == 2) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 a = __builtin_bit_cast(h2, t.x), b = __builtin_bit_cast(h2, t.y);
    return make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
  }
  return make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u),
                     __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u));
}
__device__ __forceinline__ float row16_to_f32(unsigned short bits, int fmt) {
  return fmt == 2 ? (float)__builtin_bit_cast(_Float16, bits) : __uint_as_float((uint32_t)bits << 16);
}
// f32 -> four stored elements, each rounded to nearest even
__device__ __forceinline__ uint2 f32_to_rows16(float v0, float v1, float v2, float v3, int fmt) {
  if (fmt == 2) {
    // f32 values first, then one rounding each (no v_fma_mixlo_f16 folding of the producing multiply-add: see linear.hip)
    asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    const _Float16 o[4] = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
    return *reinterpret_cast<const uint2*>(o);
  }
  const __bf16 o[4] = {(__bf16)v0, (__bf16)v1, (__bf16)v2, (__bf16)v3};
  return *reinterpret_cast<const uint2*>(o);
}

__device__ __forceinline__ float ordered_parts_sum(const float* __restrict__ part, int n_parts, int64_t stride, int i,
                                                   int len) {
  __shared__ float red[kSumGroups][kWave];
  const int e = threadIdx.x & (kWave - 1), g = threadIdx.x >> 6;
  float s = 0.f;
  if (i < len) {
    // same order of additions as the plain loop; eight loads in flight instead of one dependent load-add per L2 round trip
    // (the one-workgroup finish of 2048 x 64 band partials took 35 us)
    int p = g;
    for (; p + 7 * kSumGroups < n_parts; p += 8 * kSumGroups) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(p + u * kSumGroups) * stride + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; p < n_parts; p += kSumGroups) s += part[(int64_t)p * stride + i];
  }
  red[g][e] = s;
  __syncthreads();
  float t = 0.f;
  if (g == 0)
    for (int k = 0; k < kSumGroups; ++k) t += red[k][e];
  return t;
}
#endif

}  // namespace pangnn
