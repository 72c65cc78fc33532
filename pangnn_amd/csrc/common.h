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
