// Diagnostic: do v_mfma_f32_32x32x2_f32 (wave A) and plain f32 VALU work (wave B on the same SIMD) overlap?
// 512-thread blocks, one per CU: waves 0-3 run MFMAs, waves 4-7 run dependent-free v_fma_f32 streams.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>   // 0: f32 MFMA, 1: bf16 MFMA
__global__ __launch_bounds__(512) void k(float* out, int mfma_iters, int valu_iters, float a, float b) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float s = 0.f;
  if (wave < 4) {
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)a; bb[i] = (__bf16)b; }
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (MODE == 0) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b + c, acc[c], 0, 0, 0);
          else acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[c], 0, 0, 0);
        }
    }
    for (int c = 0; c < 4; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  } else {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = a + i;
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], b, a);
    }
    for (int i = 0; i < 16; ++i) s += v[i];
  }
  if (s == 1234.5f) out[threadIdx.x] = s;
}
template <int MODE>
float run(int mi, int vi) {
  float* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, 10, 10, 1.f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, mi, vi, 1.f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  // 4000 iters x 32 f32 MFMAs = 128000 x 64 cyc = 8.2M cycles; VALU: iters x 64 fma x 4 cyc
  const int mi = 4000, vi = 32000;   // 32000 x 64 x 4 = 8.2M cycles
  printf("f32 MFMA only      : %.3f ms\n", run<0>(mi, 0));
  printf("VALU only          : %.3f ms\n", run<0>(0, vi));
  printf("f32 MFMA + VALU    : %.3f ms\n", run<0>(mi, vi));
  printf("bf16 MFMA only (x2 iters): %.3f ms\n", run<1>(2 * mi, 0));
  printf("bf16 MFMA + VALU   : %.3f ms\n", run<1>(2 * mi, vi));
  return 0;
}
