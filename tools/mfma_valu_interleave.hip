// Diagnostic: one wave per SIMD; between consecutive MFMAs of the SAME wave, F independent v_fma_f32 are placed.
// How many fit into an MFMA's shadow?  (f32: 32x32x2, 64 cycles; bf16: 32x32x16, 32 cycles)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE, int F>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)a; bb[i] = (__bf16)b; }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (MODE == 0) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b + c, acc[c], 0, 0, 0);
        else acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[c], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < F; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[f % 16]) : "v"(b), "v"(a));
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  float s = 0.f;
  for (int c = 0; c < 4; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 1234.5f) out[threadIdx.x] = s;
}
template <int MODE, int F>
void run() {
  float* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  hipLaunchKernelGGL((k<MODE, F>), dim3(256), dim3(256), 0, 0, out, 10, 1.f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, F>), dim3(256), dim3(256), 0, 0, out, iters, 1.f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s F=%2d: %.3f ms  = %.1f ns per (MFMA + F fma)\n", MODE ? "bf16 32x32x16" : "f32  32x32x2 ", F, ms, ms * 1e6 / (iters * 32.0));
}
int main() {
  run<0, 0>(); run<0, 4>(); run<0, 8>(); run<0, 12>(); run<0, 14>(); run<0, 16>(); run<0, 24>();
  run<1, 0>(); run<1, 2>(); run<1, 4>(); run<1, 6>(); run<1, 8>(); run<1, 12>();
  return 0;
}
