// Diagnostic: issue rate of v_mfma_f32_32x32x2_f32 with every CU busy.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32 / CHAINS; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b + c, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  if (s == 1234.5f) out[threadIdx.x] = s;
}
template <int CHAINS>
void run(int blocks, int threads, const char* name) {
  float* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mfma_per_wave = 32.0 * iters;
  double waves = (double)blocks * threads / 64;
  double tf = waves * mfma_per_wave * 4096 / (ms * 1e-3) / 1e12;
  printf("%s chains=%d blocks=%d threads=%d: %.3f ms, %.1f ns per MFMA per wave, %.1f TFLOP/s\n", name, CHAINS, blocks, threads, ms,
         ms * 1e6 / mfma_per_wave, tf);
}
int main() {
  run<4>(256, 256, "1 wave/SIMD");
  run<1>(256, 256, "1 wave/SIMD");
  run<4>(512, 256, "2 waves/SIMD");
  run<4>(64, 256, "quarter chip");
  run<4>(256, 64, "1 wave/CU");
  return 0;
}
