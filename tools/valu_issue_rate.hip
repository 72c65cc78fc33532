// gfx950 micro-benchmark: vector-instruction issue rate of ONE SIMD as a function of the number of resident waves.
// Every wave runs a long stream of independent v_fma_f32 (16 accumulators, no memory traffic); the grid puts `w` waves
// on every SIMD of every CU.  Prints cycles per wave-instruction per SIMD (kernel time x clock / instructions per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_issue_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);                      // v_fma_f32
      if (MODE == 1) x[i] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x[i]) & 0xffff0fffu) - b;   // v_and + v_sub
      if (MODE == 2) x[i] = __builtin_bit_cast(float, __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x[i]), __builtin_bit_cast(unsigned, x[(i + 1) & 15]), 0x07060302u));
      if (MODE == 3) x[0] = __builtin_fmaf(x[0], a, b);                      // one dependent chain of v_fma_f32
      if (MODE == 4) {                                                       // v_cvt_pk_bf16_f32
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        asm volatile("" : "+v"(x[i]));
        x[i] = __builtin_bit_cast(float, __builtin_convertvector((f2){x[i], x[(i + 1) & 15]}, b2));
      }
      if (MODE == 5 && (i & 1) == 0) {                                       // v_pk_add_f32 (two floats per lane)
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 v = {x[i], x[i + 1]};
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"((f2){a, b}));
        x[i] = v[0]; x[i + 1] = v[1];
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  int clk_khz = 0;
  (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  float* out;
  (void)hipMalloc(&out, (size_t)cus * 2048 * sizeof(float));
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int mode = 0; mode < 6; ++mode)
    for (int waves_per_simd : {1, 2, 3, 4}) {
      const int threads = 64 * 4 * waves_per_simd;            // one block per CU
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
        if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
        if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
      };
      launch();
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      launch();
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double inst_per_simd = (double)iters * (mode == 1 ? 32 : mode == 5 ? 8 : 16) * waves_per_simd;
      static const char* names[] = {"v_fma_f32", "v_and+v_sub", "v_perm_b32", "v_fma_f32, one dependent chain", "v_cvt_pk_bf16_f32", "v_pk_add_f32"};
      printf("mode %d (%s) waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD at %.2f GHz nominal\n", mode,
             names[mode], waves_per_simd, ms,
             ms * 1e-3 * clk_khz * 1e3 / inst_per_simd, clk_khz / 1e6);
    }
  return 0;
}
