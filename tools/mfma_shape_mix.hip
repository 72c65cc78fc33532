// gfx950 micro-benchmark (round 5): does the MFMA SHAPE matter for a kernel that alternates a matrix burst with a vector
// burst, at 1 / 2 / 4 waves per SIMD?  The decoder's T kernel issues, per 32-edge tile and wave, 48 v_mfma_f32_16x16x32_bf16
// (s_setprio 1 around them) and ~276 other vector instructions; the same product on v_mfma_f32_32x32x16_bf16 is 24
// instructions of twice the flops.  Both shapes deliver the same flops per matrix-pipe cycle (16 vs 32 cycles per
// instruction); what differs is the VALU issue time an MFMA holds (8 cycles either way: 384 vs 192 per tile).
// Two cost models were on the table after round 4: "issue slots" (4 x VALU + 8 x MFMA: the 32x32 form saves
// 192 of ~1 490 cycles) and "additive" (4 x VALU + matrix-pipe cycles: the shape changes nothing).  This program measures it:
// (cycles at the NOMINAL clock from event times, and shader-clock cycles from s_memtime: the clock a mix sustains differs)
// per iteration a burst of NM matrix instructions (operands in registers, accumulator chains as in the kernel: 8 chains of 6
// for the 16x16 form, 2 chains of 12 for the 32x32 form) between s_setprio 1 / 0, then NV independent v_fma_f32.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_mix.hip -o /tmp/mfma_shape_mix && /tmp/mfma_shape_mix
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int NV, bool PRIO, int IL>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float a, float b, int stagger, unsigned long long* cyc) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  bf16x8 A[2], B[3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) A[i][j] = (__bf16)(float)((threadIdx.x + i + j) & 3);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) B[i][j] = (__bf16)(float)((threadIdx.x * 3 + i + j) & 1);
  f32x4 c16[8];
  f32x16 c32[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) c16[i] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) c32[i][j] = 0.f;
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
  // stagger: every second wave of a SIMD (waves w and w + 4 of a workgroup share SIMD w % 4) starts half an iteration late —
  // one vector burst first — so that one wave's matrix burst meets its neighbour's vector burst, as the phases of a real
  // kernel's waves drift apart; without it all waves of a SIMD run their bursts in phase
  if (stagger && ((threadIdx.x >> 8) & 1)) {
#pragma unroll
    for (int v = 0; v < NV; ++v) x[v & 15] = __builtin_fmaf(x[v & 15], a, b);
  }
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_sched_barrier(0);
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    int vdone = 0;
    if (SHAPE == 0) {
#pragma unroll
      for (int s = 0; s < 6; ++s)
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
          c16[ch] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ch & 1], B[s % 3], c16[ch], 0, 0, 0);
          if (IL > 0) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < IL; ++q)
              if ((s * 8 + ch) * IL + q < NV) x[((s * 8 + ch) * IL + q) & 15] = __builtin_fmaf(x[((s * 8 + ch) * IL + q) & 15], a, b);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      vdone = 48 * IL;
    } else {
#pragma unroll
      for (int s = 0; s < 12; ++s)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          c32[ch] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ch], B[s % 3], c32[ch], 0, 0, 0);
          if (IL > 0) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 2 * IL; ++q)          // the same VALU per matrix-pipe cycle as the 16x16 form: 2 IL per instruction
              if ((s * 2 + ch) * 2 * IL + q < NV) x[((s * 2 + ch) * 2 * IL + q) & 15] = __builtin_fmaf(x[((s * 2 + ch) * 2 * IL + q) & 15], a, b);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      vdone = 48 * IL;
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < NV; ++v)
      if (v >= vdone) x[v & 15] = __builtin_fmaf(x[v & 15], a, b);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += c16[i][0] + c16[i][3];
#pragma unroll
  for (int i = 0; i < 2; ++i) s += c32[i][0] + c32[i][15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  // shader-clock cycles of this wave's lifetime (s_memtime): the REAL clock under this instruction mix, not the nominal one
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = __builtin_readcyclecounter() - t0;
}

template <int SHAPE, int NV, int IL>
static double run(float* out, int cus, int waves_per_simd, int iters, bool prio, int stagger, double* real_cyc) {
  static unsigned long long* cyc = nullptr;
  if (!cyc) (void)hipMalloc(&cyc, 16 * sizeof(unsigned long long));
  const int threads = 64 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  auto launch = [&]() {
    if (prio) hipLaunchKernelGGL((k<SHAPE, NV, true, IL>), dim3(cus), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f, stagger, cyc);
    else hipLaunchKernelGGL((k<SHAPE, NV, false, IL>), dim3(cus), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f, stagger, cyc);
  };
  launch();
  (void)hipDeviceSynchronize();
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  unsigned long long h[16];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double sum = 0;
  for (int w = 0; w < 4 * waves_per_simd; ++w) sum += (double)h[w];
  // the waves of one SIMD run side by side: SIMD time per (wave, tile) = a wave's lifetime / (tiles x waves per SIMD)
  *real_cyc = sum / (4.0 * waves_per_simd) / ((double)iters * waves_per_simd);
  return best;
}

template <int NV, int IL>
static void row(float* out, int cus, int clk_khz) {
  const int iters = 4000;
  for (int w : {1, 2, 4})
    for (int stagger = 0; stagger <= (w > 1 ? 1 : 0); ++stagger)
      for (int prio = 1; prio >= 0; --prio) {
        double ra = 0, rb = 0;
        const double a = run<0, NV, IL>(out, cus, w, iters, prio, stagger, &ra), b = run<1, NV, IL>(out, cus, w, iters, prio, stagger, &rb);
        // cycles of SIMD time per wave-iteration (one "tile"): kernel time x clock / (iterations x waves on the SIMD)
        const double ca = a * 1e-3 * clk_khz * 1e3 / ((double)iters * w), cb = b * 1e-3 * clk_khz * 1e3 / ((double)iters * w);
        printf("VALU %3d (%d / %d behind each MFMA)  waves/SIMD %d  stagger %d  setprio %d :  48 x 16x16x32  %7.1f cyc/tile   24 x 32x32x16  %7.1f cyc/tile   ratio "
               "%.3f | s_memtime cycles/tile %7.1f  %7.1f  (clock %.2f / %.2f GHz)\n",
               NV, IL, 2 * IL, w, stagger, prio, ca, cb, cb / ca, ra, rb, ra / ca * clk_khz / 1e6, rb / cb * clk_khz / 1e6);
      }
}

int main() {
  int cus = 256, clk_khz = 0;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  float* out;
  (void)hipMalloc(&out, (size_t)cus * 1024 * sizeof(float));
  printf("# %d CUs, nominal clock %.2f GHz; cycles are nominal-clock cycles of SIMD time per wave and tile\n", cus, clk_khz / 1e6);
  row<0, 0>(out, cus, clk_khz);
  row<180, 0>(out, cus, clk_khz);     // the inference kernel's vector work per 16 edges beside 48 matrix instructions
  row<276, 0>(out, cus, clk_khz);     // the T kernel's per 32-edge tile: bursts
  row<276, 1>(out, cus, clk_khz);     // ... with 1 (16x16) / 2 (32x32) of them behind every matrix instruction
  row<276, 2>(out, cus, clk_khz);     // 2 / 4
  row<276, 3>(out, cus, clk_khz);     // 3 / 6
  row<400, 0>(out, cus, clk_khz);
  row<400, 3>(out, cus, clk_khz);
  return 0;
}
