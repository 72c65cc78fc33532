// gfx950 micro-benchmark: ways of packing the HIGH halves of two registers into one (the bf16 term packing of the exact
// three-term splits: 48 v_perm_b32 per 16 edges in the S kernel) — issue rate at 1 / 2 / 4 waves per SIMD and bit-exactness.
//   v_perm_b32 (what ships) | v_or_b32_sdwa (src1_sel:WORD_1 on a pre-masked src0) | v_pack_b32_f16 op_sel:[1,1,0] |
//   v_and_or_b32 | v_alignbit_b32 | v_fma_f32 (reference rate)
//   hipcc --offload-arch=gfx950 -O3 tools/pack_rate_probe.hip -o /tmp/pack_probe && /tmp/pack_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE>
__global__ void k(unsigned* out, int iters, unsigned seed) {
  unsigned x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = (threadIdx.x * 2654435761u) ^ (seed + i * 40503u);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      unsigned a = x[i], b = x[(i + 5) & 15], d;
      if (MODE == 0) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(a), "v"(0x07060302u));
      if (MODE == 1) asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(d) : "v"(b), "v"(a));
      if (MODE == 2) asm volatile("v_pack_b32_f16 %0, %1, %2 op_sel:[1,1,0]" : "=v"(d) : "v"(a), "v"(b));
      if (MODE == 3) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(0xffff0000u), "v"(a));
      if (MODE == 4) asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(d) : "v"(b), "v"(a));
      if (MODE == 5) asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(d) : "v"(a), "v"(b));
      if (MODE == 6) asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(d) : "v"(b), "v"(a));
      if (MODE == 7) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "=v"(d) : "v"(a), "0"(b));
      x[i] = d;
    }
  }
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// bit-exactness of the candidates against (b & 0xffff0000) | (a >> 16) over every pair of high halves drawn from a table that
// covers bf16 zeros, denormals, normals, infinities and NaNs (also as f16 patterns: the pack instruction reads them as f16)
__global__ void exact(unsigned* bad) {
  const unsigned hi_a = blockIdx.x, hi_b0 = threadIdx.x * 256;       // 65536 blocks x 256 threads x 256 = all 2^32 pairs
  unsigned n1 = 0, n2 = 0, n7 = 0;
  for (unsigned j = 0; j < 256; ++j) {
    const unsigned a = (hi_a << 16) | 0x1234u, b = ((hi_b0 + j) << 16) | 0xabcdu;
    const unsigned ref = (b & 0xffff0000u) | (a >> 16);
    unsigned d1, d2, d7;
    asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(d1) : "v"(b & 0xffff0000u), "v"(a));
    asm volatile("v_pack_b32_f16 %0, %1, %2 op_sel:[1,1,0]" : "=v"(d2) : "v"(a), "v"(b));
    asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "=v"(d7) : "v"(a), "0"(b & 0xffff0000u));
    n1 += d1 != ref; n2 += d2 != ref; n7 += d7 != ref;
  }
  if (n1) atomicAdd(bad + 0, n1);
  if (n2) atomicAdd(bad + 1, n2);
  if (n7) atomicAdd(bad + 2, n7);
}

int main() {
  int cus = 256, clk_khz = 0;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  unsigned* out;
  (void)hipMalloc(&out, (size_t)cus * 2048 * sizeof(unsigned));
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  static const char* names[] = {"v_perm_b32", "v_or_b32_sdwa src1_sel:WORD_1", "v_pack_b32_f16 op_sel:[1,1,0]", "v_and_or_b32",
                                "v_alignbit_b32", "v_fma_f32", "v_lshl_or_b32", "v_mov_b32_sdwa dst_sel:WORD_0 preserve"};
  for (int mode = 0; mode < 8; ++mode)
    for (int w : {1, 2, 4}) {
      const int threads = 64 * 4 * w;
      auto launch = [&]() {
#define L(M) if (mode == M) hipLaunchKernelGGL(k<M>, dim3(cus), dim3(threads), 0, 0, out, iters, 7u)
        L(0); L(1); L(2); L(3); L(4); L(5); L(6); L(7);
#undef L
      };
      launch();
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      launch();
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("%-42s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD (nominal %.2f GHz)\n", names[mode], w,
             ms * 1e-3 * clk_khz * 1e3 / ((double)iters * 16 * w), clk_khz / 1e6);
    }
  unsigned* bad;
  (void)hipMalloc(&bad, 3 * sizeof(unsigned));
  (void)hipMemset(bad, 0, 3 * sizeof(unsigned));
  hipLaunchKernelGGL(exact, dim3(65536), dim3(256), 0, 0, bad);
  unsigned h[3];
  (void)hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
  printf("mismatches over all 2^32 pairs of high halves: v_or_b32_sdwa %u, v_pack_b32_f16 %u, v_mov_b32_sdwa %u\n", h[0], h[1], h[2]);
  return 0;
}
