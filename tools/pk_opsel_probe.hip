// Reduced reproducer for the S kernel's SLP-build miscompute (DESIGN.md §4 "what went wrong" 2, tools/slp_probe*.py).
//
// In the SLP build of csrc/decoder16.hip the epilogue of the second product multiplies pairs of accumulator values
// by g_e with v_pk_mul_f32; element (lane group 3, i = 1) of two of the four column blocks then came out as v * (+0)
// in ~2 % of the half tiles — only in lanes 48-63, only with two waves per SIMD.  This program runs ONE such
// instruction (several forms) in a loop on the first wave of every SIMD, checks each result against plain v_mul_f32 of
// untouched copies, and lets the second wave of the SIMD do something else meanwhile.
//   hipcc --offload-arch=gfx950 -O2 tools/pk_opsel_probe.hip -o /tmp/pk_opsel_probe && /tmp/pk_opsel_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Report { unsigned long long wrong, wrong_hi_lanes, wrong_lo_half, wrong_hi_half, zero_results; int lane[8]; float got[8], want[8]; };

// FORM  0: v_pk_mul_f32 D, A, B op_sel:[0,1]      (lo = A.lo * B.hi, hi = A.hi * B.hi)    in the kernel's neighbourhood
//       1: v_pk_mul_f32 D, A, B op_sel_hi:[1,0]   (lo = A.lo * B.lo, hi = A.hi * B.lo)    same neighbourhood
//       2: v_pk_mul_f32 D, A, B                   (lo = A.lo * B.lo, hi = A.hi * B.hi)    same neighbourhood
//       3: two v_mul_f32 (no packed instruction)                                            same neighbourhood
//       4: v_pk_mul_f32 D, A, B op_sel:[0,1] alone (register-built operands, idle cycles around it)
// SIB   what the SECOND wave of every SIMD does: 0 nothing (exits), 1 back-to-back v_mfma_f32_32x32x16_bf16,
//       2 back-to-back v_mfma_f32_16x16x32_bf16, 3 v_fma_f32 chain (no matrix instructions)
template <int FORM, int SIB>
__global__ __launch_bounds__(512) void probe(Report* rep, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[8][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8 fa, fb;
  for (int q = 0; q < 8; ++q) { fa[q] = (__bf16)(float)(lane + q); fb[q] = (__bf16)(float)(lane - q); }
  if (wave >= 4) {
    if (SIB == 0) return;
    f32x16 big;
    for (int i = 0; i < 16; ++i) big[i] = 0.f;
    f32x4 small = {0.f, 0.f, 0.f, 0.f};
    float u = seed + lane;
    for (int it = 0; it < iters * 4; ++it) {
      if (SIB == 1) big = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, big, 0, 0, 0);
      if (SIB == 2) { small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, small, 0, 0, 0); small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, small, 0, 0, 0); }
      if (SIB == 3) { u = fmaf(u, 1.0001f, 0.5f); u = fmaf(u, 0.9999f, -0.5f); u = fmaf(u, 1.0001f, 0.5f); u = fmaf(u, 0.9999f, -0.5f); }
    }
    if (u + small[0] + big[0] == 12345.678f) atomicAdd(&rep->wrong, 1ull << 40);
    return;
  }
  float a0 = seed + lane, a1 = 2.f * seed - lane, b0 = 0.5f * seed + 3.f * lane;
  unsigned long long nbad = 0, nhi = 0, nlo_half = 0, nhi_half = 0, nzero = 0;
  const uint32_t lds_addr = (uint32_t)(uintptr_t)&lds[wave][(lane >> 4) * 4];     // the 16-byte group of the lane's 16-lane row
  for (int it = 0; it < iters; ++it) {
    float r0, r1, e0, e1;
    lds[wave][lane] = b0 + (lane & 3);
    // operands: A = (a0, a1) in v[100:101]; B = dwords 0, 1 of a ds_read_b128 (as g_e in the kernel) in v[108:109]
#define NEIGHBOURHOOD(EXPECT, PACKED)                                                                                   \
    asm volatile(                                                                                                       \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                      \
        "ds_read_b128 v[108:111], %6\n\t"                                                                               \
        "ds_read2_b32 v[104:105], %6 offset0:1 offset1:2\n\t"                                                           \
        "v_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\t"                                                                  \
        "s_waitcnt lgkmcnt(0)\n\t" EXPECT                                                                               \
        "v_bfe_i32 v112, v104, %7, 1\n\tv_bfe_i32 v113, v105, %7, 1\n\tv_bfe_i32 v114, v104, %7, 1\n\tv_bfe_i32 v115, v105, %7, 1\n\t" \
        "v_and_b32 v116, v112, v100\n\t" PACKED                                                                         \
        "v_mov_b32 v117, v110\n\tv_mov_b32 v118, v111\n\tv_mov_b32 v119, v112\n\tv_mov_b32 v120, v113\n\t"              \
        "v_mov_b32 v121, v114\n\tv_mov_b32 v122, v115\n\tv_mov_b32 v123, v116\n\tv_mov_b32 v112, v117\n\t"              \
        "v_pk_mul_f32 v[114:115], v[118:119], v[110:111] op_sel_hi:[1,0]\n\t"                                           \
        "v_pk_mul_f32 v[116:117], v[120:121], v[108:109] op_sel_hi:[1,0]\n\t"                                           \
        "s_nop 7\n\t"                                                                                                   \
        "v_mov_b32 %0, v104\n\tv_mov_b32 %1, v105"                                                                      \
        : "=&v"(r0), "=&v"(r1), "=&v"(e0), "=&v"(e1)                                                                    \
        : "v"(a0), "v"(a1), "v"(lds_addr), "v"(lane & 31)                                                               \
        : "v100", "v101", "v104", "v105", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", \
          "v118", "v119", "v120", "v121", "v122", "v123", "memory")
    // A = v[100:101], B = v[108:109], C = v[110:111] (dwords 2, 3 of the ds_read_b128); results in v[104:105]
    if (FORM == 0)
      NEIGHBOURHOOD("v_mul_f32 %2, %4, v109\n\tv_mul_f32 %3, %5, v109\n\t", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel:[0,1]\n\t");
    else if (FORM == 1)
      NEIGHBOURHOOD("v_mul_f32 %2, %4, v108\n\tv_mul_f32 %3, %5, v108\n\t", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel_hi:[1,0]\n\t");
    else if (FORM == 2)
      NEIGHBOURHOOD("v_mul_f32 %2, %4, v108\n\tv_mul_f32 %3, %5, v109\n\t", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109]\n\t");
    else if (FORM == 3)
      NEIGHBOURHOOD("v_mul_f32 %2, %4, v109\n\tv_mul_f32 %3, %5, v109\n\t", "v_mul_f32 v104, v100, v109\n\tv_mul_f32 v105, v101, v109\n\t");
    else if (FORM == 5)    // lo = A.hi * B.lo, hi = A.hi * B.hi
      NEIGHBOURHOOD("v_mul_f32 %2, %5, v108\n\tv_mul_f32 %3, %5, v109\n\t", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel:[1,0]\n\t");
    else if (FORM == 6)    // lo = A.hi * B.hi, hi = A.hi * B.hi
      NEIGHBOURHOOD("v_mul_f32 %2, %5, v109\n\tv_mul_f32 %3, %5, v109\n\t", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel:[1,1]\n\t");
    else if (FORM == 7)    // lo = A.lo + B.hi, hi = A.hi + B.hi
      NEIGHBOURHOOD("v_add_f32 %2, %4, v109\n\tv_add_f32 %3, %5, v109\n\t", "v_pk_add_f32 v[104:105], v[100:101], v[108:109] op_sel:[0,1]\n\t");
    else if (FORM == 8)    // fma: lo = A.lo * B.hi + C.lo, hi = A.hi * B.hi + C.hi
      NEIGHBOURHOOD("v_fma_f32 %2, %4, v109, v110\n\tv_fma_f32 %3, %5, v109, v111\n\t", "v_pk_fma_f32 v[104:105], v[100:101], v[108:109], v[110:111] op_sel:[0,1,0]\n\t");
    else if (FORM == 9)    // fma: lo = A.hi * B.lo + C.lo, hi = A.hi * B.hi + C.hi   (the form of the SLP build's skip-feature gradient)
      NEIGHBOURHOOD("v_fma_f32 %2, %5, v108, v110\n\tv_fma_f32 %3, %5, v109, v111\n\t", "v_pk_fma_f32 v[104:105], v[100:101], v[108:109], v[110:111] op_sel:[1,0,0]\n\t");
    else if (FORM == 10)   // fma: lo = A.lo * B.lo + C.hi, hi = A.hi * B.hi + C.hi
      NEIGHBOURHOOD("v_fma_f32 %2, %4, v108, v111\n\tv_fma_f32 %3, %5, v109, v111\n\t", "v_pk_fma_f32 v[104:105], v[100:101], v[108:109], v[110:111] op_sel:[0,0,1]\n\t");
    else if (FORM == 11)   // hi = A.hi * B.lo ... with op_sel_hi on src0: lo = A.lo * B.lo, hi = A.lo * B.hi
      NEIGHBOURHOOD("v_mul_f32 %2, %4, v108\n\tv_mul_f32 %3, %4, v109\n\t", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel_hi:[0,1]\n\t");
    else
      asm volatile(
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v108, %6\n\tv_mov_b32 v109, %7\n\t"
          "v_mul_f32 %2, %4, %7\n\tv_mul_f32 %3, %5, %7\n\t"
          "s_nop 4\n\t"
          "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel:[0,1]\n\t"
          "s_nop 7\n\t"
          "v_mov_b32 %0, v104\n\tv_mov_b32 %1, v105"
          : "=&v"(r0), "=&v"(r1), "=&v"(e0), "=&v"(e1)
          : "v"(a0), "v"(a1), "v"(b0), "v"(b0 + 1.f)
          : "v100", "v101", "v104", "v105", "v108", "v109", "memory");
#undef NEIGHBOURHOOD
    const int w0 = r0 != e0, w1 = r1 != e1;
    if (w0 | w1) {
      const unsigned long long slot = atomicAdd(&rep->wrong, (unsigned long long)(w0 + w1));
      if (slot < 8) { rep->lane[slot] = lane; rep->got[slot] = w0 ? r0 : r1; rep->want[slot] = w0 ? e0 : e1; }
      nhi += (lane >= 48) * (w0 + w1); nlo_half += w0; nhi_half += w1;
      nzero += (w0 && r0 == 0.f) + (w1 && r1 == 0.f);
    }
    a0 += 1.25f; a1 -= 0.75f; b0 += 0.5f;
  }
  if (nhi | nlo_half | nhi_half) {
    atomicAdd(&rep->wrong_hi_lanes, nhi); atomicAdd(&rep->wrong_lo_half, nlo_half); atomicAdd(&rep->wrong_hi_half, nhi_half);
    atomicAdd(&rep->zero_results, nzero);
  }
  (void)nbad;
}

template <int FORM, int SIB>
static void run(const char* what, Report* d) {
  (void)hipMemset(d, 0, sizeof(Report));
  const int iters = 400000;
  hipLaunchKernelGGL((probe<FORM, SIB>), dim3(256), dim3(512), 0, 0, d, iters, 1.5f);
  (void)hipDeviceSynchronize();
  Report h;
  (void)hipMemcpy(&h, d, sizeof(Report), hipMemcpyDeviceToHost);
  printf("%-78s %8llu wrong of %.3g results: %llu in lanes 48-63, %llu low halves, %llu high halves, %llu are +-0", what, h.wrong,
         2.0 * iters * 256.0 * 256.0, h.wrong_hi_lanes, h.wrong_lo_half, h.wrong_hi_half, h.zero_results);
  for (int i = 0; i < 3 && i < (int)h.wrong; ++i) printf("  [lane %d got %g want %g]", h.lane[i], h.got[i], h.want[i]);
  printf("\n");
  fflush(stdout);
}

int main() {
  Report* d;
  (void)hipMalloc(&d, sizeof(Report));
  printf("first wave of every SIMD: the checked instruction; second wave of the SIMD: see each line\n");
  run<0, 0>("pk_mul op_sel:[0,1]     | sibling: none (one wave per SIMD)", d);
  run<0, 3>("pk_mul op_sel:[0,1]     | sibling: v_fma_f32 chain", d);
  run<0, 2>("pk_mul op_sel:[0,1]     | sibling: v_mfma_f32_16x16x32_bf16 back to back", d);
  run<0, 1>("pk_mul op_sel:[0,1]     | sibling: v_mfma_f32_32x32x16_bf16 back to back", d);
  run<1, 1>("pk_mul op_sel_hi:[1,0]  | sibling: v_mfma_f32_32x32x16_bf16 back to back", d);
  run<2, 1>("pk_mul (no op_sel)      | sibling: v_mfma_f32_32x32x16_bf16 back to back", d);
  run<3, 1>("2 x v_mul_f32 (control) | sibling: v_mfma_f32_32x32x16_bf16 back to back", d);
  run<4, 1>("pk_mul op_sel:[0,1] alone, idle cycles around | sibling: v_mfma_f32_32x32x16_bf16", d);
  run<4, 2>("pk_mul op_sel:[0,1] alone, idle cycles around | sibling: v_mfma_f32_16x16x32_bf16", d);
  printf("other operand-half selections, sibling: v_mfma_f32_16x16x32_bf16 back to back\n");
  run<1, 2>("pk_mul op_sel_hi:[1,0]   (hi: A.hi * B.lo)", d);
  run<2, 2>("pk_mul, no op_sel", d);
  run<3, 2>("2 x v_mul_f32 (control)", d);
  run<5, 2>("pk_mul op_sel:[1,0]      (lo: A.hi * B.lo)", d);
  run<6, 2>("pk_mul op_sel:[1,1]      (lo: A.hi * B.hi)", d);
  run<11, 2>("pk_mul op_sel_hi:[0,1]   (hi: A.lo * B.hi)", d);
  run<7, 2>("pk_add op_sel:[0,1]      (lo: A.lo + B.hi)", d);
  run<8, 2>("pk_fma op_sel:[0,1,0]    (lo: A.lo * B.hi + C.lo)", d);
  run<9, 2>("pk_fma op_sel:[1,0,0]    (lo: A.hi * B.lo + C.lo)", d);
  run<10, 2>("pk_fma op_sel:[0,0,1]    (lo: A.lo * B.lo + C.hi)", d);
  return 0;
}
