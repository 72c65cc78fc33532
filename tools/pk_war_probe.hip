// Does a VALU write that FOLLOWS a packed-f32 instruction overtake that instruction's operand reads on gfx950?
// (write-after-read on a v_pk_*_f32 source, with one or two waves per SIMD and with / without matrix instructions
// issued around it.)  Background: DESIGN.md §4 "what went wrong" 2 and tools/slp_probe.py — the SLP build of the S
// kernel (-O3 packs adjacent f32 math into v_pk_*_f32 and reuses the pair registers right behind them) miscomputes only
// with two waves per SIMD.  Every lane computes r = a + b with v_pk_add_f32 from registers that the NEXT instruction(s)
// overwrite, and compares with the same sums taken by plain v_add_f32 from untouched copies.
//   hipcc --offload-arch=gfx950 -O2 tools/pk_war_probe.hip -o /tmp/pk_war_probe && /tmp/pk_war_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// GAP: independent VALU instructions between the packed op and the overwrite of its sources (0 = adjacent)
// MFMA: issue matrix instructions in the loop (the sibling wave of the SIMD then has them in flight around our packed op)
template <int GAP, bool MFMA>
__global__ __launch_bounds__(512) void probe(unsigned long long* bad, int iters, float seed) {
  const int lane = threadIdx.x & 63;
  float x = seed + lane, y = 2.f * seed - lane, z = 0.5f * seed + 3.f * lane, w = seed - 7.f * lane;
  unsigned long long nbad = 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  bf16x8 fa, fb;
  for (int q = 0; q < 8; ++q) { fa[q] = (__bf16)(float)(lane + q); fb[q] = (__bf16)(float)(lane - q); }
  for (int it = 0; it < iters; ++it) {
    float r0, r1, e0, e1, junk0 = 1.0f + it, junk1 = -3.0f - it;
    if (MFMA) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
    }
    if (GAP == 0)
      asm volatile(
          "v_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v102, %6\n\tv_mov_b32 v103, %7\n\t"
          "v_add_f32 %2, %4, %6\n\tv_add_f32 %3, %5, %7\n\t"
          "s_nop 4\n\t"
          "v_pk_add_f32 v[104:105], v[100:101], v[102:103]\n\t"
          "v_mov_b32 v102, %8\n\t"           // src1.lo overwritten by the very next instruction
          "v_mov_b32 v101, %9\n\t"           // src0.hi by the one after
          "v_mov_b32 v100, %8\n\t"
          "v_mov_b32 v103, %9\n\t"
          "s_nop 7\n\t"
          "v_mov_b32 %0, v104\n\tv_mov_b32 %1, v105"
          : "=&v"(r0), "=&v"(r1), "=&v"(e0), "=&v"(e1)
          : "v"(x), "v"(y), "v"(z), "v"(w), "v"(junk0), "v"(junk1)
          : "v100", "v101", "v102", "v103", "v104", "v105");
    else if (GAP == 1)
      asm volatile(
          "v_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v102, %6\n\tv_mov_b32 v103, %7\n\t"
          "v_add_f32 %2, %4, %6\n\tv_add_f32 %3, %5, %7\n\t"
          "s_nop 4\n\t"
          "v_pk_add_f32 v[104:105], v[100:101], v[102:103]\n\t"
          "v_mov_b32 v106, %8\n\t"           // one independent instruction in between
          "v_mov_b32 v102, %8\n\t"
          "v_mov_b32 v101, %9\n\t"
          "v_mov_b32 v100, %8\n\t"
          "v_mov_b32 v103, %9\n\t"
          "s_nop 7\n\t"
          "v_mov_b32 %0, v104\n\tv_mov_b32 %1, v105"
          : "=&v"(r0), "=&v"(r1), "=&v"(e0), "=&v"(e1)
          : "v"(x), "v"(y), "v"(z), "v"(w), "v"(junk0), "v"(junk1)
          : "v100", "v101", "v102", "v103", "v104", "v105", "v106");
    else   // GAP == 2: the overwrites are themselves packed / the packed op carries op_sel (the shapes the SLP build emits)
      asm volatile(
          "v_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v102, %6\n\tv_mov_b32 v103, %7\n\t"
          "v_mul_f32 %2, %4, %6\n\tv_mul_f32 %3, %5, %6\n\t"
          "s_nop 4\n\t"
          "v_pk_mul_f32 v[104:105], v[100:101], v[102:103] op_sel_hi:[1,0]\n\t"     // (x z, y z)
          "v_mov_b32 v100, %8\n\t"
          "v_mov_b32 v102, %9\n\t"
          "v_mov_b32 v101, %8\n\t"
          "s_nop 7\n\t"
          "v_mov_b32 %0, v104\n\tv_mov_b32 %1, v105"
          : "=&v"(r0), "=&v"(r1), "=&v"(e0), "=&v"(e1)
          : "v"(x), "v"(y), "v"(z), "v"(w), "v"(junk0), "v"(junk1)
          : "v100", "v101", "v102", "v103", "v104", "v105");
    nbad += (r0 != e0) + (r1 != e1);
    x += 1.25f; y -= 0.75f; z += 0.5f; w += 2.0f;
  }
  if (acc[0] == 12345.678f) nbad += 1u << 30;          // keep the matrix instructions alive
  if (nbad) atomicAdd(bad, nbad);
}

template <int GAP, bool MFMA>
static void run(const char* what, int threads, unsigned long long* dbad) {
  hipMemset(dbad, 0, 8);
  const int iters = 200000;
  hipLaunchKernelGGL((probe<GAP, MFMA>), dim3(256), dim3(threads), 0, 0, dbad, iters, 1.5f);
  hipDeviceSynchronize();
  unsigned long long h = 0;
  hipMemcpy(&h, dbad, 8, hipMemcpyDeviceToHost);
  printf("%-70s waves/SIMD %d: %llu wrong results of %.3g\n", what, threads / 256, h, 2.0 * iters * 256.0 * threads);
  fflush(stdout);
}

int main() {
  unsigned long long* dbad;
  hipMalloc(&dbad, 8);
  for (int threads : {256, 512}) {
    if (threads == 256) {
      run<0, false>("pk_add, sources overwritten by the next instructions, no MFMA", 256, dbad);
      run<0, true>("pk_add, sources overwritten by the next instructions, MFMA in loop", 256, dbad);
      run<1, true>("pk_add, one instruction in between, MFMA in loop", 256, dbad);
      run<2, true>("pk_mul op_sel_hi:[1,0], sources overwritten next, MFMA in loop", 256, dbad);
    } else {
      run<0, false>("pk_add, sources overwritten by the next instructions, no MFMA", 512, dbad);
      run<0, true>("pk_add, sources overwritten by the next instructions, MFMA in loop", 512, dbad);
      run<1, true>("pk_add, one instruction in between, MFMA in loop", 512, dbad);
      run<2, true>("pk_mul op_sel_hi:[1,0], sources overwritten next, MFMA in loop", 512, dbad);
    }
  }
  return 0;
}
