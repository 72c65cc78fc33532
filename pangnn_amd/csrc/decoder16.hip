// Training form of the fused link decoder for D = 64 (src/gnn.py:110-116,171-177; pangnn.py:200-207), gfx950,
// bf16 matrix pipe with fp32-exact operand splitting, TWO waves per SIMD.
//
//   h1[e]   = relu(P[src_e] + Q[dst_e] (+ w_e c))     P = z W1a^T, Q = z W1b^T + b1   (node level)
//   h2[e]   = relu(W2 h1[e] + b2)
//   logit_e = w3 . h2[e] + b3 ;  loss = mean BCEWithLogits(pos_weight)
//
// Two kernels replace the one-wave-per-SIMD kernel of round 1 (decoder_bwd_x3_kernel) and the 19 GB
// dL/dh1 [E,64] round trip behind it:
//
//  S  decoder_train16_kernel   edges in the caller's order, 16-edge half tiles on v_mfma_f32_16x16x32_bf16:
//       P1  C[j][e] = sum_k W2[j][k] h1[e][k]            (three-way split operands, 6 partial products)
//       loss, g_e = dL/dlogit_e, m2[j][e] = [h2 > 0], m1[e][k] = [h1 > 0]
//       P2  dL/dh1[e][k] = m1 g_e sum_j m2[j][e] W2'[j][k],   W2' = diag(w3) W2   (A = m2 is EXACT in bf16,
//           B = W2' split three ways: 3 products, fp32-exact)          -> per-(tile, source) run sums
//       P3  dL/dW2[j][k] = w3[j] sum_e m2[j][e] (g_e h1[e][k])        (A = m2 exact, B = g_e h1 split three
//           ways: v_mfma_f32_32x32x16_bf16 with K = the 16 edges)
//     and a 32-byte record per edge {m1 | m2 bit masks (16 B), g_e} for the target side.
//  T  decoder_dgrad16_kernel   edges in by-target order (through the CSR permutation): rebuilds m2 / m1 / g_e
//       from the records (no node-row gathers, no P1), runs P2 and sums dL/dh1 over target runs.
//
// Per wave and 16 edges: 48 + 24 MFMA(16x16x32) + 12 MFMA(32x32x16) in S, 24 in T; every product carries
// fp32-level error (no two-term shortcuts).  No divergent control flow around matrix operands: the last tile is
// padded by clamping edge ids to E-1 (dead lanes compute a duplicate whose dL/dlogit is forced to 0), so there is
// ONE tile body, every lane is active in every MFMA / transposing LDS read, and only global stores are predicated.
#include "common.h"

// This file is compiled TWICE (csrc/Makefile): as decoder16.o — everything, the 2-byte table format of the PQ16 instances being
// bfloat16 — and, with -DPANGNN_D16_TABLES_F16, as decoder16_f16.o: the S and inference kernels again with PQ16 = IEEE half
// (`--mixed_precision fp16`), under their own names, plus their two launchers; nothing else of the file.  The float32 /
// bfloat16 kernels of decoder16.o are thereby exactly what they were before the float16 format existed (same source, same
// template signatures, same code object).
#ifdef PANGNN_D16_TABLES_F16
#define decoder_train16_kernel decoder_train16_f16_kernel
#define decoder_infer16_kernel decoder_infer16_f16_kernel
#define D16_TABLES16(name) name##_f16
#else
#define D16_TABLES16(name) name##_bf16
#endif

namespace pangnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int D16 = 64;
constexpr int SLAB16 = 64 * 64 + 64 + 64 + 64 + 16;   // gW2 | gb2 | gw3 | gcvec | gb3, loss (+pad): layout of decoder.hip
#ifdef PANGNN_D16_PROBE_ONE_WAVE        // tools/slp_probe.sh: diagnostic builds only (one wave per SIMD)
constexpr int S_WAVES = 4;
#else
constexpr int S_WAVES = 8;                            // 512 threads, one workgroup per CU, two waves per SIMD
#endif
// Run sums are taken per CHUNK of 2^chunk_log consecutive 32-edge tiles: a wave walks the tiles of a chunk in order and
// carries an open run from tile to tile, so a run only ends where its key changes or the chunk ends — one part row per
// (chunk, key) run (N + E / 512 rows instead of one per (tile, key) run, N + E / 32), and most half tiles have no
// boundary to handle at all.  Chunk boundaries depend on the edge list alone, never on the grid: results stay
// independent of the number of workgroups.
// The chunk size is a function of the edge count alone (chunk_log_for below): 16 tiles where that still gives every wave of
// the chip several chunks (E >= 1e6), fewer for short lists — a mini-batch of 2 000 edges as ONE chunk per tile runs on 63
// waves instead of 4 (S 138 -> 19 us, T 63 -> 15 us at E = 2 000: its waves walked 16 tiles each, one after the other).
constexpr int D16_CHUNK_LOG_MAX = 4;
constexpr int64_t D16_MIN_CHUNKS = 2048;               // waves of one full launch (256 CUs x 8)
__attribute__((unused)) static inline int chunk_log_for(int64_t n_tiles) {
  int c = D16_CHUNK_LOG_MAX;
  while (c > 0 && (n_tiles >> c) < D16_MIN_CHUNKS) --c;
  return c;
}
// the wave's next tile: the next one of its chunk, or the first of the chunk `cstride` chunks on
__device__ __forceinline__ int64_t next_tile(int64_t tile, int64_t cstride, int64_t n_tiles, bool& last, int clog) {
  const int64_t t1 = tile + 1;
  last = (t1 & ((1 << clog) - 1)) == 0 || t1 >= n_tiles;
  return last ? ((tile >> clog) + cstride) << clog : t1;
}
constexpr int T_WAVES = 16;                           // dgrad kernel: four waves per SIMD (128 registers each): its record gathers are latency bound

// ---- LDS images.  Rows of 64 bf16 = 128 B = 8 chunks of 16 B, unpadded; chunk ch of row r sits at
// physical chunk ch ^ f(r).  The swizzles make every access pattern below conflict-free (tools/lds_banks.py):
//  weight images (rows j or k, read with ds_read_b128 as 16x16x32 A / B fragments): f(r) = r & 6
//  per-wave tile images (rows e, read with ds_read_b64_tr_b16):                     f(r) = ((r&2)<<1) | ((r&4)>>1)
__device__ __forceinline__ constexpr int wsw(int r) { return r & 6; }
__device__ __forceinline__ constexpr int tsw(int r) { return ((r & 2) << 1) | ((r & 4) >> 1); }

constexpr int W_IMG = 64 * 128;                       // bytes of one weight image
constexpr int T_IMG = 16 * 128;                       // bytes of one tile image
constexpr int LDS_W2 = 0;                             // W2 hi | mid | lo          (P1 A operand)
constexpr int LDS_W2P = 3 * W_IMG;                    // W2'^T hi | mid | lo       (P2 B operand)
constexpr int LDS_VEC = 6 * W_IMG;                    // b2 | w3 | cvec   (3 x 64 floats)
constexpr int LDS_WAVE0 = LDS_VEC + 3 * 64 * 4;
// per wave: Hg hi | mid | lo | m2 image | recl[16][4] | gl[16] | wl[16]
constexpr int WV_HG = 0, WV_M2 = 3 * T_IMG, WV_REC = 4 * T_IMG, WV_GL = WV_REC + 256, WV_WL = WV_GL + 64;
constexpr int WV_BYTES = WV_WL + 64;                  // 8576
constexpr int S_LDS = LDS_WAVE0 + S_WAVES * WV_BYTES;

// s_setprio around every run of matrix instructions: with two to four waves per SIMD the arbiter otherwise lets a
// wave in its vector phase starve the one feeding the matrix pipe; raised priority for the MFMA issuer keeps the
// pipe busy while the other waves fill the issue slots in between (measured: -4 % on the S kernel).
#ifdef PANGNN_D16_PROBE_NOPRIO          // tools/slp_probe.sh: diagnostic builds only
#define D16_SETPRIO(x) ((void)0)
#else
#define D16_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#endif

struct Split3 { bf16x8 hi, mid, lo; };
// x = hi + mid + lo EXACTLY: three bf16 terms by truncation (8 + 8 + 8 significand bits, all of the sign of x).
// Per element: and, sub, and, sub; the three terms of two neighbours are packed by one v_perm_b32 each.
__device__ __forceinline__ Split3 split8(const float (&f)[8]) {
  u32x4 hi, mid, lo;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t a = __builtin_bit_cast(uint32_t, f[2 * q]), b = __builtin_bit_cast(uint32_t, f[2 * q + 1]);
    hi[q] = __builtin_amdgcn_perm(b, a, 0x07060302u);                       // the two high halves
    const float ra = f[2 * q] - __builtin_bit_cast(float, a & 0xffff0000u);
    const float rb = f[2 * q + 1] - __builtin_bit_cast(float, b & 0xffff0000u);
    const uint32_t ua = __builtin_bit_cast(uint32_t, ra), ub = __builtin_bit_cast(uint32_t, rb);
    mid[q] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
    const float sa = ra - __builtin_bit_cast(float, ua & 0xffff0000u);
    const float sb = rb - __builtin_bit_cast(float, ub & 0xffff0000u);
    lo[q] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, sb), __builtin_bit_cast(uint32_t, sa), 0x07060302u);
  }
  Split3 r;
  r.hi = __builtin_bit_cast(bf16x8, hi);
  r.mid = __builtin_bit_cast(bf16x8, mid);
  r.lo = __builtin_bit_cast(bf16x8, lo);
  return r;
}

// max(x, 0) as a compiler-visible instruction.  NOT inline asm: an asm statement that DEFINES a VGPR is invisible
// to LLVM's MFMA hazard recognizer, so it can be scheduled right behind an MFMA that still reads that register as a
// source operand (write-after-read on an in-flight matrix instruction) — the later passes of the MFMA then see the
// new value.  That is what made the round-1 kernel's "AGPR form" build intermittently wrong and what zeroed rows
// 12-15 of this kernel's second product in its first version (tools/find_asm_mfma_war.py).
// Written as an integer max of the bit pattern (negative floats are negative integers): one v_max_i32, without the
// canonicalising v_max_f32(x, x) hipcc puts in front of a float max.
__device__ __forceinline__ float relu1(float x) {
  const int i = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, i > 0 ? i : 0);
}

// x + x[lane ^ 16] and x + x[lane ^ 32] as one swap + one add each (gfx950 v_permlane{16,32}_swap; the
// clang builtin folds the two results of a swap of a value with itself, so the instruction is written out;
// the s_nop covers the VALU-write -> permlane-read wait states hipcc cannot see inside an asm statement).
// Both registers are read-write operands whose initial values the compiler writes with its own (hazard-checked)
// instructions, so — unlike a pure asm output — they cannot land on a register an in-flight MFMA still reads.
__device__ __forceinline__ float xsum16(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float xsum32(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
// Sums of FOUR per-lane values over the four 16-lane rows with three swaps and three adds (no copies): a swap of
// two DIFFERENT values hands each half (row pair) the other half's share of one of them.  Result: row 0 holds the
// total of s0, row 1 of s2, row 2 of s1, row 3 of s3  (row g holds s[kRow4[g]]).
__device__ __forceinline__ float red4(float s0, float s1, float s2, float s3) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(s0), "+v"(s1));   // s0: [s0.lo, s1.lo]  s1: [s0.hi, s1.hi]
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(s2), "+v"(s3));
  float u = s0 + s1, w = s2 + s3;                  // rows 0,1: s0 (s2) summed over the halves; rows 2,3: s1 (s3)
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(u), "+v"(w));     // u: [u.r0, w.r0, u.r2, w.r2]  w: [u.r1, w.r1, u.r3, w.r3]
  return u + w;
}

// [x.lo16 != 0] | [x.hi16 != 0] << 16 as ONE v_pk_min_u16 (`one2` = 0x00010001 in a register the optimiser cannot see
// through: against a literal it rewrites the minimum as two compares + selects)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t nz16x2(uint32_t x, uint32_t one2) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, x), __builtin_bit_cast(u16x2, one2)));
}

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef __attribute__((address_space(3))) short4v lds_s4;
__device__ __forceinline__ short4v ld_tr4(const char* lds_base, int off) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds_base + off));
}
__device__ __forceinline__ bf16x8 ld_tr8(const char* lds_base, int off_lo, int off_hi) {
  struct P { short4v lo, hi; } p;
  p.lo = ld_tr4(lds_base, off_lo);
  p.hi = ld_tr4(lds_base, off_hi);
  return __builtin_bit_cast(bf16x8, p);
}
__device__ __forceinline__ bf16x8 ld_b128(const char* lds_base, int off) {
  return *reinterpret_cast<const bf16x8*>(lds_base + off);
}

#ifdef PANGNN_D16_DEBUG
__device__ float* d16_dbg_v = nullptr;     // [E][64] dL/dh1pre as the S kernel sees it (diagnostic builds only)
#endif
#ifdef PANGNN_D16_STAMP                   // diagnostic builds only (tools/probe_decoder_stamps.py): where a small launch spends its time
__device__ unsigned long long* d16_stamps = nullptr;     // wave 0 of workgroup 0: wall_clock64() (100 MHz) at the marked points
#define D16_STAMP(i)                                                                          \
  do {                                                                                        \
    if (d16_stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {                       \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                             \
      d16_stamps[i] = wall_clock64();                                                         \
    }                                                                                         \
  } while (0)
#else
#define D16_STAMP(i) ((void)0)
#endif
#ifdef PANGNN_D16_CYC                     // diagnostic builds only (tools/probe_decoder_cycles.py): where the S kernel's waves spend their
// cycles AT FULL SIZE: every wave adds the shader-clock cycles (s_memtime) between consecutive marks of its loop body into
// per-phase sums over all of its half tiles — scalar registers, no waits of its own beyond the counter read's lgkmcnt — and
// the eight waves of workgroup 0 write theirs out at the end: d16_cyc[wave][0 .. 5] = cycles in phase 0 .. 5, [6] = half tiles,
// [7] = cycles from kernel entry to loop exit.
__device__ unsigned long long* d16_cyc = nullptr;
#define D16_CYC_DECL unsigned long long cyc_acc[6] = {0, 0, 0, 0, 0, 0}, cyc_halves = 0; \
  const unsigned long long cyc_t0 = __builtin_readcyclecounter(); unsigned long long cyc_last = cyc_t0
#define D16_CYC(i)                                                  \
  do {                                                              \
    const unsigned long long t__ = __builtin_readcyclecounter();    \
    cyc_acc[i] += t__ - cyc_last;                                   \
    cyc_last = t__;                                                 \
  } while (0)
#else
#define D16_CYC_DECL ((void)0)
#define D16_CYC(i) ((void)0)
#endif

struct D16Params {
  const void* p; const void* q; uint32_t ldp_b; uint32_t ldq_b;     // row strides in bytes; rows are f32 or bf16 (PQ16)
  const int64_t* ei; int64_t ld; int64_t E;
  const float* extra; const float* cvec;
  const float* w2; const float* b2; const float* w3; const float* b3;
  // e_live (nullable, device): the first *e_live edges of the list are real, the rest is padding (a fixed-shape batch whose
  // tail is inert, train.ReplayedFreshStep): padded positions keep their own logit / record slots but enter no sum
  // (dL/dlogit = 0) and the fused loss is the mean over *e_live edges.
  const int64_t* e_live;
};
struct D16Loss { const float* y; const float* pos_weight; float inv_denom; };
struct D16Run { float* part; const int32_t* part_off; int chunk_log; };   // chunk_log: tiles per chunk = 1 << chunk_log

__device__ __forceinline__ float4 ld_row16(const void* table, uint32_t byte_off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(table) + byte_off);
}
__device__ __forceinline__ uint4 ld_row16u(const void* table, uint32_t byte_off) {
  return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(table) + byte_off);
}

// ---- weights -> LDS images (once per workgroup).  which = 1: W2 (P1), 2: W2'^T (P2), 3: both.
__device__ __forceinline__ void stage_weights16(const float* w2, const float* b2, const float* w3, const float* cvec,
                                                char* lds, int nthreads, int which, int w2p_base, int vec_base) {
  // the loads of several rounds in flight (a mini-batch launch is a handful of workgroups that all start here)
#pragma unroll 8
  for (int i = threadIdx.x; i < 64 * 64; i += nthreads) {
    const int j = i >> 6, k = i & 63;
    const float w = w2[i];
    if (which & 1) {
      const __bf16 h = (__bf16)w;
      const float r1 = w - (float)h;
      const __bf16 m = (__bf16)r1;
      const __bf16 l = (__bf16)(r1 - (float)m);
      const int off = LDS_W2 + j * 128 + (((k >> 3) ^ wsw(j)) << 4) + 2 * (k & 7);
      *reinterpret_cast<__bf16*>(lds + off) = h;
      *reinterpret_cast<__bf16*>(lds + off + W_IMG) = m;
      *reinterpret_cast<__bf16*>(lds + off + 2 * W_IMG) = l;
    }
    if (which & 2) {
      // row k; the 64 j of a row in the order P1's accumulator hands them to P2: K-step t = j >> 5, lane group
      // g = (j >> 2) & 3, element s = 4 ((j >> 4) & 1) + (j & 3)   <->   chunk 4 t + g, element s
      const float wp = w3[j] * w;
      const __bf16 h = (__bf16)wp;
      const float r1 = wp - (float)h;
      const __bf16 m = (__bf16)r1;
      const __bf16 l = (__bf16)(r1 - (float)m);
      const int t = j >> 5, g = (j >> 2) & 3, s = 4 * ((j >> 4) & 1) + (j & 3);
      const int off = w2p_base + k * 128 + (((4 * t + g) ^ wsw(k)) << 4) + 2 * s;
      *reinterpret_cast<__bf16*>(lds + off) = h;
      *reinterpret_cast<__bf16*>(lds + off + W_IMG) = m;
      *reinterpret_cast<__bf16*>(lds + off + 2 * W_IMG) = l;
    }
  }
  float* vec = reinterpret_cast<float*>(lds + vec_base);
  for (int i = threadIdx.x; i < 64; i += nthreads) {
    vec[i] = b2 ? b2[i] : 0.f;
    vec[64 + i] = w3[i];
    vec[128 + i] = cvec ? cvec[i] : 0.f;
  }
}

// epilogue of P2: dL/dh1pre[e][k] = v[kb][i] * gm[kb][i] with gm = g_e where the relu mask m1 (the record bits of the 16
// edges in LDS) is set, +0 elsewhere.  The factor is built here (one bit-field extract + one AND per element); the
// product itself is left to the consumer — the run sums take it as a fused multiply-add (run_sums), so a half tile
// without a run boundary costs 16 FMAs instead of 16 multiplies + 12 adds.
__device__ __forceinline__ void dgrad_masks(const char* recl, const char* gl, int c, int g, f32x4 (&gm)[4]) {
  const int bitpos = 16 * (c & 1) + 15 - ((c & 7) >> 1);      // m1 bits: the high byte of each half of a record dword
  const char* rrow = recl + 64 * g + 4 * (c >> 3);      // edge 4 g + i at + 16 i; dwords (c >> 3) and 2 + (c >> 3)
#if defined(PANGNN_D16_PROBE_GE_SCALAR) || defined(PANGNN_D16_PROBE_WAIT0)
  // diagnostic builds (tools/slp_probe.sh): the same arithmetic with (a) g_e read as four dwords instead of one
  // ds_read_b128, (b) every LDS operand of the epilogue landed (lgkmcnt(0) + idle cycles) before the first use
  f32x4 ge4;
#ifdef PANGNN_D16_PROBE_GE_SCALAR
#pragma unroll
  for (int i = 0; i < 4; ++i) ge4[i] = *reinterpret_cast<const volatile float*>(gl + 16 * g + 4 * i);
#else
  ge4 = *reinterpret_cast<const f32x4*>(gl + 16 * g);
#endif
  uint32_t dd[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    dd[i][0] = *reinterpret_cast<const uint32_t*>(rrow + 16 * i);
    dd[i][1] = *reinterpret_cast<const uint32_t*>(rrow + 16 * i + 8);
  }
#ifdef PANGNN_D16_PROBE_WAIT0
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" : "+v"(ge4), "+v"(dd[0][0]), "+v"(dd[1][0]), "+v"(dd[2][0]), "+v"(dd[3][0]),
               "+v"(dd[0][1]), "+v"(dd[1][1]), "+v"(dd[2][1]), "+v"(dd[3][1]) :: "memory");
#endif
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int keep = __builtin_amdgcn_sbfe((int)dd[i][kb & 1], bitpos - 4 * (kb >> 1), 1);
      const float gei = ge4[i];          // a float OBJECT: __builtin_bit_cast of the vector element itself takes element 0
      gm[kb][i] = __builtin_bit_cast(float, __builtin_bit_cast(int, gei) & keep);
    }
#else
  const f32x4 ge4 = *reinterpret_cast<const f32x4*>(gl + 16 * g);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t d0 = *reinterpret_cast<const uint32_t*>(rrow + 16 * i);          // kb even
    const uint32_t d1 = *reinterpret_cast<const uint32_t*>(rrow + 16 * i + 8);      // kb odd
    // a float OBJECT first: __builtin_bit_cast applied to the vector-element expression ge4[i] itself reinterprets the
    // start of the vector (element 0 for every i — hipcc 7.2 emits one ds_read_b32 for all four)
    const float gei = ge4[i];
    const int gbits = __builtin_bit_cast(int, gei);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const uint32_t d = (kb & 1) ? d1 : d0;
      const int keep = __builtin_amdgcn_sbfe((int)d, bitpos - 4 * (kb >> 1), 1);     // 0 or -1
      gm[kb][i] = __builtin_bit_cast(float, gbits & keep);
    }
  }
#endif
}
// the masked products themselves (rare paths of the run sums, the skip-feature gradient, diagnostic dumps)
__device__ __forceinline__ void dgrad_apply(f32x4 (&v)[4], const f32x4 (&gm)[4]) {
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float val = v[kb][i] * gm[kb][i];
#ifdef PANGNN_D16_PROBE_OPAQUE_EPI
      asm volatile("" : "+v"(val));
#endif
      v[kb][i] = val;
    }
}

// ---- P2 + its epilogue + run sums, shared by S and T.  Lane (c = lane & 15, g = lane >> 4).
//   a2[t]   mask m2 as bf16 0/1 A fragments: element s of K-step t is j = 32 t + 16 (s >> 2) + 4 g + (s & 3), edge c
//   recl    [16][4] dwords in LDS: the m1 bits of edge e, lane group g' (bit 16 (s&1) + 7 - (4 ks + (s >> 1)) for
//           k = 32 ks + 8 g' + s)
//   gl      [16] floats in LDS: g_e
// returns v[kb][i] and gm[kb][i] with dL/dh1pre[e = 4 g + i][k = 16 kb + c] = v * gm (dgrad_masks)
template <bool PIPE>
__device__ __forceinline__ void dgrad_tile(const char* lds, const char* recl, const char* gl, const bf16x8 (&a2)[2],
                                           int c, int g, int w2p_off0, int w2p_off1, f32x4 (&v)[4], f32x4 (&gm)[4]) {
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) v[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the three W2' terms of step (t, kb) are read one step ahead of their MFMAs (an LDS round trip is ~100 cycles,
  // the three MFMAs of a step 48)
  bf16x8 bq[2][3];
  if (PIPE) {
#pragma unroll
    for (int x = 0; x < 3; ++x) bq[0][x] = ld_b128(lds, w2p_off0 + x * W_IMG);
  }
  D16_SETPRIO(1);
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int t = st >> 2, kb = st & 3;
    if (PIPE) {
      if (st < 7) {
        const int off = (((st + 1) >> 2) ? w2p_off1 : w2p_off0) + ((st + 1) & 3) * 2048;     // rows 16 kb + c
#pragma unroll
        for (int x = 0; x < 3; ++x) bq[(st + 1) & 1][x] = ld_b128(lds, off + x * W_IMG);
      }
    } else {
      const int off = (t ? w2p_off1 : w2p_off0) + kb * 2048;
#pragma unroll
      for (int x = 0; x < 3; ++x) bq[st & 1][x] = ld_b128(lds, off + x * W_IMG);
    }
    v[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[t], bq[st & 1][2], v[kb], 0, 0, 0);     // lo, mid, hi
    v[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[t], bq[st & 1][1], v[kb], 0, 0, 0);
    v[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[t], bq[st & 1][0], v[kb], 0, 0, 0);
  }
  D16_SETPRIO(0);
  dgrad_masks(recl, gl, c, g, gm);
}

// P2 of BOTH halves of a 32-edge tile per W2' fragment read (the T kernel): one ds_read_b128 triple feeds six MFMAs,
// i.e. half the LDS bytes per edge of dgrad_tile — the T kernel at one fragment triple per three MFMAs ran the LDS
// at ~70 % of its bandwidth — and two independent accumulator chains in the matrix pipe.
__device__ __forceinline__ void dgrad_tile2(const char* lds, const bf16x8 (&a2)[2][2], int w2p_off0, int w2p_off1,
                                            f32x4 (&v)[2][4]) {
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) v[h][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 bq[2][3];
#pragma unroll
  for (int x = 0; x < 3; ++x) bq[0][x] = ld_b128(lds, w2p_off0 + x * W_IMG);
  D16_SETPRIO(1);
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int t = st >> 2, kb = st & 3;
    if (st < 7) {
      const int off = (((st + 1) >> 2) ? w2p_off1 : w2p_off0) + ((st + 1) & 3) * 2048;     // rows 16 kb + c
#pragma unroll
      for (int x = 0; x < 3; ++x) bq[(st + 1) & 1][x] = ld_b128(lds, off + x * W_IMG);
    }
#pragma unroll
    for (int x = 2; x >= 0; --x)                                                            // lo, mid, hi
#pragma unroll
      for (int h = 0; h < 2; ++h)
        v[h][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[h][t], bq[st & 1][x], v[h][kb], 0, 0, 0);
  }
  D16_SETPRIO(0);
}

// Sums of the dL/dh1 rows of every run of equal keys (sources in S, targets in T) inside a 32-edge tile, written as
// 256-byte "part" rows; the tile is processed as two 16-edge halves and a run that crosses the middle is carried
// in ONE register per lane.  m16: bit e set = edge e of this half is the last of its run.  Layout of every summed
// row: lane (c, g) holds column colp = 16 kRow4[g] + c (what red4 leaves in row g).
// The rows are given as factors: row value = v[kb][i] * gm[kb][i] (dgrad_masks) — multiplied inside the sums.
// part row `pidx` (wave-uniform: a scalar base), column `col` (32-bit lane offset): the saddr form of global_store
__device__ __forceinline__ void part_store(float* part, int64_t pidx, int col, float x) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(part + pidx * D16) + 4u * (uint32_t)col) = x;
}
__device__ __forceinline__ void run_sums(const f32x4 (&v)[4], const f32x4 (&gm)[4], unsigned m16, float& carry, float* part,
                                         int64_t& pidx, char* wv, int lane, int c, int g, int colp) {
  const unsigned inner = m16 & 0x7fffu;
  if (inner == 0u) {
    // no boundary before the last edge: one run (open or closing at edge 15) — by far the most frequent case with
    // chunked run sums: one multiply + three fused multiply-adds per column block
    float x[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      float t = v[kb][0] * gm[kb][0];
      t = fmaf(v[kb][1], gm[kb][1], t);
      t = fmaf(v[kb][2], gm[kb][2], t);
      x[kb] = fmaf(v[kb][3], gm[kb][3], t);
#ifdef PANGNN_D16_PROBE_OPAQUE_RUNSUM
      asm volatile("" : "+v"(x[kb]));
#endif
    }
    const float s = carry + red4(x[0], x[1], x[2], x[3]);
    if (m16 & 0x8000u) {
      part_store(part, pidx, colp, s);
      ++pidx;
      carry = 0.f;
    } else {
      carry = s;
    }
  } else if ((inner & (inner - 1u)) == 0u) {
    // one boundary inside: rows <= bnd close the first run, the rest is a second run (open or closing)
    const int bnd = __builtin_ctz(inner);
    float a[4], b[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      a[kb] = 0.f;
      b[kb] = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool first = 4 * g + i <= bnd;
        a[kb] = fmaf(v[kb][i], first ? gm[kb][i] : 0.f, a[kb]);
        b[kb] = fmaf(v[kb][i], first ? 0.f : gm[kb][i], b[kb]);
#ifdef PANGNN_D16_PROBE_OPAQUE_RUNSUM
        asm volatile("" : "+v"(a[kb]), "+v"(b[kb]));
#endif
      }
    }
    const float lo = carry + red4(a[0], a[1], a[2], a[3]);
    const float hi = red4(b[0], b[1], b[2], b[3]);
    part_store(part, pidx, colp, lo);
    ++pidx;
    if (m16 & 0x8000u) {
      part_store(part, pidx, colp, hi);
      ++pidx;
      carry = 0.f;
    } else {
      carry = hi;
    }
  } else {
    // three or more runs: the tile goes through LDS (the tile images are free at this point) as [16][64] floats,
    // the carried sum as row 16; lane = column walks the rows
    wave_sync();
    float* tile = reinterpret_cast<float*>(wv);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int i = 0; i < 4; ++i) tile[(4 * g + i) * 64 + 16 * kb + c] = v[kb][i] * gm[kb][i];
    tile[16 * 64 + colp] = carry;
    wave_sync();
    float sum = tile[16 * 64 + lane];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      sum += tile[e * 64 + lane];
      if ((m16 >> e) & 1u) {
        part_store(part, pidx, lane, sum);
        sum = 0.f;
        ++pidx;
      }
    }
    wave_sync();
    tile[lane] = sum;                  // the open remainder back into the carry layout
    wave_sync();
    carry = tile[colp];
    wave_sync();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// S kernel
// ------------------------------------------------------------------------------------------------------------------
struct HalfIn { uint32_t poff, qoff; float aux, w_e; int key, key_nxt; };
// The gathered row pieces of lane (c, g): columns 32 ks + 8 g + 0..7 of P[src_c] and Q[dst_c], ks = 0, 1.
// PQ16: the tables are stored as bfloat16 (config 5's autocast: mlp[0] is an autocast Linear, src/gnn.py:173) — half
// the gather bytes and registers; bf16 -> f32 is exact, so everything downstream is the f32-table arithmetic.
template <bool PQ16> struct HalfRowsT;
template <> struct HalfRowsT<false> { float4 p[4], q[4]; };
template <> struct HalfRowsT<true> { uint4 p[2], q[2]; };
typedef HalfRowsT<false> HalfRows;

// ids / label (or given gradient) / skip feature of the 16 edges of half `hx` of tile `tile`.  Addresses are a
// wave-uniform tile base (scalar registers) plus a 32-bit lane offset; edges past the end of the list (the last
// tile, and the prefetch of a tile the wave will not run) are clamped to the last edge.
template <bool EXTRA>
__device__ __forceinline__ HalfIn load_half(const D16Params& a, const float* aux, int64_t tile, int64_t n_tiles, int hx,
                                            int c) {
  HalfIn h;
  const int64_t tc = tile < n_tiles ? tile : n_tiles - 1;
  const int64_t e_tile = tc * 32;
  const int64_t rest = a.E - 1 - e_tile;
  const uint32_t lim = (uint32_t)(rest < 31 ? rest : 31);  // uniform: last valid position inside the tile
  const uint32_t lim_n = (uint32_t)(rest < 32 ? rest : 32);  // position 32 = first edge of the next tile
  const uint32_t k = min((uint32_t)(16 * hx + c), lim);
  const uint32_t kn = min(k + 1u, lim_n);                  // the edge after k
  // uniform tile bases + 32-bit lane byte offsets: the saddr form of global_load, no 64-bit vector arithmetic
  const char* es = reinterpret_cast<const char*>(a.ei + e_tile);
  const char* ed = reinterpret_cast<const char*>(a.ei + a.ld + e_tile);
  const int64_t s = *reinterpret_cast<const int64_t*>(es + 8u * k), d = *reinterpret_cast<const int64_t*>(ed + 8u * k);
  h.key = (int)s;
  h.key_nxt = (int)*reinterpret_cast<const int64_t*>(es + 8u * kn);
  h.poff = (uint32_t)s * a.ldp_b;
  h.qoff = (uint32_t)d * a.ldq_b;
  h.aux = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(aux + e_tile) + 4u * k);
  h.w_e = EXTRA ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.extra + e_tile) + 4u * k) : 0.f;
  return h;
}
__device__ __forceinline__ void issue_half_rows(const D16Params& a, const HalfIn& in, int g, HalfRowsT<false>& r) {
  const uint32_t po = in.poff + 32u * g, qo = in.qoff + 32u * g;
#pragma unroll
  for (int x = 0; x < 4; ++x) {                          // x = 2 ks + half: k = 32 ks + 8 g + 4 half ..
    r.p[x] = ld_row16(a.p, po + 128u * (x >> 1) + 16u * (x & 1));
    r.q[x] = ld_row16(a.q, qo + 128u * (x >> 1) + 16u * (x & 1));
  }
}
__device__ __forceinline__ void issue_half_rows(const D16Params& a, const HalfIn& in, int g, HalfRowsT<true>& r) {
  const uint32_t po = in.poff + 16u * g, qo = in.qoff + 16u * g;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {                       // 8 bf16 = 16 bytes: k = 32 ks + 8 g ..
    r.p[ks] = ld_row16u(a.p, po + 64u * ks);
    r.q[ks] = ld_row16u(a.q, qo + 64u * ks);
  }
}

// ---- first product + logit, shared by the training (S) and inference kernels so that both give bit-identical
// logits: h1 fragments of lane (c, g) -> split -> C[j][e] = b2[j] + sum_k W2[j][k] h1[e][k] (six partial products,
// smallest first) -> h2 = relu(C) left in acc (j = 16 jb + 4 g + i), logit of edge c in every lane group.
// PIPE: the W2 terms of step (ks, jb) are read one step ahead of their MFMAs (12 more registers: the inference
// kernel has them, the training kernel at 256 registers does not and lets its SIMD partner cover the LDS latency;
// the MFMA order, hence the result, is the same).
// M1: also pack the relu mask of h1 into `m1` — element s = 2 qd + half of K-step ks at bit 16 half + 7 - (4 ks + qd) — from
// the split's hi terms (the packed top halves of two neighbours: non-zero exactly when the element is a positive
// normal float), one v_pk_min_u16 + one shift-or per pair.
template <bool PIPE, bool M1 = false>
__device__ __forceinline__ float p1_logit(const char* lds, const float (&h)[2][8], int wfrag0, int wfrag1, int g,
                                          float b3v, f32x4 (&acc)[4], uint32_t one2 = 0u, uint32_t* m1 = nullptr) {
  const float* b2l = reinterpret_cast<const float*>(lds + LDS_VEC);
  const float* w3l = b2l + 64;
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) acc[jb] = *reinterpret_cast<const f32x4*>(b2l + 16 * jb + 4 * g);
  bf16x8 aq[2][3];
  if (PIPE) {
#pragma unroll
    for (int x = 0; x < 3; ++x) aq[0][x] = ld_b128(lds, LDS_W2 + wfrag0 + x * W_IMG);
  }
  Split3 hbk;                                       // the terms of one K-step at a time (12 registers)
  D16_SETPRIO(1);
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int ks = st >> 2, jb = st & 3;
    if (jb == 0) {
      hbk = split8(h[ks]);
      if (M1) {
        const u32x4 hw = __builtin_bit_cast(u32x4, hbk.hi);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) *m1 = (*m1 << 1) | nz16x2(hw[qd], one2);
      }
    }
    if (PIPE) {
      if (st < 7) {
        const int off = LDS_W2 + (((st + 1) >> 2) ? wfrag1 : wfrag0) + ((st + 1) & 3) * 2048;
#pragma unroll
        for (int x = 0; x < 3; ++x) aq[(st + 1) & 1][x] = ld_b128(lds, off + x * W_IMG);
      }
    } else {
      const int off = LDS_W2 + (ks ? wfrag1 : wfrag0) + jb * 2048;
#pragma unroll
      for (int x = 0; x < 3; ++x) aq[st & 1][x] = ld_b128(lds, off + x * W_IMG);
    }
    const bf16x8 w_hi = aq[st & 1][0], w_mid = aq[st & 1][1], w_lo = aq[st & 1][2];
    acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_lo, hbk.hi, acc[jb], 0, 0, 0);
    acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_mid, hbk.mid, acc[jb], 0, 0, 0);
    acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_hi, hbk.lo, acc[jb], 0, 0, 0);
    acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_mid, hbk.hi, acc[jb], 0, 0, 0);
    acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_hi, hbk.mid, acc[jb], 0, 0, 0);
    acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_hi, hbk.hi, acc[jb], 0, 0, 0);
  }
  D16_SETPRIO(0);
  float part = 0.f;
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) {
    const f32x4 ww = *reinterpret_cast<const f32x4*>(w3l + 16 * jb + 4 * g);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[jb][i] = relu1(acc[jb][i]);
      part = fmaf(acc[jb][i], ww[i], part);
    }
  }
  return xsum32(xsum16(part)) + b3v;
}
// h1 fragments from the gathered row pieces: h = relu(p + q (+ w_e c))
__device__ __forceinline__ void sum_rows(const HalfRowsT<false>& rows, float (&h)[2][8]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const float4 pv = rows.p[2 * ks + hf], qv = rows.q[2 * ks + hf];
      h[ks][4 * hf + 0] = pv.x + qv.x;
      h[ks][4 * hf + 1] = pv.y + qv.y;
      h[ks][4 * hf + 2] = pv.z + qv.z;
      h[ks][4 * hf + 3] = pv.w + qv.w;
    }
}
__device__ __forceinline__ void sum_rows(const HalfRowsT<true>& rows, float (&h)[2][8]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const uint32_t pw[4] = {rows.p[ks].x, rows.p[ks].y, rows.p[ks].z, rows.p[ks].w};
    const uint32_t qw[4] = {rows.q[ks].x, rows.q[ks].y, rows.q[ks].z, rows.q[ks].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#ifdef PANGNN_D16_TABLES_F16
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));              // IEEE half: v_cvt_f32_f16, exact
      const h2 pv = __builtin_bit_cast(h2, pw[j]), qv = __builtin_bit_cast(h2, qw[j]);
      h[ks][2 * j] = (float)pv[0] + (float)qv[0];
      h[ks][2 * j + 1] = (float)pv[1] + (float)qv[1];
#else
      h[ks][2 * j] = __builtin_bit_cast(float, pw[j] << 16) + __builtin_bit_cast(float, qw[j] << 16);
      h[ks][2 * j + 1] = __builtin_bit_cast(float, pw[j] & 0xffff0000u) + __builtin_bit_cast(float, qw[j] & 0xffff0000u);
#endif
    }
  }
}
template <bool PQ16>
__device__ __forceinline__ void h1_frags(const HalfRowsT<PQ16>& rows, bool has_extra, float w_e, const float* cvl, int g,
                                         float (&h)[2][8]) {
  sum_rows(rows, h);
  if (has_extra) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int s = 0; s < 8; ++s) h[ks][s] = fmaf(w_e, cvl[32 * ks + 8 * g + s], h[ks][s]);
  }
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int s = 0; s < 8; ++s) h[ks][s] = relu1(h[ks][s]);
}

template <bool FUSED_LOSS, bool RUNSUM, bool PQ16, bool EXTRA>
__global__ __launch_bounds__(S_WAVES * 64) void decoder_train16_kernel(
    D16Params a, const float* __restrict__ g_logits, D16Loss lp, float* __restrict__ logits, D16Run rs,
    uint32_t* __restrict__ rec, float* __restrict__ slabs, int64_t n_tiles) {
  __shared__ __attribute__((aligned(16))) char lds[S_LDS];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int c = lane & 15, g = lane >> 4;
  char* wv = lds + LDS_WAVE0 + wave * WV_BYTES;
  D16_STAMP(0);
  stage_weights16(a.w2, a.b2, a.w3, a.cvec, lds, S_WAVES * 64, 3, LDS_W2P, LDS_VEC);
  D16_STAMP(1);
  __syncthreads();
  D16_STAMP(2);
  const float* w3l = reinterpret_cast<const float*>(lds + LDS_VEC) + 64;
  const float* cvl = w3l + 64;

  // lane-constant LDS offsets
  const int wfrag0 = c * 128 + ((g ^ wsw(c)) << 4);                    // weight images: row 16 x + c, chunk g (K-step 0)
  const int wfrag1 = wfrag0 ^ 64;                                      //                               chunk 4 + g
  const int hgw0 = c * 128 + ((g ^ tsw(c)) << 4);                      // tile images: row c, chunk g / 4 + g (16-B writes)
  const int hgw1 = hgw0 ^ 64;
  const int m2w = c * 128 + ((((g >> 1)) ^ tsw(c)) << 4) + 8 * (g & 1); // m2 image: row c, 8 B at column 4 g (+ 16 jb: ^ (jb << 5))
  const int colp = 16 * (((g & 1) << 1) | (g >> 1)) + c;               // column a lane holds after red4: block {0, 2, 1, 3}[g]
  // transposing reads of the 32x32x16 operands (P3): 16-lane group q4, block rows 8 (q4 >> 1) + (li >> 2) (+4),
  // columns 32 blk + 16 (q4 & 1) + 4 (li & 3)
  int tr3h[2];                             // column block 0; block 1 is the same offset ^ 64 (chunk bit 2)
  {
    const int li = lane & 15, q4 = lane >> 4, p = li & 3;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int row = 8 * (q4 >> 1) + (li >> 2) + 4 * half;
      const int ch = 2 * (q4 & 1) + (p >> 1);
      tr3h[half] = row * 128 + ((ch ^ tsw(row)) << 4) + 8 * (p & 1);
    }
  }

  f32x16 acc3[2][2];                       // sum_e m2[j][e] g_e h1[e][k]:  j = 32 mb + jr(i, h), k = 32 nb + (lane & 31)
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc3[x][0][i] = 0.f; acc3[x][1][i] = 0.f; }
  f32x4 gw3a[4];                           // per (j = 16 jb + 4 g + i, edge slot c) partial sums of g_e h2 over tiles
#pragma unroll                             // (dL/db2 = w3[j] sum_e g_e m2[j][e] is left to the dgrad pass: it needs only the records)
  for (int jb = 0; jb < 4; ++jb) gw3a[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float gcv[4] = {0.f, 0.f, 0.f, 0.f};     // lane (c, g): partial of gcvec[16 kb + c] over its edges
  float gb3p = 0.f, lossp = 0.f;
  const float b3v = a.b3[0];
  const float pw = (FUSED_LOSS && lp.pos_weight) ? lp.pos_weight[0] : 1.f;
  const int64_t e_live = a.e_live ? min(a.e_live[0], a.E) : a.E;                       // uniform (one scalar load)
  const float inv_denom = a.e_live ? 1.0f / (float)max(e_live, (int64_t)1) : lp.inv_denom;
  constexpr bool has_extra = EXTRA;              // skip connections: compile-time, like the other shape switches
  const float* auxp = FUSED_LOSS ? lp.y : g_logits;
  uint32_t one2 = 0x00010001u;             // opaque to the optimiser: the packed min against it stays ONE v_pk_min_u16 (a
  asm volatile("" : "+v"(one2));           // literal is rewritten as compares + selects); the statement emits no instruction

  const int clog = rs.chunk_log;                                        // uniform: tiles per chunk = 1 << clog
  const int64_t cstride = (int64_t)gridDim.x * S_WAVES;                // in chunks
  const int64_t n_chunks = (n_tiles + (1 << clog) - 1) >> clog;
  int64_t tile = ((int64_t)blockIdx.x * S_WAVES + wave) << clog;
  HalfIn in_cur = load_half<EXTRA>(a, auxp, tile, n_tiles, 0, c);
  HalfRowsT<PQ16> rows;
  issue_half_rows(a, in_cur, g, rows);
  int poff_cur = (RUNSUM && tile < n_tiles) ? rs.part_off[tile >> clog] : 0;
  float carry = 0.f;
  int64_t pidx = 0;
  D16_STAMP(3);
  D16_CYC_DECL;

  while (tile < n_tiles) {
    // chunk bookkeeping as selects (no branch in the loop body): a chunk's first tile starts from its part offset with
    // no open run; the offset of the wave's next chunk is fetched every tile (one cached dword) and taken over at the end
    const bool first_of_chunk = (tile & ((1 << clog) - 1)) == 0;
    carry = first_of_chunk ? 0.f : carry;
    pidx = first_of_chunk ? (int64_t)__builtin_amdgcn_readfirstlane(poff_cur) : pidx;
    bool last_tile;
    const int64_t tile_nxt = next_tile(tile, cstride, n_tiles, last_tile, clog);
    const int64_t chunk_nxt = (tile >> clog) + cstride;
    const int poff_nxt = RUNSUM ? rs.part_off[chunk_nxt < n_chunks ? chunk_nxt : n_chunks - 1] : 0;
    const int store_lim = (int)min((int64_t)31, a.E - 1 - tile * 32);  // uniform: positions <= store_lim exist in the list
    const int live_lim = (int)min((int64_t)store_lim, e_live - 1 - tile * 32);   // ... <= live_lim are real edges (may be < 0)
    uint32_t* rec_tile = rec + tile * 256;                             // 8 dwords per edge (rec is required)
    float* logit_tile = logits + tile * 32;                            // (required with the fused loss, optional otherwise)
#pragma unroll 1
    for (int hx = 0; hx < 2; ++hx) {
      // ids of the next half tile (this tile's second half, or the first half of the wave's next tile)
      D16_CYC(5);                          // phase 5: loop / chunk bookkeeping between half tiles
      const HalfIn in_nxt = load_half<EXTRA>(a, auxp, hx == 0 ? tile : tile_nxt, n_tiles, hx ^ 1, c);    // one load site
      const int pos = 16 * hx + c;
      const bool live = pos <= live_lim;

      // h1 fragments (B operand of P1): lane (c, g) holds h1[c][32 ks + 8 g + 0..7]
      float h[2][8];
      h1_frags(rows, has_extra, in_cur.w_e, cvl, g, h);

      // ---- P1: C[j][e] = b2[j] + sum_k W2[j][k] h1[e][k]
      f32x4 acc[4];
      uint32_t m1 = 0;                     // relu mask bits of h1 (packed inside the first product, off its split terms)
      D16_STAMP(4 + 4 * hx);
      D16_CYC(0);                          // phase 0: next ids requested, the gathered rows of THIS half awaited, h1 = relu(p + q)
      const float xv = p1_logit<false, true>(lds, h, wfrag0, wfrag1, g, b3v, acc, one2, &m1);
      D16_STAMP(5 + 4 * hx);
      D16_CYC(1);                          // phase 1: P1 (splits, 48 matrix instructions, relu, w3 dot, lane sums)
      // the rows of the next half tile fly during the epilogue and the other two products
      issue_half_rows(a, in_nxt, g, rows);

      // ---- loss, g_e.  A lane past the end of the list looks at the list's last edge again (clamped ids): g_raw is
      // that edge's dL/dlogit, bit-identical to what its own lane computes, so records and logits are stored by
      // every lane without a predicate (a dead lane rewrites the last edge's values); only g_e, which feeds the sums,
      // is zeroed.  No divergent store, hence no branch, in the loop body.
      float g_e, g_raw;
      if (FUSED_LOSS) {
        const float y_e = in_cur.aux;
        const float scale = live ? inv_denom : 0.f;
        const float lw = 1.f + (pw - 1.f) * y_e;
        // t = exp(-|x|) in (0, 1] as ONE v_exp_f32 of -|x| log2(e) (no over / underflow to guard on this side; the single
        // rounding of the scaled argument moves t by <= |x| 2^-24 relative, i.e. sigmoid by <= 2e-8 absolute at its worst
        // point |x| = 1): the library expf's two-term range reduction and its range selects were 11 more instructions
        const float t = __expf(-fabsf(xv));
        const float u = 1.f + t;
        float ru = __builtin_amdgcn_rcpf(u);
        ru = ru * (2.f - u * ru);                     // 1 / (1 + t): one Newton step on the hardware reciprocal
        const float sig_neg = xv >= 0.f ? t * ru : ru;                              // sigmoid(-x)
        g_raw = ((1.f - y_e) - lw * sig_neg) * inv_denom;
        g_e = live ? g_raw : 0.f;
        // log1p(t) = -log(1 / (1 + t)) off the reciprocal the gradient needs anyway: absolute error <= 1e-7 per term of a
        // MEAN of O(1) terms (the loss value only; round 3 evaluated log(u) t / (u - 1) with a second Newton reciprocal)
        const float l1p = -__logf(ru);
        lossp = fmaf((1.f - y_e) * xv + lw * (l1p + fmaxf(-xv, 0.f)), scale, lossp);
      } else {
        g_raw = in_cur.aux;
        g_e = live ? g_raw : 0.f;
      }
      gb3p += g_e;
      const int posc = min(pos, store_lim);
      if (FUSED_LOSS || logits != nullptr)                                // four lane groups, same value
        *reinterpret_cast<float*>(reinterpret_cast<char*>(logit_tile) + 4u * (uint32_t)posc) = xv;

      // ---- m2 = [h2 > 0] as bf16 0/1 (A operand of P2 and, transposed through LDS, of P3); gw3 partials
      bf16x8 a2[2];
      uint32_t m2 = 0;
      {
        u32x4 aw[2];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const float h2a = acc[jb][2 * pr], h2b = acc[jb][2 * pr + 1];
            gw3a[jb][2 * pr] = fmaf(g_e, h2a, gw3a[jb][2 * pr]);
            gw3a[jb][2 * pr + 1] = fmaf(g_e, h2b, gw3a[jb][2 * pr + 1]);
            // the pair as two 0/1 halves: top halves packed by one v_perm_b32, [!= 0] by one v_pk_min_u16 (h2 >= 0 after
            // the relu: its top half is non-zero exactly for a positive normal float)
            const uint32_t t2 = nz16x2(__builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, h2b), __builtin_bit_cast(uint32_t, h2a),
                                                             0x07060302u), one2);
            m2 = (m2 << 1) | t2;
            aw[jb >> 1][2 * (jb & 1) + pr] = __umul24(t2, 0x3f80u);    // bf16 1.0 / 0.0 (v_mul_u32_u24: full rate; the
                                                                      // optimiser cannot see t2 < 2^17 behind `one2`)
          }
        a2[0] = __builtin_bit_cast(bf16x8, aw[0]);
        a2[1] = __builtin_bit_cast(bf16x8, aw[1]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        struct P { short4v lo, hi; };
        const P pk = __builtin_bit_cast(P, a2[t]);
        *reinterpret_cast<short4v*>(wv + WV_M2 + (m2w ^ ((2 * t) << 5))) = pk.lo;        // columns 16 (2t) + 4 g ..
        *reinterpret_cast<short4v*>(wv + WV_M2 + (m2w ^ ((2 * t + 1) << 5))) = pk.hi;    // columns 16 (2t+1) + 4 g ..
      }
      // ---- Hg = g_e h1 split three ways -> tile images (B operand of P3), one K-step (= one 32-column block of
      // dL/dW2) at a time: the matrix instructions of block 0 run while the vector unit splits block 1 and packs the
      // relu bits, those of block 1 while it starts on P2
      auto write_hg = [&](int ks) {
        float hg[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) hg[s] = g_e * h[ks][s];
        const Split3 sb = split8(hg);
        const int off = WV_HG + (ks ? hgw1 : hgw0);
        *reinterpret_cast<bf16x8*>(wv + off) = sb.hi;
        *reinterpret_cast<bf16x8*>(wv + off + T_IMG) = sb.mid;
        *reinterpret_cast<bf16x8*>(wv + off + 2 * T_IMG) = sb.lo;
      };
      // P3: acc3[mb][nb] += m2^T (Hg_lo + Hg_mid + Hg_hi) of column block nb, K = the 16 edges
      bf16x8 am[2];
      auto p3_block = [&](int nb) {
#pragma unroll
        for (int term = 2; term >= 0; --term) {
          const bf16x8 bh = ld_tr8(wv + WV_HG + term * T_IMG, tr3h[0] ^ (nb << 6), tr3h[1] ^ (nb << 6));
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
            acc3[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[mb], bh, acc3[mb][nb], 0, 0, 0);
        }
      };
      write_hg(0);
      wave_sync();
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) am[mb] = ld_tr8(wv + WV_M2, tr3h[0] ^ (mb << 6), tr3h[1] ^ (mb << 6));
      D16_SETPRIO(1);
      p3_block(0);
      write_hg(1);
      // record dword g of the edge: m2 bits in the low byte of each half (bit 7 - n / 23 - n for the pair n), m1 bits in the
      // high byte — the T kernel turns the m2 bits into bf16 0 / 1 with one AND + one 24-bit multiply per dword
      const uint32_t recw = (m2 & 0x00ff00ffu) | ((m1 & 0x00ff00ffu) << 8);
      *reinterpret_cast<uint32_t*>(wv + WV_REC + 16 * c + 4 * g) = recw;
      *reinterpret_cast<float*>(wv + WV_GL + 4 * c) = g_e;               // the four lane groups write the same value
      if (has_extra) *reinterpret_cast<float*>(wv + WV_WL + 4 * c) = in_cur.w_e;
      {
        // scalar tile base + 32-bit lane offset (saddr stores); dL/dlogit goes into dwords 4 .. 7: lane group g of the T
        // kernel reads its mask dword and g_e through ONE address (offsets 0 and 16)
        char* r = reinterpret_cast<char*>(rec_tile) + (32u * (uint32_t)posc + 4u * (uint32_t)g);
        *reinterpret_cast<uint32_t*>(r) = recw;
        *reinterpret_cast<uint32_t*>(r + 16) = __builtin_bit_cast(uint32_t, g_raw);
      }
      wave_sync();
      p3_block(1);
      D16_SETPRIO(0);

      D16_STAMP(6 + 4 * hx);
      D16_CYC(2);                          // phase 2: row requests of the next half, loss, m2 operands, g_e h1 splits, P3, records
      // ---- P2 + run sums by source
      if (RUNSUM || has_extra) {
        f32x4 v[4], gm[4];
        dgrad_tile<false>(lds, wv + WV_REC, wv + WV_GL, a2, c, g, LDS_W2P + wfrag0, LDS_W2P + wfrag1, v, gm);
#ifdef PANGNN_D16_DEBUG
        if (d16_dbg_v != nullptr) {
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (tile * 32 + 16 * hx + 4 * g + i < a.E)
                d16_dbg_v[(tile * 32 + 16 * hx + 4 * g + i) * 64 + 16 * kb + c] = v[kb][i] * gm[kb][i];
        }
#endif
        if (has_extra) {
          // the skip feature's gradient needs the rows themselves: multiplied out on the side (config 5's instances only);
          // the run sums below still take the factors, so that they are the same arithmetic in every instance
          f32x4 pv[4];
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) pv[kb] = v[kb];
          dgrad_apply(pv, gm);
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(wv + WV_WL + 16 * g);
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              gcv[kb] = fmaf(w4[i], pv[kb][i], gcv[kb]);
#ifdef PANGNN_D16_PROBE_OPAQUE_GCV
              asm volatile("" : "+v"(gcv[kb]));
#endif
            }
        }
        if (RUNSUM) {
          const bool closes = in_cur.key != in_cur.key_nxt || (pos == 31 && last_tile);     // key change, or end of the chunk
          const unsigned long long bal = __ballot(closes);
          const unsigned m16 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bal & 0xffffull));
          run_sums(v, gm, m16, carry, rs.part, pidx, wv, lane, c, g, colp);
        }
      }
      D16_CYC(3);                          // phase 3: P2 (24 matrix instructions), mask factors, run sums
      wave_sync();          // the next half tile overwrites the images / recl / gl
      D16_STAMP(7 + 4 * hx);
      D16_CYC(4);                          // phase 4: the wave barrier that frees the tile images
#ifdef PANGNN_D16_CYC
      ++cyc_halves;
#endif
      in_cur = in_nxt;
    }
    poff_cur = last_tile ? poff_nxt : poff_cur;
    tile = tile_nxt;
  }

#ifdef PANGNN_D16_CYC
  if (d16_cyc != nullptr && blockIdx.x == 0 && lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) d16_cyc[wave * 8 + i] = cyc_acc[i];
    d16_cyc[wave * 8 + 6] = cyc_halves;
    d16_cyc[wave * 8 + 7] = cyc_last - cyc_t0;
  }
#endif
  // ---- finish: per-lane partials -> workgroup slab.  The eight waves are added as a fixed binary tree,
  // ((w0 + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7)), through LDS: in round d the waves with bit d set (lower bits clear)
  // write their values, their partners d below add them — every value at its own address (value v of lane l at
  // [v][l]: conflict-free, no read-modify-write chains), three rounds.  (Rounds 2-4 let the waves take turns adding into one
  // LDS image: eight serialised rounds of 64 dependent read-add-write steps, ~20 us — nothing at 75 M edges per launch,
  // most of the kernel on a 5 000-edge mini-batch.)
  // gw3[j] = sum over the 16 edge slots c
#pragma unroll
  for (int jb = 0; jb < 4; ++jb)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) gw3a[jb][i] += __shfl_xor(gw3a[jb][i], off);
    }
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) gcv[kb] = xsum32(xsum16(gcv[kb]));
  if (g != 0) { gb3p = 0.f; lossp = 0.f; }                   // the four lane groups carried the same per-edge values
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    gb3p += __shfl_xor(gb3p, off);
    lossp += __shfl_xor(lossp, off);
  }
  D16_STAMP(12);
  constexpr int NV = 64 + 16 + 4 + 2;                        // values per lane: dL/dW2 tile | gw3 | gcvec | gb3, loss
  float val[NV];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) val[32 * mb + 16 * nb + i] = acc3[mb][nb][i];
#pragma unroll
  for (int jb = 0; jb < 4; ++jb)
#pragma unroll
    for (int i = 0; i < 4; ++i) val[64 + 4 * jb + i] = gw3a[jb][i];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) val[80 + kb] = gcv[kb];
  val[84] = gb3p;
  val[85] = lossp;
  const int hh = lane >> 5, r = lane & 31;
  float w3r[2][16];                                          // w3 of this lane's 32 rows of dL/dW2 (its LDS copy goes under the tree buffers)
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) w3r[mb][i] = w3l[32 * mb + (i & 3) + 8 * (i >> 2) + 4 * hh];
  __syncthreads();                                           // every wave is done with the weight / tile images
  float* buf = reinterpret_cast<float*>(lds);                // [4 writers][NV][64 lanes] = 88 KB of the 116 KB
  static_assert(4 * NV * 64 * 4 <= S_LDS, "tree buffers");
#pragma unroll
  for (int d = 1; d < S_WAVES; d <<= 1) {
    const bool writer = (wave & (2 * d - 1)) == d, reader = (wave & (2 * d - 1)) == 0;
    float* mine = buf + (size_t)((wave >> 1) & 3) * (NV * 64) + lane;   // writer w and its reader w - d share slot (w >> 1) & 3 ...
    if (d == 2) mine = buf + (size_t)(wave >> 2) * (NV * 64) + lane;    // ... (round 1: pairs (0,1) (2,3) (4,5) (6,7); round 2: (0,2) (4,6))
    if (d == 4) mine = buf + lane;
    if (writer) {
#pragma unroll
      for (int v = 0; v < NV; ++v) mine[v * 64] = val[v];
    }
    __syncthreads();
    if (reader) {
#pragma unroll
      for (int v = 0; v < NV; ++v) val[v] += mine[v * 64];
    }
    __syncthreads();
  }
  D16_STAMP(13);
  // wave 0 lays the sums out as the slab in LDS (the tree buffers are free again), every thread copies its share out:
  // coalesced stores from 512 threads — written by wave 0 alone (68 stores per lane behind 32 dependent loads of w3) the
  // slab took 7 us of a 25 us mini-batch launch
  float* img = buf;
  if (wave == 0) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int j = 32 * mb + (i & 3) + 8 * (i >> 2) + 4 * hh;
          img[j * 64 + r + 32 * nb] = w3r[mb][i] * val[32 * mb + 16 * nb + i];
        }
    if (c == 0) {
#pragma unroll
      for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int j = 16 * jb + 4 * g + i;
          img[4096 + j] = 0.f;                                    // dL/db2: dgrad pass
          img[4096 + 64 + j] = val[64 + 4 * jb + i];
        }
    }
    if (g == 0) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) img[4096 + 128 + 16 * kb + c] = val[80 + kb];
    }
    if (lane == 0) {
      img[4096 + 192] = val[84];
      img[4096 + 193] = val[85];
    }
  }
  __syncthreads();
  float* slab = slabs + (int64_t)blockIdx.x * SLAB16;
  for (int i = threadIdx.x; i < 4096 + 194; i += S_WAVES * 64) slab[i] = img[i];
  D16_STAMP(14);
}

// ------------------------------------------------------------------------------------------------------------------
// T kernel: dL/dh1 summed over runs of equal keys in a permuted edge order, from the records of S.
//   perm[k] = edge id at sorted position k (NULL = identity); keys[k] = run key of position k (CSR row id).
// ------------------------------------------------------------------------------------------------------------------
// Two dependent loads per position (perm[k], then the record of edge perm[k]): a two-stage pipeline — the ids of half
// tile n + 2 and the records of half tile n + 1 are in flight while half tile n is multiplied, so each of the two
// random-access latencies has a whole iteration to land.
// Addressing (round 4): every per-position load is `scalar tile base + 32-bit lane offset` (the saddr form of
// global_load: no 64-bit vector arithmetic), and a record is read through ONE 64-bit address per lane — mask dword g at
// +0, g_e at +16 (S replicates dL/dlogit into dwords 4 .. 7) — instead of three 64-bit adds per load.
struct TIds { uint32_t e; int key, key_nxt; };
struct TIn { uint32_t recw; float g_e; int key, key_nxt; };
template <bool PERM, bool KEYS>
__device__ __forceinline__ TIds load_ids(const int32_t* perm, const int32_t* keys, int64_t E, int64_t tile, int64_t n_tiles,
                                         int hx, int c) {
  TIds t;
  const int64_t tc = tile < n_tiles ? tile : n_tiles - 1;
  const int64_t p_tile = tc * 32;
  const int64_t rest = E - 1 - p_tile;
  const uint32_t lim = (uint32_t)(rest < 31 ? rest : 31);                 // uniform: last valid position inside the tile
  const uint32_t lim_n = (uint32_t)(rest < 32 ? rest : 32);               // position 32 = first position of the next tile
  const uint32_t k = min((uint32_t)(16 * hx + c), lim);
  const uint32_t kn = min(k + 1u, lim_n);
  t.key = t.key_nxt = 0;
  if (KEYS) {                                                             // (the parameter sums alone take no keys: NULL)
    const char* kt = reinterpret_cast<const char*>(keys + p_tile);        // uniform base, 32-bit lane offsets
    t.key = *reinterpret_cast<const int32_t*>(kt + 4u * k);
    t.key_nxt = *reinterpret_cast<const int32_t*>(kt + 4u * kn);
  }
  t.e = PERM ? (uint32_t) * reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(perm + p_tile) + 4u * k)
             : (uint32_t)p_tile + k;
  return t;
}
// recg: the record table + 4 g bytes (this lane's mask dword of record 0)
__device__ __forceinline__ TIn load_rec(const char* recg, const TIds& id) {
  TIn t;
  const char* r = recg + ((uint64_t)(id.e << 1) << 4);     // e < 2^31; a shift of 4 folds into v_lshl_add_u64, one of 5 does not
  t.recw = *reinterpret_cast<const uint32_t*>(r);
  t.g_e = *reinterpret_cast<const float*>(r + 16);
  t.key = id.key;
  t.key_nxt = id.key_nxt;
  return t;
}

// PERM (a permutation is given) / RUN (run sums wanted; false: the parameter sums alone) are compile-time, so that the
// loads of the loop body sit in straight-line code.  Whether dL/db2 is wanted stays a runtime-uniform branch: as a template
// parameter the scheduler's longer reach costs its 16 accumulators 25 spilled registers at the 128-register cap.
template <bool PERM, bool RUN>
__global__ __launch_bounds__(T_WAVES * 64) void decoder_dgrad16_kernel(
    const uint32_t* __restrict__ rec, const int32_t* __restrict__ perm, const int32_t* __restrict__ keys,
    const float* __restrict__ w2, const float* __restrict__ w3, int64_t E, const int64_t* __restrict__ e_live_p, D16Run rs,
    float* __restrict__ gb2_slabs, int64_t n_tiles) {
  // LDS: W2'^T hi | mid | lo, (b2 w3 cvec: w3 used by the dL/db2 finish), per wave: run-sum tile [17][64] floats | recl |
  // gl of both halves of the wave's 32-edge tile
  constexpr int TW_REC = 17 * 64 * 4, TW_GL = TW_REC + 2 * 256, TW_BYTES = TW_GL + 2 * 64;
  constexpr int T_VEC = 3 * W_IMG, T_WAVE0 = T_VEC + 3 * 64 * 4;
  constexpr int T_LDS = T_WAVE0 + T_WAVES * TW_BYTES;
  __shared__ __attribute__((aligned(16))) char lds[T_LDS];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int c = lane & 15, g = lane >> 4;
  char* wv = lds + T_WAVE0 + wave * TW_BYTES;
  stage_weights16(w2, nullptr, w3, nullptr, lds, T_WAVES * 64, 2, 0, T_VEC);
  __syncthreads();
  const int wfrag0 = c * 128 + ((g ^ wsw(c)) << 4), wfrag1 = wfrag0 ^ 64;
  const int colp = 16 * (((g & 1) << 1) | (g >> 1)) + c;
  const char* recg = reinterpret_cast<const char*>(rec) + 4 * g;
  // positions >= e_live of the order are padding (the pads of a fixed-shape batch sort last): dL/dlogit = 0 there
  const int64_t e_live = e_live_p ? min(e_live_p[0], E) : E;
  f32x4 gb2a[4];                           // per (j = 16 jb + 4 g + i, edge slot c): sum of g_e m2[j][e] over tiles
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) gb2a[jb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int clog = rs.chunk_log;
  const int64_t cstride = (int64_t)gridDim.x * T_WAVES;                // in chunks
  const int64_t n_chunks = (n_tiles + (1 << clog) - 1) >> clog;
  int64_t tile = ((int64_t)blockIdx.x * T_WAVES + wave) << clog;
  // pipeline over the wave's tiles: records of tile n + 1 and ids of tile n + 2 in flight under tile n
  TIn cur[2];
  TIds ids_nxt[2];
  bool last1;
  int64_t tile1 = next_tile(tile, cstride, n_tiles, last1, clog);      // the wave's next tile
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    cur[h] = load_rec(recg, load_ids<PERM, RUN>(perm, keys, E, tile, n_tiles, h, c));
    ids_nxt[h] = load_ids<PERM, RUN>(perm, keys, E, tile1, n_tiles, h, c);
  }
  int poff_cur = (RUN && tile < n_tiles) ? rs.part_off[tile >> clog] : 0;
  float carry = 0.f;
  int64_t pidx = 0;
  while (tile < n_tiles) {
    const bool first_of_chunk = (tile & ((1 << clog) - 1)) == 0;
    carry = first_of_chunk ? 0.f : carry;
    pidx = first_of_chunk ? (int64_t)__builtin_amdgcn_readfirstlane(poff_cur) : pidx;
    const bool last_tile = last1;                                      // this tile ends its chunk
    const int64_t chunk_nxt = (tile >> clog) + cstride;
    const int poff_nxt = RUN ? rs.part_off[chunk_nxt < n_chunks ? chunk_nxt : n_chunks - 1] : 0;
    const int live_lim = (int)min((int64_t)31, e_live - 1 - tile * 32);
    bool last2;
    const int64_t tile2 = next_tile(tile1 < n_tiles ? tile1 : n_tiles - 1, cstride, n_tiles, last2, clog);
    TIn nxt[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      nxt[h] = load_rec(recg, ids_nxt[h]);
      ids_nxt[h] = load_ids<PERM, RUN>(perm, keys, E, tile2, n_tiles, h, c);
    }
    bf16x8 a2[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float g_e = 16 * h + c <= live_lim ? cur[h].g_e : 0.f;
      const uint32_t recw = cur[h].recw;
      *reinterpret_cast<uint32_t*>(wv + TW_REC + 256 * h + 16 * c + 4 * g) = recw;
      *reinterpret_cast<float*>(wv + TW_GL + 64 * h + 4 * c) = g_e;
      // m2 bits of lane (c, g): pair n = 4 t + qd (elements 2 qd, 2 qd + 1 of K-step t) sits at bits 7 - n and 23 - n:
      // (recw & (0x10001 << p)) * (0x3f80 >> p) = bf16 1.0 / 0.0 in each half (p <= 7: the products stay inside their halves)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        u32x4 w;
        uint32_t yb[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int p = 7 - (4 * t + qd);
          yb[qd] = recw & (0x00010001u << p);
          asm volatile("" : "+v"(yb[qd]));      // opaque: keeps ONE and per pair (the optimiser otherwise re-derives byte 0
          w[qd] = __umul24(yb[qd], 0x3f80u >> p);     //                                  from recw with a second and below)
        }
        a2[h][t] = __builtin_bit_cast(bf16x8, w);
        if (gb2_slabs != nullptr) {
          // dL/db2 partials: the masked record bits themselves as floats — byte 0 / byte 2 of (recw & (0x10001 << p)) is
          // 2^p or 0, one v_cvt_f32_ubyte each — so accumulator (jb, i) carries 2^p times its sum, p = 7 - 2 jb - (i >> 1),
          // an exact scale taken out once at the end.  (Round 3 rebuilt 1.0 / 0.0 from the bf16 operand: the optimiser
          // folded that into a v_mul_lo_u32 per pair, a quarter-rate instruction, 16 per tile.)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const int jb = 2 * t + (qd >> 1), i0 = 2 * (qd & 1);
            gb2a[jb][i0] = fmaf((float)(yb[qd] & 0xffu), g_e, gb2a[jb][i0]);
            gb2a[jb][i0 + 1] = fmaf((float)((yb[qd] >> 16) & 0xffu), g_e, gb2a[jb][i0 + 1]);
          }
        }
      }
    }
    wave_sync();
    if (RUN) {
      f32x4 v[2][4];
      dgrad_tile2(lds, a2, wfrag0, wfrag1, v);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 gm[4];
        dgrad_masks(wv + TW_REC + 256 * h, wv + TW_GL + 64 * h, c, g, gm);
        const bool closes = cur[h].key != cur[h].key_nxt || (h == 1 && c == 15 && last_tile);   // key change, or end of the chunk
        const unsigned long long bal = __ballot(closes);
        const unsigned m16 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bal & 0xffffull));
        run_sums(v[h], gm, m16, carry, rs.part, pidx, wv, lane, c, g, colp);
      }
    }
    wave_sync();          // the next tile overwrites recl / gl
#pragma unroll
    for (int h = 0; h < 2; ++h) cur[h] = nxt[h];
    poff_cur = last_tile ? poff_nxt : poff_cur;
    tile = tile1;
    tile1 = tile2;
    last1 = last2;
  }
  if (gb2_slabs != nullptr) {
    // dL/db2 partials of the workgroup in fixed wave order
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) gb2a[jb][i] += __shfl_xor(gb2a[jb][i], off);
      }
    // every wave writes its 64 sums to its own row, then 64 threads add the 16 rows in wave order (one round; rounds 2-4
    // let the waves take turns: 16 serialised rounds of dependent LDS read-add-writes, most of a mini-batch launch)
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds + T_WAVE0);
    const float* w3v = reinterpret_cast<const float*>(lds + T_VEC) + 64;
    if (c == 0) {
#pragma unroll
      for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float unscale = __builtin_bit_cast(float, (uint32_t)(127 - (7 - 2 * jb - (i >> 1))) << 23);   // 2^-p
          red[wave * 64 + 16 * jb + 4 * g + i] = gb2a[jb][i] * unscale;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      float t = red[threadIdx.x];
#pragma unroll
      for (int w = 1; w < T_WAVES; ++w) t += red[w * 64 + threadIdx.x];
      gb2_slabs[(int64_t)blockIdx.x * 64 + threadIdx.x] = w3v[threadIdx.x] * t;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Inference: logits only.  Same first product (p1_logit) as the training kernel => bit-identical logits.
// 16-edge tiles, 16 waves per CU (the kernel holds no gradient accumulators), rows of the next tile prefetched.
// ------------------------------------------------------------------------------------------------------------------
constexpr int I_WAVES = 16;
template <bool EXTRA>
__device__ __forceinline__ HalfIn load_tile16(const D16Params& a, int64_t tile, int64_t n_tiles, int c) {
  HalfIn h;
  const int64_t tc = tile < n_tiles ? tile : n_tiles - 1;
  const int64_t e_tile = tc * 16;
  const int64_t rest = a.E - 1 - e_tile;
  const int lim = (int)(rest < 15 ? rest : 15);
  const int k = min(c, lim);
  const int64_t* ei = a.ei + e_tile;
  const int64_t s = ei[k], d = ei[a.ld + k];
  h.key = h.key_nxt = 0;
  h.poff = (uint32_t)s * a.ldp_b;
  h.qoff = (uint32_t)d * a.ldq_b;
  h.aux = 0.f;
  h.w_e = EXTRA ? (a.extra + e_tile)[k] : 0.f;
  return h;
}
template <bool PQ16, bool EXTRA>
__global__ __launch_bounds__(I_WAVES * 64) void decoder_infer16_kernel(D16Params a, float* __restrict__ logits,
                                                                      int64_t n_tiles) {
  // LDS: W2 hi | mid | lo at LDS_W2 (the W2' slot stays unused), vectors at LDS_VEC
  __shared__ __attribute__((aligned(16))) char lds[LDS_WAVE0];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  stage_weights16(a.w2, a.b2, a.w3, a.cvec, lds, I_WAVES * 64, 1, LDS_W2P, LDS_VEC);
  __syncthreads();
  const float* cvl = reinterpret_cast<const float*>(lds + LDS_VEC) + 128;
  const int wfrag0 = c * 128 + ((g ^ wsw(c)) << 4), wfrag1 = wfrag0 ^ 64;
  const float b3v = a.b3[0];
  constexpr bool has_extra = EXTRA;
  const int64_t stride = (int64_t)gridDim.x * I_WAVES;
  int64_t tile = (int64_t)blockIdx.x * I_WAVES + wave;
  HalfIn cur = load_tile16<EXTRA>(a, tile, n_tiles, c);
  HalfIn nxt = load_tile16<EXTRA>(a, tile + stride, n_tiles, c);
  HalfRowsT<PQ16> rows;
  issue_half_rows(a, cur, g, rows);
  for (; tile < n_tiles; tile += stride) {
    float h[2][8];
    h1_frags(rows, has_extra, cur.w_e, cvl, g, h);
    const HalfIn nn = load_tile16<EXTRA>(a, tile + 2 * stride, n_tiles, c);
    issue_half_rows(a, nxt, g, rows);                       // next tile's rows fly during the product
    f32x4 acc[4];
    const float xv = p1_logit<true>(lds, h, wfrag0, wfrag1, g, b3v, acc);
    // no predicate (no branch in the loop body): a lane past the end looks at the last edge again (clamped ids) and
    // rewrites its bit-identical logit; the four lane groups store the same value
    const int64_t e = tile * 16 + c;
    logits[e < a.E ? e : a.E - 1] = xv;
    cur = nxt;
    nxt = nn;
  }
}

}  // namespace pangnn

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
namespace pangnn {
// launchers of the PQ16 instances (2-byte tables): launch_train16_tables_bf16 / launch_infer16_tables_bf16 in decoder16.o,
// ..._f16 in decoder16_f16.o.  `a` carries the row strides in bytes; y / part_buf / extra select the instance.
int D16_TABLES16(launch_train16_tables)(const D16Params& a, const float* g_logits, const D16Loss& lp, float* logits,
                                        const D16Run& rs, uint32_t* rec, float* ws, int64_t n_tiles, unsigned grid,
                                        hipStream_t s) {
  const dim3 gd(grid), bd(S_WAVES * 64);
#define PG_S16(F, R)                                                                                                  \
  do {                                                                                                                \
    if (a.extra)                                                                                                      \
      hipLaunchKernelGGL((decoder_train16_kernel<F, R, true, true>), gd, bd, 0, s, a, g_logits, lp, logits, rs, rec, ws, n_tiles);   \
    else                                                                                                              \
      hipLaunchKernelGGL((decoder_train16_kernel<F, R, true, false>), gd, bd, 0, s, a, g_logits, lp, logits, rs, rec, ws, n_tiles);  \
  } while (0)
  if (lp.y && rs.part) PG_S16(true, true);
  else if (lp.y) PG_S16(true, false);
  else if (rs.part) PG_S16(false, true);
  else PG_S16(false, false);
#undef PG_S16
  return 0;
}
int D16_TABLES16(launch_infer16_tables)(const D16Params& a, float* logits, int64_t n_tiles, unsigned grid, hipStream_t s) {
  if (a.extra)
    hipLaunchKernelGGL((decoder_infer16_kernel<true, true>), dim3(grid), dim3(I_WAVES * 64), 0, s, a, logits, n_tiles);
  else
    hipLaunchKernelGGL((decoder_infer16_kernel<true, false>), dim3(grid), dim3(I_WAVES * 64), 0, s, a, logits, n_tiles);
  return 0;
}
}  // namespace pangnn

#ifndef PANGNN_D16_TABLES_F16        // ---- everything below: decoder16.o only
namespace pangnn {
int launch_train16_tables_f16(const D16Params& a, const float* g_logits, const D16Loss& lp, float* logits, const D16Run& rs,
                              uint32_t* rec, float* ws, int64_t n_tiles, unsigned grid, hipStream_t s);      // decoder16_f16.o
int launch_infer16_tables_f16(const D16Params& a, float* logits, int64_t n_tiles, unsigned grid, hipStream_t s);
// decoder.hip: sums the per-workgroup slabs (layout SLAB16) in index order
int launch_decoder_reduce(const float* slabs, int n_slabs, float* g_w2, float* g_b2, float* g_w3, float* g_cvec,
                          float* g_b3, float* loss, hipStream_t s);

__global__ __launch_bounds__(kSumThreads) void gcv_reduce_kernel(const float* __restrict__ slabs, int n_slabs,
                                                             float* __restrict__ g_cvec) {
  const int i = threadIdx.x & (kWave - 1);
  const float s = ordered_parts_sum(slabs, n_slabs, 64, i, 64);
  if (threadIdx.x < kWave) g_cvec[i] = s;
}

static int cu_count() {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  return cus;
}
}  // namespace pangnn

namespace pangnn {
// decoder.hip (pangnn_decoder_mlp_infer_f32, precision = 1) -> the 16-edge-tile inference kernel of this file
static int launch_infer16_any(const void* p, int64_t ldp, const void* q, int64_t ldq, int pq_fmt, const int64_t* edge_index,
                              int64_t ld, int64_t num_edges, const float* extra, const float* cvec, const float* w2,
                              const float* b2, const float* w3, const float* b3, float* logits, hipStream_t s) {
  const int64_t n_tiles = (num_edges + 15) / 16;
  int64_t grid = (n_tiles + I_WAVES - 1) / I_WAVES;
  const int cus = cu_count();
  if (grid > cus) grid = cus;
  const uint32_t esz = pq_fmt ? 2u : 4u;                 // pq_fmt: PANGNN_DTYPE_F32 / _BF16 / _F16
  D16Params a{p, q, (uint32_t)ldp * esz, (uint32_t)ldq * esz, edge_index, ld, num_edges, extra, cvec, w2, b2, w3, b3,
              nullptr};
#define PG_I(H, X) hipLaunchKernelGGL((decoder_infer16_kernel<H, X>), dim3((unsigned)grid), dim3(I_WAVES * 64), 0, s, a, logits, n_tiles)
  if (pq_fmt == PANGNN_DTYPE_F16) launch_infer16_tables_f16(a, logits, n_tiles, (unsigned)grid, s);
  else if (pq_fmt) launch_infer16_tables_bf16(a, logits, n_tiles, (unsigned)grid, s);
  else { if (extra) PG_I(false, true); else PG_I(false, false); }
#undef PG_I
  PG_CHECK_LAUNCH("pangnn_decoder_mlp_infer");
  return 0;
}
int launch_decoder_infer16(const float* p, int64_t ldp, const float* q, int64_t ldq, const int64_t* edge_index, int64_t ld,
                           int64_t num_edges, const float* extra, const float* cvec, const float* w2, const float* b2,
                           const float* w3, const float* b3, float* logits, hipStream_t s) {
  return launch_infer16_any(p, ldp, q, ldq, PANGNN_DTYPE_F32, edge_index, ld, num_edges, extra, cvec, w2, b2, w3, b3, logits, s);
}
}  // namespace pangnn

using namespace pangnn;

#ifdef PANGNN_D16_STAMP
extern "C" int pangnn_debug_set_stamps(unsigned long long* ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(d16_stamps), &ptr, sizeof(ptr));
}
#endif
#ifdef PANGNN_D16_CYC
extern "C" int pangnn_debug_set_cyc(unsigned long long* ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(d16_cyc), &ptr, sizeof(ptr));
}
#endif
#ifdef PANGNN_D16_DEBUG
extern "C" int pangnn_debug_set_v(float* ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(d16_dbg_v), &ptr, sizeof(ptr));
}
#endif

extern "C" int pangnn_decoder_chunk_tiles(void) { return 1 << D16_CHUNK_LOG_MAX; }
extern "C" int pangnn_decoder_chunk_tiles_for(int64_t num_edges) {
  return 1 << chunk_log_for(num_edges <= 0 ? 0 : (num_edges + 31) / 32);
}

extern "C" size_t pangnn_decoder_train_workspace_bytes(void) { return (size_t)cu_count() * SLAB16 * sizeof(float); }

extern "C" int pangnn_decoder_mlp_infer_mixed(const void* p, int64_t ldp, const void* q, int64_t ldq, int32_t pq_dtype,
                                             int64_t num_nodes, const int64_t* edge_index, int64_t ld, int64_t num_edges,
                                             const float* extra, const float* cvec, const float* w2, const float* b2,
                                             const float* w3, const float* b3, int32_t D, float* logits,
                                             pangnn_stream_t stream) {
  const char* who = "pangnn_decoder_mlp_infer_mixed";
  const bool pq16 = pq_dtype == PANGNN_DTYPE_BF16 || pq_dtype == PANGNN_DTYPE_F16;          // a 2-byte table format
  PG_CHECK_ARG(pq16 || pq_dtype == PANGNN_DTYPE_F32, PANGNN_E_BADARG, "%s: pq_dtype is PANGNN_DTYPE_F32 / _BF16 / _F16", who);
  PG_CHECK_ARG(D == D16, PANGNN_E_BADARG, "%s: built for node_dim 64, got %d", who, (int)D);
  PG_CHECK_ARG(num_edges >= 0 && ld >= num_edges && num_nodes >= 0, PANGNN_E_BADARG, "%s: bad size", who);
  PG_CHECK_ARG(ldp >= D16 && ldq >= D16 && ldp % 8 == 0 && ldq % 8 == 0, PANGNN_E_BADARG,
               "%s: ldp / ldq must be multiples of 8 and >= 64", who);
  PG_CHECK_ARG((double)num_nodes * (double)(ldp > ldq ? ldp : ldq) * (pq16 ? 2.0 : 4.0) < 4294967296.0, PANGNN_E_TOOLARGE,
               "%s: node tables must stay under 4 GiB (32-bit gather offsets)", who);
  if (num_edges == 0) return 0;
  PG_CHECK_ARG(p && q && edge_index && w2 && b2 && w3 && b3 && logits && (!extra || cvec), PANGNN_E_BADARG,
               "%s: null pointer", who);
  PG_CHECK_ARG(aligned16(p) && aligned16(q) && aligned16(w2), PANGNN_E_ALIGN, "%s: p / q / w2 must be 16-byte aligned", who);
  return launch_infer16_any(p, ldp, q, ldq, pq_dtype, edge_index, ld, num_edges, extra, cvec, w2, b2, w3, b3, logits,
                            (hipStream_t)stream);
}

extern "C" int pangnn_decoder_train_mixed(const void* p, int64_t ldp, const void* q, int64_t ldq, int32_t pq_dtype,
                                          int64_t num_nodes, const int64_t* edge_index, int64_t ld, int64_t num_edges,
                                          const float* extra, const float* cvec, const float* w2, const float* b2,
                                          const float* w3, const float* b3, int32_t D, const float* y,
                                          const float* pos_weight, int64_t denom, const float* g_logits, float* logits,
                                          float* loss, uint32_t* rec, float* part_buf, const int32_t* part_off,
                                          float* g_w2, float* g_w3, float* g_b3, float* g_cvec,
                                          const int64_t* live_edges, void* workspace, size_t workspace_bytes,
                                          pangnn_stream_t stream) {
  const char* who = "pangnn_decoder_train";
  const bool pq16 = pq_dtype == PANGNN_DTYPE_BF16 || pq_dtype == PANGNN_DTYPE_F16;          // a 2-byte table format
  PG_CHECK_ARG(pq16 || pq_dtype == PANGNN_DTYPE_F32, PANGNN_E_BADARG, "%s: pq_dtype is PANGNN_DTYPE_F32 / _BF16 / _F16", who);
  PG_CHECK_ARG(D == D16, PANGNN_E_BADARG, "%s: built for node_dim 64, got %d", who, (int)D);
  PG_CHECK_ARG(num_edges >= 0 && ld >= num_edges && num_nodes >= 0, PANGNN_E_BADARG, "%s: bad size", who);
  // the records this pass writes are read back by pangnn_decoder_dgrad_f32 through int32 edge ids
  PG_CHECK_ARG(num_edges < 2147483647LL, PANGNN_E_TOOLARGE, "%s: num_edges must fit int32 (partition the graph first)", who);
  PG_CHECK_ARG(ldp >= D16 && ldq >= D16 && ldp % (pq16 ? 8 : 4) == 0 && ldq % (pq16 ? 8 : 4) == 0, PANGNN_E_BADARG,
               "%s: ldp / ldq must be multiples of 4 (f32) / 8 (bf16, f16) and >= 64", who);
  PG_CHECK_ARG((double)num_nodes * (double)(ldp > ldq ? ldp : ldq) * (pq16 ? 2.0 : 4.0) < 4294967296.0, PANGNN_E_TOOLARGE,
               "%s: node tables must stay under 4 GiB (32-bit gather offsets)", who);
  PG_CHECK_ARG(g_w2 && g_w3 && g_b3, PANGNN_E_BADARG, "%s: null gradient output", who);
  PG_CHECK_ARG((y != nullptr) != (g_logits != nullptr) || num_edges == 0, PANGNN_E_BADARG,
               "%s: exactly one of y (fused loss) and g_logits (given gradient) must be set", who);
  PG_CHECK_ARG(!y || (denom > 0 && loss && logits), PANGNN_E_BADARG, "%s: fused loss needs denom > 0, loss, logits", who);
  PG_CHECK_ARG((part_buf == nullptr) == (part_off == nullptr), PANGNN_E_BADARG, "%s: part_buf and part_off go together",
               who);
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_tiles = (num_edges + 31) / 32;
  const int clog = chunk_log_for(n_tiles);                             // = pangnn_decoder_chunk_tiles_for(num_edges)
  const int64_t n_chunks = (n_tiles + (1 << clog) - 1) >> clog;        // a wave takes whole chunks of tiles
  int64_t grid = (n_chunks + S_WAVES - 1) / S_WAVES;
  const int cus = cu_count();
  if (grid > cus) grid = cus;
  if (grid < 1) grid = 1;
  PG_CHECK_ARG(workspace && workspace_bytes >= (size_t)grid * SLAB16 * sizeof(float), PANGNN_E_WORKSPACE,
               "%s: workspace too small", who);
  float* ws = static_cast<float*>(workspace);
  if (num_edges == 0) {
    hipError_t e = hipMemsetAsync(workspace, 0, (size_t)grid * SLAB16 * sizeof(float), s);
    PG_CHECK_ARG(e == hipSuccess, (int)e, "%s: memset failed", who);
  } else {
    PG_CHECK_ARG(p && q && edge_index && w2 && b2 && w3 && b3 && (!extra || cvec), PANGNN_E_BADARG, "%s: null pointer",
                 who);
    PG_CHECK_ARG(aligned16(p) && aligned16(q) && aligned16(w2), PANGNN_E_ALIGN, "%s: p / q / w2 must be 16-byte aligned",
                 who);
    PG_CHECK_ARG(rec && aligned16(rec), PANGNN_E_ALIGN, "%s: rec is required and must be 16-byte aligned", who);
    const uint32_t esz = pq16 ? 2u : 4u;
    D16Params a{p, q, (uint32_t)ldp * esz, (uint32_t)ldq * esz, edge_index, ld, num_edges, extra, cvec, w2, b2, w3, b3,
                live_edges};
    const D16Loss lp{y, pos_weight, y ? 1.0f / (float)denom : 0.f};
    const D16Run rs{part_buf, part_off, clog};
    const dim3 gd((unsigned)grid), bd(S_WAVES * 64);
#define PG_S2(F, R, H)                                                                                              \
  do {                                                                                                              \
    if (extra)                                                                                                      \
      hipLaunchKernelGGL((decoder_train16_kernel<F, R, H, true>), gd, bd, 0, s, a, g_logits, lp, logits, rs, rec, ws, n_tiles);  \
    else                                                                                                            \
      hipLaunchKernelGGL((decoder_train16_kernel<F, R, H, false>), gd, bd, 0, s, a, g_logits, lp, logits, rs, rec, ws, n_tiles); \
  } while (0)
#define PG_S(F, R, H) PG_S2(F, R, H)
    if (pq_dtype == PANGNN_DTYPE_F16) {
      launch_train16_tables_f16(a, g_logits, lp, logits, rs, rec, ws, n_tiles, (unsigned)grid, s);
    } else if (pq16) {
      launch_train16_tables_bf16(a, g_logits, lp, logits, rs, rec, ws, n_tiles, (unsigned)grid, s);
    } else {
      if (y && part_buf) PG_S(true, true, false);
      else if (y) PG_S(true, false, false);
      else if (part_buf) PG_S(false, true, false);
      else PG_S(false, false, false);
    }
#undef PG_S2
#undef PG_S
    PG_CHECK_LAUNCH(who);
  }
  return launch_decoder_reduce(ws, (int)grid, g_w2, nullptr, g_w3, g_cvec, g_b3, loss, s);
}

extern "C" int pangnn_decoder_train_f32(const float* p, int64_t ldp, const float* q, int64_t ldq, int64_t num_nodes,
                                        const int64_t* edge_index, int64_t ld, int64_t num_edges, const float* extra,
                                        const float* cvec, const float* w2, const float* b2, const float* w3,
                                        const float* b3, int32_t D, const float* y, const float* pos_weight,
                                        int64_t denom, const float* g_logits, float* logits, float* loss,
                                        uint32_t* rec, float* part_buf, const int32_t* part_off, float* g_w2,
                                        float* g_w3, float* g_b3, float* g_cvec, const int64_t* live_edges,
                                        void* workspace, size_t workspace_bytes, pangnn_stream_t stream) {
  return pangnn_decoder_train_mixed(p, ldp, q, ldq, PANGNN_DTYPE_F32, num_nodes, edge_index, ld, num_edges, extra, cvec,
                                    w2, b2, w3, b3, D, y, pos_weight, denom, g_logits, logits, loss, rec, part_buf,
                                    part_off, g_w2, g_w3, g_b3, g_cvec, live_edges, workspace, workspace_bytes, stream);
}

extern "C" size_t pangnn_decoder_dgrad_workspace_bytes(void) { return (size_t)cu_count() * 64 * sizeof(float); }

extern "C" int pangnn_decoder_dgrad_f32(const uint32_t* rec, const int32_t* perm, const int32_t* keys, const float* w2,
                                        const float* w3, int64_t num_edges, float* part_buf, const int32_t* part_off,
                                        float* g_b2, const int64_t* live_edges, void* workspace, size_t workspace_bytes,
                                        pangnn_stream_t stream) {
  const char* who = "pangnn_decoder_dgrad_f32";
  PG_CHECK_ARG(num_edges >= 0, PANGNN_E_BADARG, "%s: bad size", who);
  // edge ids are int32 in perm and 32-bit in the kernel's record addressing ((e << 1) << 4 with a 32-bit e << 1)
  PG_CHECK_ARG(num_edges < 2147483647LL, PANGNN_E_TOOLARGE, "%s: num_edges must fit int32 (partition the graph first)", who);
  hipStream_t s = (hipStream_t)stream;
  if (num_edges == 0) {
    if (g_b2) {
      hipError_t e = hipMemsetAsync(g_b2, 0, 64 * sizeof(float), s);
      PG_CHECK_ARG(e == hipSuccess, (int)e, "%s: memset failed", who);
    }
    return 0;
  }
  PG_CHECK_ARG(rec && w2 && w3 && (part_buf == nullptr) == (part_off == nullptr) && (!part_buf || keys),
               PANGNN_E_BADARG, "%s: null pointer", who);
  PG_CHECK_ARG(part_buf || g_b2, PANGNN_E_BADARG, "%s: nothing to compute", who);
  const int64_t n_tiles = (num_edges + 31) / 32;
  const int clog = chunk_log_for(n_tiles);
  const int64_t n_chunks = (n_tiles + (1 << clog) - 1) >> clog;
  int64_t grid = (n_chunks + T_WAVES - 1) / T_WAVES;
  const int cus = cu_count();
  if (grid > cus) grid = cus;
  PG_CHECK_ARG(!g_b2 || (workspace && workspace_bytes >= (size_t)grid * 64 * sizeof(float)), PANGNN_E_WORKSPACE,
               "%s: workspace too small", who);
  const D16Run rs{part_buf, part_off, clog};
  float* b2_slabs = g_b2 ? static_cast<float*>(workspace) : nullptr;
#define PG_T(P, R)                                                                                                   \
  hipLaunchKernelGGL((decoder_dgrad16_kernel<P, R>), dim3((unsigned)grid), dim3(T_WAVES * 64), 0, s, rec, perm, keys, \
                     w2, w3, num_edges, live_edges, rs, b2_slabs, n_tiles)
  if (part_buf) {
    if (perm) PG_T(true, true); else PG_T(false, true);
  } else {
    PG_T(false, false);                   // the parameter sum alone: the order of the positions does not matter
  }
#undef PG_T
  PG_CHECK_LAUNCH(who);
  if (g_b2) {
    hipLaunchKernelGGL(gcv_reduce_kernel, dim3(1), dim3(kSumThreads), 0, s, b2_slabs, (int)grid, g_b2);
    PG_CHECK_LAUNCH(who);
  }
  return 0;
}
#endif  // !PANGNN_D16_TABLES_F16
