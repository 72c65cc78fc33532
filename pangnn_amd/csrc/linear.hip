// Node-level dense layers with a short inner dimension (K, M in {64, 128}) over N ~ 1e6 rows:
//   y[N,M]  = x[N,K] w[M,K]^T (+ bias)                      GCNConv.lin, decoder P|Q, dL/dx = g w
//   gw[M,K] = g[N,M]^T x[N,K],  gb[M] = sum_n g[n,:]        weight / bias gradients
// These are HBM-bound (0.25 flop/B at K = 64): a library GEMM tuned for square problems leaves an
// order of magnitude on the table (hipBLASLt: 1.7 - 3.3 ms for 0.77 GB at N = 1e6).  Here one wave
// owns 32 consecutive rows: row-contiguous 16-byte loads into a padded LDS tile, w resident in LDS,
// f32 MFMA 32x32x2 (exact fp32 FMA chains), output rows stored as 128-byte segments.
// The weight gradient keeps its [M,K] accumulator in registers across all of a wave's tiles and is
// reduced over waves and workgroups in a fixed order (slabs), so it is bitwise reproducible.
#include "common.h"

namespace pangnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef PANGNN_LIN_NOMFMA   // diagnostic builds only (tools/ablate_linear.sh): one VALU op per MFMA
#define LIN_MFMA(a, b, c) ([&] { f32x16 t_ = (c); t_[0] += (a) * (b); return t_; }())
#else
#define LIN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#endif

__device__ __forceinline__ constexpr int jrow(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// wave-uniform base pointer + 32-bit per-lane byte offset: the form that maps to `global_* v_off, v_data, s[base]`
// (scalar address arithmetic per tile, no 64-bit vector pointer per access).  The compiler only selects it when
// the base is known to sit in SGPRs and the offset's zero-extension is in the same basic block, hence `pin()`
// at the top of every block of accesses: an empty asm that ties `base` to an SGPR pair and `off` to a VGPR.
__device__ __forceinline__ void pin(int64_t& base, uint32_t& off) { asm volatile("" : "+s"(base), "+v"(off)); }
__device__ __forceinline__ float4 ld_f4(const float* ubase, uint32_t byte_off) {
#ifndef PANGNN_LIN_NO_NT      // rows are read once per kernel: streaming loads (3-8 % in isolation, tools/ablate_linear.sh)
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(ubase) + byte_off));
  return make_float4(t[0], t[1], t[2], t[3]);
#else
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(ubase) + byte_off);
#endif
}
__device__ __forceinline__ void st_f32(float* ubase, uint32_t byte_off, float v) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(ubase) + byte_off) = v;
}

// ---- storage types.  Arithmetic is fp32 everywhere (f32 MFMA); a matrix may be STORED as bfloat16 (config 5's
// autocast: the reference's Linear outputs are bf16 tensors, gnn.py:93,111,173 under accelerate's mixed precision):
// half the HBM bytes of the row streams these kernels are bound by.  bf16 -> f32 is a shift, f32 -> bf16 rounds to
// nearest even.  Elem<T>: 4 consecutive elements as raw registers (converted only when they go to LDS, so the
// prefetch distance of the f32 path is kept), one element in / out for the strided epilogue accesses.
typedef unsigned short bf16s;
template <typename T> struct Elem;
template <> struct Elem<float> {
  typedef float4 raw4;
  static __device__ __forceinline__ raw4 ld4(const float* ubase, uint32_t byte_off) { return ld_f4(ubase, byte_off); }
  static __device__ __forceinline__ raw4 ld4_plain(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ raw4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ float4 cvt4(raw4 r) { return r; }
  static __device__ __forceinline__ float ld1(const float* ubase, uint32_t byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ubase) + byte_off);
  }
  static __device__ __forceinline__ void st1(float* ubase, uint32_t byte_off, float v) { st_f32(ubase, byte_off, v); }
};
template <> struct Elem<bf16s> {
  typedef uint2 raw4;
  static __device__ __forceinline__ raw4 ld4(const bf16s* ubase, uint32_t byte_off) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(reinterpret_cast<const char*>(ubase) + byte_off));
    return make_uint2(t[0], t[1]);
  }
  static __device__ __forceinline__ raw4 ld4_plain(const bf16s* p) { return *reinterpret_cast<const uint2*>(p); }
  static __device__ __forceinline__ raw4 zero4() { return make_uint2(0u, 0u); }
  static __device__ __forceinline__ float4 cvt4(raw4 r) {
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                       __uint_as_float(r.y & 0xffff0000u));
  }
  static __device__ __forceinline__ float ld1(const bf16s* ubase, uint32_t byte_off) {
    return __uint_as_float((uint32_t)*reinterpret_cast<const bf16s*>(reinterpret_cast<const char*>(ubase) + byte_off) << 16);
  }
  static __device__ __forceinline__ void st1(bf16s* ubase, uint32_t byte_off, float v) {
    *reinterpret_cast<__bf16*>(reinterpret_cast<char*>(ubase) + byte_off) = (__bf16)v;     // round to nearest even
  }
};

// IEEE half rows (`--mixed_precision fp16`, src/setup.py:50: the autocast Linear outputs are float16 tensors): the same
// 2-byte row streams, f16 -> f32 exact (v_cvt_f32_f16), f32 -> f16 rounds to nearest even (v_cvt_f16_f32)
struct f16s { unsigned short bits; };
template <> struct Elem<f16s> {
  typedef uint2 raw4;
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ raw4 ld4(const f16s* ubase, uint32_t byte_off) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(reinterpret_cast<const char*>(ubase) + byte_off));
    return make_uint2(t[0], t[1]);
  }
  static __device__ __forceinline__ raw4 ld4_plain(const f16s* p) { return *reinterpret_cast<const uint2*>(p); }
  static __device__ __forceinline__ raw4 zero4() { return make_uint2(0u, 0u); }
  static __device__ __forceinline__ float4 cvt4(raw4 r) {
    const h2 a = __builtin_bit_cast(h2, r.x), b = __builtin_bit_cast(h2, r.y);
    return make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
  }
  static __device__ __forceinline__ float ld1(const f16s* ubase, uint32_t byte_off) {
    return (float)*reinterpret_cast<const _Float16*>(reinterpret_cast<const char*>(ubase) + byte_off);
  }
  static __device__ __forceinline__ void st1(f16s* ubase, uint32_t byte_off, float v) {
    // the value is rounded to f32 FIRST (as every other storage type sees it), then to nearest-even f16: without the barrier the
    // compiler folds a preceding multiply into v_fma_mixlo_f16 — one rounding of the exact product, one ulp away from the f32
    // kernel's result in about one element of 10^4
    asm volatile("" : "+v"(v));
    *reinterpret_cast<_Float16*>(reinterpret_cast<char*>(ubase) + byte_off) = (_Float16)v;
  }
};

// rows [base, base+32) of a [n, C] matrix (leading dim ld): issued into registers one tile ahead
// (`load_rows`), written to the LDS tile [32][C+4] when the previous tile's MFMAs are done
// (`store_rows`); rows >= n are zero.
template <int C, typename T = float>
struct RowRegs { typename Elem<T>::raw4 v[32 / (64 / (C / 4))]; };

template <int C, typename T>
__device__ __forceinline__ void load_rows_full(const T* __restrict__ src, int64_t ld, int64_t base, int lane,
                                               RowRegs<C, T>& rg) {   // whole tile inside the matrix: no predicate
  constexpr int LPR = C / 4;            // lanes per row
  constexpr int RPI = 64 / LPR;         // rows per wave-instruction
  const int c4 = lane % LPR, r0 = lane / LPR;
  uint32_t loff = ((uint32_t)r0 * (uint32_t)ld + 4u * c4) * (uint32_t)sizeof(T);
  pin(base, loff);
#pragma unroll
  for (int i = 0; i < 32 / RPI; ++i) rg.v[i] = Elem<T>::ld4(src + (base + i * RPI) * ld, loff);
}

template <int C, typename T>
__device__ __forceinline__ void load_rows(const T* __restrict__ src, int64_t ld, int64_t n, int64_t base,
                                          int lane, RowRegs<C, T>& rg) {
  constexpr int LPR = C / 4;
  constexpr int RPI = 64 / LPR;
  const int c4 = lane % LPR, r0 = lane / LPR;
  if (base + 32 <= n) {                 // wave-uniform
    load_rows_full<C, T>(src, ld, base, lane, rg);
    return;
  }
#pragma unroll
  for (int i = 0; i < 32 / RPI; ++i) {
    const int row = i * RPI + r0;
    rg.v[i] = Elem<T>::zero4();
    if (base + row < n) rg.v[i] = Elem<T>::ld4_plain(src + (base + row) * ld + 4 * c4);
  }
}

// ELU (alpha = 1) and its derivative, both as functions of the PRE-activation value
__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v : __expf(v) - 1.f; }   // v_exp_f32: 2 ulp, abs 1e-7
__device__ __forceinline__ float elu1_grad(float v) { return v > 0.f ? 1.f : __expf(v); }
// one value of the first layer by linearity, h = r a + s c + b_in (embed_conv_in_rows_kernel's two fused multiply-adds)
__device__ __forceinline__ float gen_h(float rv, float sv, float a, float c, float b) { return fmaf(rv, a, fmaf(sv, c, b)); }

// ACT = 1: the rows go through ELU on their way to LDS, i.e. the kernel multiplies elu(x) without elu(x) ever
// existing in HBM (gnn.py:131-166: every activation of the encoder is followed by exactly one dense layer)
template <int C, int ACT = 0, typename T = float>
__device__ __forceinline__ void store_rows(const RowRegs<C, T>& rg, int lane, float* tile) {
  constexpr int LPR = C / 4;
  constexpr int RPI = 64 / LPR;
  constexpr int RS = C + 4;
  const int c4 = lane % LPR, r0 = lane / LPR;
#pragma unroll
  for (int i = 0; i < 32 / RPI; ++i) {
    float4 v = Elem<T>::cvt4(rg.v[i]);
    if (ACT == 1) { v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w); }
    *reinterpret_cast<float4*>(tile + (i * RPI + r0) * RS + 4 * c4) = v;
  }
}

// Waves per workgroup of the forward kernel.  Measured at N = 1e6 (tools/ablate_linear.sh): 4 waves per CU
// (one per SIMD) is the fastest arrangement — an f32 MFMA occupies its SIMD's vector lanes, so a second wave
// on the SIMD cannot hide the first one's epilogue, it only adds LDS / issue contention (8 waves +5..10 %,
// 12 waves +30 %).
template <int K, int M>
struct FwdGeo {
  static constexpr int KS = K + 4;
#ifdef PANGNN_LIN_WAVES      // diagnostic builds only
  static constexpr int WAVES = PANGNN_LIN_WAVES;
#elif defined(PANGNN_LIN_F32_MFMA) || defined(PANGNN_LIN_128_W4)
  static constexpr int WAVES = 4;
#else
  // (8 waves for K = 64 measured: no gain — these kernels are at their HBM time.)  128 x 128: the three split images of the
  // weight take 102 KB, which leaves room for THREE 16.5 KB row tiles — three waves on the split-bf16 product beat four on
  // v_mfma_f32_32x32x2_f32 (round 5, N = 1e6: forward 0.23-0.25 ms vs 0.30, dL/dx + dL/dW 0.65 vs 0.71; -DPANGNN_LIN_128_W4)
  static constexpr int WAVES = (K == 128 && M == 128) ? 3 : 4;
#endif
};

// one 32-row tile out of LDS: acc[b] = x_tile . w[32b .. 32b+32)^T.  Operand fragments of k-step i+1 are read
// from LDS before the 4*M/32 MFMAs of step i are issued (sched_barrier keeps that order).
template <int K, int M>
__device__ __forceinline__ void tile_product(const float* Xt, const float* Wl, int r, int hh, f32x16 (&acc)[M / 32]) {
  constexpr int KS = K + 4;
#pragma unroll
  for (int b = 0; b < M / 32; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
  const float* xa = Xt + r * KS + 4 * (K / 8) * hh;
  const float* wb = Wl + r * KS + 4 * (K / 8) * hh;
  float4 a = *reinterpret_cast<const float4*>(xa);
  float4 bw[M / 32];
#pragma unroll
  for (int b = 0; b < M / 32; ++b) bw[b] = *reinterpret_cast<const float4*>(wb + 32 * b * KS);
#pragma unroll
  for (int i = 0; i < K / 8; ++i) {
    float4 an = a, bn[M / 32];
#pragma unroll
    for (int b = 0; b < M / 32; ++b) bn[b] = bw[b];
    if (i + 1 < K / 8) {
      an = *reinterpret_cast<const float4*>(xa + 4 * (i + 1));
#pragma unroll
      for (int b = 0; b < M / 32; ++b) bn[b] = *reinterpret_cast<const float4*>(wb + 32 * b * KS + 4 * (i + 1));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < M / 32; ++b) {
      acc[b] = LIN_MFMA(a.x, bw[b].x, acc[b]);
      acc[b] = LIN_MFMA(a.y, bw[b].y, acc[b]);
      acc[b] = LIN_MFMA(a.z, bw[b].z, acc[b]);
      acc[b] = LIN_MFMA(a.w, bw[b].w, acc[b]);
    }
    __builtin_amdgcn_sched_barrier(0);
    a = an;
#pragma unroll
    for (int b = 0; b < M / 32; ++b) bw[b] = bn[b];
  }
}

// ---- the same product on the bf16 matrix pipe with fp32-exact operand handling (default; -DPANGNN_LIN_F32_MFMA: the
// f32-MFMA product above).  v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate and holds its SIMD for 16 passes: at
// N = 1e6 a 128 x 64 layer is 1.6e10 flop = 0.10 ms of matrix time at peak, 0.20 ms measured — above the layers' HBM
// time once their row streams are short (the generated first layer reads 8 bytes per row).  Here both operands are split
// into three bf16 terms by truncation (x = hi + mid + lo EXACTLY: 8 + 8 + 8 significand bits) and the six partial
// products of order <= 2^-16 are accumulated in fp32 by v_mfma_f32_32x32x16_bf16, smallest first; the three dropped
// terms (mid.lo, lo.mid, lo.lo) are <= 2^-23 |x||w| — below fp32's own rounding of the product.  Weights are split
// once per workgroup into three LDS images [M][K+8] bf16; a tile's x values are split as they are read (5.5 VALU per
// value, under the matrix instructions).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
struct Split3 { bf16x8 hi, mid, lo; };
__device__ __forceinline__ Split3 split8(const float4& f0, const float4& f1) {
  const float f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
  u32x4v hi, mid, lo;
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

#ifdef PANGNN_LIN_F32_MFMA
constexpr bool kLinX3 = false;
#else
constexpr bool kLinX3 = true;
#endif

// floats of LDS the weight image of a [M][K] layer takes
template <int K, int M>
struct WImg {
  static constexpr int WS = K + 8;                                     // bf16 row stride of a split image
  // 128 x 128: three 34 KB images + three 16.5 KB row tiles (FwdGeo: three waves) = 152 KB of the 160 KB; with four tiles it
  // does not fit and the f32 product is used
  static constexpr bool X3 = kLinX3 && (3 * M * WS * 2 + FwdGeo<K, M>::WAVES * 32 * (K + 4) * 4 <= 156 * 1024);
  static constexpr int FLOATS = X3 ? (3 * M * WS + 1) / 2 : M * (K + 4);
};

// stage w into LDS: element (m, k) of the [M][K] operand is w[m * sm + k * sk] (sk = 1: row-major [M][K]; sm = 1: the
// transpose of a row-major [K][M])
template <int K, int M>
__device__ __forceinline__ void stage_w(const float* __restrict__ w, int sm, int sk, float* Wl, int n_threads) {
  if constexpr (WImg<K, M>::X3) {
    constexpr int WS = WImg<K, M>::WS, IMG = M * WS;
    unsigned short* W3 = reinterpret_cast<unsigned short*>(Wl);
    // eight weight loads in flight per thread: on a mini-batch (a few workgroups, every one of them staging w before its
    // first tile) the loop of dependent load -> split -> store rounds was most of the kernel (13-16 us for 900 rows)
#pragma unroll 8
    for (int i = threadIdx.x; i < M * K; i += n_threads) {
      const int m = sk == 1 ? i / K : i % M, k = sk == 1 ? i % K : i / M;       // consecutive threads: consecutive addresses
      const float v = w[m * sm + k * sk];
      const uint32_t a = __builtin_bit_cast(uint32_t, v);
      const float r1 = v - __builtin_bit_cast(float, a & 0xffff0000u);
      const uint32_t b = __builtin_bit_cast(uint32_t, r1);
      const float r2 = r1 - __builtin_bit_cast(float, b & 0xffff0000u);
      W3[m * WS + k] = (unsigned short)(a >> 16);
      W3[IMG + m * WS + k] = (unsigned short)(b >> 16);
      W3[2 * IMG + m * WS + k] = (unsigned short)(__builtin_bit_cast(uint32_t, r2) >> 16);
    }
  } else {
    constexpr int KS = K + 4;
#pragma unroll 8
    for (int i = threadIdx.x; i < M * K; i += n_threads) {
      const int m = sk == 1 ? i / K : i % M, k = sk == 1 ? i % K : i / M;
      Wl[m * KS + k] = w[m * sm + k * sk];
    }
  }
}

template <int K, int M>
__device__ __forceinline__ void tile_product_x3(const float* Xt, const float* Wl, int r, int hh, f32x16 (&acc)[M / 32]) {
  constexpr int KS = K + 4, WS = WImg<K, M>::WS, IMG = M * WS;
#pragma unroll
  for (int b = 0; b < M / 32; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
  const float* xa = Xt + r * KS + 8 * hh;
  const unsigned short* wb = reinterpret_cast<const unsigned short*>(Wl) + r * WS + 8 * hh;
  float4 x0 = *reinterpret_cast<const float4*>(xa), x1 = *reinterpret_cast<const float4*>(xa + 4);
  // weight fragments of a k-step (hi, mid, lo per output block): double-buffered one step ahead where the registers
  // allow it (M = 64: 2 x 24), read at the top of their own step, ahead of the x split, for M = 128 (48)
  constexpr bool DB = (M / 32) <= 2;
  bf16x8 wf[DB ? 2 : 1][M / 32][3];
  if constexpr (DB) {
#pragma unroll
    for (int b = 0; b < M / 32; ++b)
#pragma unroll
      for (int t = 0; t < 3; ++t) wf[0][b][t] = *reinterpret_cast<const bf16x8*>(wb + 32 * b * WS + t * IMG);
  }
#pragma unroll
  for (int i = 0; i < K / 16; ++i) {
    if constexpr (!DB) {
#pragma unroll
      for (int b = 0; b < M / 32; ++b)
#pragma unroll
        for (int t = 0; t < 3; ++t) wf[0][b][t] = *reinterpret_cast<const bf16x8*>(wb + 32 * b * WS + 16 * i + t * IMG);
    }
    const Split3 a = split8(x0, x1);
    if (i + 1 < K / 16) {                  // operands of step i + 1 are read from LDS before step i's matrix instructions
      x0 = *reinterpret_cast<const float4*>(xa + 16 * (i + 1));
      x1 = *reinterpret_cast<const float4*>(xa + 16 * (i + 1) + 4);
      if constexpr (DB) {
#pragma unroll
        for (int b = 0; b < M / 32; ++b)
#pragma unroll
          for (int t = 0; t < 3; ++t)
            wf[(i + 1) & 1][b][t] = *reinterpret_cast<const bf16x8*>(wb + 32 * b * WS + 16 * (i + 1) + t * IMG);
      }
    }
    // the six partial products, smallest first; the output blocks interleaved so that consecutive matrix instructions
    // never wait for each other's accumulator
#define PG_X3(AT, WT)                                                                                     \
    _Pragma("unroll") for (int b = 0; b < M / 32; ++b)                                                    \
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.AT, wf[DB ? (i & 1) : 0][b][WT], acc[b], 0, 0, 0);
    PG_X3(lo, 0) PG_X3(hi, 2) PG_X3(mid, 1) PG_X3(mid, 0) PG_X3(hi, 1) PG_X3(hi, 0)
#undef PG_X3
  }
}

template <int K, int M>
__device__ __forceinline__ void tile_mult(const float* Xt, const float* Wl, int r, int hh, f32x16 (&acc)[M / 32]) {
  if constexpr (WImg<K, M>::X3) tile_product_x3<K, M>(Xt, Wl, r, hh, acc);
  else tile_product<K, M>(Xt, Wl, r, hh, acc);
}

// Pipeline per wave over the FULL tiles (rows [32t, 32t+32) all inside the matrix):
//   registers hold the x rows of tile t+1 (global loads issued one tile ahead) while tile t is multiplied out
//   of LDS; they are committed to LDS at the END of iteration t, in the same basic block as tile t's 64 output
//   stores.  The loads are older than those stores, so the wait the compiler puts before the LDS write is
//   vmcnt(63): loads landed, stores still in flight.  Committing at the loop top instead costs vmcnt(0) — every
//   tile then waits until its predecessor's writes are acknowledged (0.30 ms vs 0.19 ms per launch at N = 1e6).
//   The loop body has no branch: the prefetch index is clamped instead of guarded.
// The one partial tile at the end of the matrix is handled after the loop by the wave whose turn it is.
// ACT: input activation (0 none, 1 ELU).  GATE: the result is multiplied by elu'(gate[row][col]) — the backward of
// "dense layer after ELU" w.r.t. the pre-activation, produced directly by the dL/dx product.
// TX / TY / TG: storage types of x, y and gate (float or bf16s).
template <int K, int M, int ACT, bool GATE, typename TX = float, typename TY = float, typename TG = float>
__global__ __launch_bounds__((FwdGeo<K, M>::WAVES * 64)) void linear_fwd_kernel(const TX* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ w, int w_sm, int w_sk,
                                                         const float* __restrict__ bias,
                                                         TY* __restrict__ y, int64_t ldy, int64_t n,
                                                         int64_t n_tiles, const TG* __restrict__ gate,
                                                         int64_t ldgate) {
  constexpr int KS = K + 4;
  constexpr int WAVES = FwdGeo<K, M>::WAVES;
  constexpr int WF = (WImg<K, M>::FLOATS + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float lds[WF + WAVES * 32 * KS];
  float* Wl = lds;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // SGPR
  float* Xt = lds + WF + wave * (32 * KS);
  stage_w<K, M>(w, w_sm, w_sk, Wl, WAVES * 64);   // (K, 1): w is [M][K]; (1, M): w is the [K][M] matrix whose transpose is meant
  __syncthreads();
  const int r = lane & 31, hh = lane >> 5;
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  const int64_t n_full = n / 32;
  int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
  float bv[M / 32];
#pragma unroll
  for (int b = 0; b < M / 32; ++b) bv[b] = bias ? bias[r + 32 * b] : 0.f;
  f32x16 acc[M / 32];
  RowRegs<K, TX> rg;
  if (tile < n_full) {
    const int64_t last = n_full - 1;
    load_rows_full<K, TX>(x, ldx, tile * 32, lane, rg);
    store_rows<K, ACT, TX>(rg, lane, Xt);
    load_rows_full<K, TX>(x, ldx, (tile + stride < last ? tile + stride : last) * 32, lane, rg);
    wave_sync_lds();
    for (; tile < n_full; tile += stride) {
      // GATE: the tile's gate values are fetched before the product so that they land behind the MFMAs (fetched
      // in the epilogue, one load-and-use at a time, the 128-wide variant took 0.98 ms instead of 0.17)
      float gv[GATE ? M / 32 : 1][16];
      if (GATE) {
        uint32_t goff = (4u * hh * (uint32_t)ldgate + r) * (uint32_t)sizeof(TG);
        int64_t gbase = tile * 32;
        pin(gbase, goff);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const TG* gr = gate + (gbase + (i & 3) + 8 * (i >> 2)) * ldgate;
#pragma unroll
          for (int b = 0; b < M / 32; ++b) gv[b][i] = Elem<TG>::ld1(gr + 32 * b, goff);
        }
      }
      tile_mult<K, M>(Xt, Wl, r, hh, acc);
      // C[row = jrow(i,hh)][m = r + 32b]: 128-byte row segments
      uint32_t loff = (4u * hh * (uint32_t)ldy + r) * (uint32_t)sizeof(TY);
      int64_t sbase = tile * 32;
      pin(sbase, loff);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        TY* yr = y + (sbase + (i & 3) + 8 * (i >> 2)) * ldy;              // scalar
#pragma unroll
        for (int b = 0; b < M / 32; ++b) {
          float v = acc[b][i] + bv[b];
          if (GATE) v *= elu1_grad(gv[b][i]);
          Elem<TY>::st1(yr + 32 * b, loff, v);
        }
      }
      store_rows<K, ACT, TX>(rg, lane, Xt);           // tile + stride (see the pipeline note)
      const int64_t nxt = tile + 2 * stride;
      load_rows_full<K, TX>(x, ldx, (nxt < last ? nxt : last) * 32, lane, rg);
      wave_sync_lds();
    }
  }
  if (tile == n_full && n_full < n_tiles) {            // the partial tile, rows [32 n_full, n)
    const int64_t base = n_full * 32;
    load_rows<K, TX>(x, ldx, n, base, lane, rg);
    wave_sync_lds();
    store_rows<K, ACT, TX>(rg, lane, Xt);
    wave_sync_lds();
    tile_mult<K, M>(Xt, Wl, r, hh, acc);
#pragma unroll
    for (int b = 0; b < M / 32; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t row = base + jrow(i, hh);
        if (row < n) {
          float v = acc[b][i] + bv[b];
          if (GATE) v *= elu1_grad(Elem<TG>::ld1(gate + row * ldgate + r + 32 * b, 0u));
          Elem<TY>::st1(y + row * ldy + r + 32 * b, 0u, v);
        }
      }
  }
}

// One 32-row tile of the weight gradient: acc[a][b] (32 x 32 block of gw[M][K]) += G_tile^T X_tile, gbp[a] += column sums
// of G.  f32 product: 16 steps of 2 rows on v_mfma_f32_32x32x2_f32.  Split-bf16 product (default): 2 steps of 16 rows on
// v_mfma_f32_32x32x16_bf16 — both operands are data here, so both are split (three exact bf16 terms each, six partial
// products, smallest first): per tile 96 column reads, 12 splits, 96 matrix instructions of 8 passes instead of 128 of 16.
template <int K, int M>
__device__ __forceinline__ void wgrad_tile(const float* Xt, const float* Gt, int r, int hh, f32x16 (&acc)[M / 32][K / 32],
                                           float (&gbp)[M / 32]) {
  constexpr int KS = K + 4, MS = M + 4;
  if constexpr (kLinX3) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row0 = 16 * j + 8 * hh;
      Split3 ga[M / 32], xb[K / 32];
#pragma unroll
      for (int a = 0; a < M / 32; ++a) {
        float f[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { f[q] = Gt[(row0 + q) * MS + r + 32 * a]; gbp[a] += f[q]; }
        ga[a] = split8(make_float4(f[0], f[1], f[2], f[3]), make_float4(f[4], f[5], f[6], f[7]));
      }
#pragma unroll
      for (int b = 0; b < K / 32; ++b) {
        float f[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = Xt[(row0 + q) * KS + r + 32 * b];
        xb[b] = split8(make_float4(f[0], f[1], f[2], f[3]), make_float4(f[4], f[5], f[6], f[7]));
      }
#define PG_W3(GT, XT)                                                                                     \
      _Pragma("unroll") for (int a = 0; a < M / 32; ++a)                                                  \
      _Pragma("unroll") for (int b = 0; b < K / 32; ++b)                                                  \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[a].GT, xb[b].XT, acc[a][b], 0, 0, 0);
      PG_W3(lo, hi) PG_W3(hi, lo) PG_W3(mid, mid) PG_W3(mid, hi) PG_W3(hi, mid) PG_W3(hi, hi)
#undef PG_W3
    }
  } else {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int row = 2 * s + hh;
      float av[M / 32], bv[K / 32];
#pragma unroll
      for (int a = 0; a < M / 32; ++a) { av[a] = Gt[row * MS + r + 32 * a]; gbp[a] += av[a]; }
#pragma unroll
      for (int b = 0; b < K / 32; ++b) bv[b] = Xt[row * KS + r + 32 * b];
#pragma unroll
      for (int a = 0; a < M / 32; ++a)
#pragma unroll
        for (int b = 0; b < K / 32; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
  }
}

// wgrad_tile's split-bf16 form with the X operand GENERATED in registers (functional._EmbedConvInLinear): lane (column
// k = r + 32 b, row half hh) forms ELU(h[row][k]) for its 8 rows of a 16-row step from the rows' (r, s) — broadcast reads of
// the wave's 64-float rs tile — and its own columns' (a, c, b_in).  Same values, same splits, same matrix instructions in
// the same order as wgrad_tile over a stored tile: bit-identical accumulators, no [32][K] tile, no shuffles.
template <int K, int M>
__device__ __forceinline__ void gen_wgrad_tile_x3(const float* Gt, const float* rs, const float (&ca)[K / 32],
                                                  const float (&cc)[K / 32], const float (&cb)[K / 32], int r, int hh,
                                                  f32x16 (&acc)[M / 32][K / 32], float (&gbp)[M / 32]) {
  constexpr int MS = M + 4;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row0 = 16 * j + 8 * hh;
    const float4 R0 = *reinterpret_cast<const float4*>(rs + row0), R1 = *reinterpret_cast<const float4*>(rs + row0 + 4);
    const float4 S0 = *reinterpret_cast<const float4*>(rs + 32 + row0), S1 = *reinterpret_cast<const float4*>(rs + 32 + row0 + 4);
    const float rr[8] = {R0.x, R0.y, R0.z, R0.w, R1.x, R1.y, R1.z, R1.w};
    const float ss[8] = {S0.x, S0.y, S0.z, S0.w, S1.x, S1.y, S1.z, S1.w};
    Split3 ga[M / 32], xb[K / 32];
#pragma unroll
    for (int a = 0; a < M / 32; ++a) {
      float f[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { f[q] = Gt[(row0 + q) * MS + r + 32 * a]; gbp[a] += f[q]; }
      ga[a] = split8(make_float4(f[0], f[1], f[2], f[3]), make_float4(f[4], f[5], f[6], f[7]));
    }
#pragma unroll
    for (int b = 0; b < K / 32; ++b) {
      float f[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) f[q] = elu1(gen_h(rr[q], ss[q], ca[b], cc[b], cb[b]));
      xb[b] = split8(make_float4(f[0], f[1], f[2], f[3]), make_float4(f[4], f[5], f[6], f[7]));
    }
#define PG_W3(GT, XT)                                                                                     \
    _Pragma("unroll") for (int a = 0; a < M / 32; ++a)                                                    \
    _Pragma("unroll") for (int b = 0; b < K / 32; ++b)                                                    \
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[a].GT, xb[b].XT, acc[a][b], 0, 0, 0);
    PG_W3(lo, hi) PG_W3(hi, lo) PG_W3(mid, mid) PG_W3(mid, hi) PG_W3(hi, mid) PG_W3(hi, hi)
#undef PG_W3
  }
}

template <int K, int M>
struct WgradGeo {
  static constexpr int SLAB = M * K + M;     // gw | gb
};

// Workgroup finish of the weight-gradient kernels: the four waves' accumulators are added as a fixed binary tree,
// (w0 + w1) + (w2 + w3), through LDS — value v of lane l at [v][l]: every store and load at its own address, two rounds —
// and wave 0 writes the slab (gw [M][K] | gb [M]).  (Until round 4 the waves took turns adding into one LDS image: four
// serialised rounds of dependent read-add-write steps — nothing at N = 1e6 rows, a third of the kernel on a mini-batch.)
template <int K, int M, int LDS_F>
__device__ __forceinline__ void wgrad_finish(const f32x16 (&acc)[M / 32][K / 32], const float (&gbp)[M / 32], float* lds,
                                             float* __restrict__ slab, int wave, int lane) {
  constexpr int NA = (M / 32) * (K / 32) * 16, NV = NA + M / 32;
  static_assert(2 * NV * 64 <= LDS_F, "tree buffers of the wgrad finish");
  const int r = lane & 31, hh = lane >> 5;
  float val[NV];
#pragma unroll
  for (int a = 0; a < M / 32; ++a) {
#pragma unroll
    for (int b = 0; b < K / 32; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) val[(a * (K / 32) + b) * 16 + i] = acc[a][b][i];
    val[NA + a] = gbp[a];
  }
  __syncthreads();                                   // every wave is done with its tile images
#pragma unroll
  for (int d = 1; d < 4; d <<= 1) {
    const bool writer = (wave & (2 * d - 1)) == d, reader = (wave & (2 * d - 1)) == 0;
    float* mine = lds + (size_t)(d == 1 ? (wave >> 1) : 0) * (NV * 64) + lane;      // pairs (0,1) (2,3), then (0,2)
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
  // wave 0 lays the sums out as the slab in LDS (the tree buffers are free again), every thread copies its share out
  // (coalesced stores from the whole workgroup instead of 130 stores per lane of one wave)
  if (wave == 0) {
#pragma unroll
    for (int a = 0; a < M / 32; ++a) {
#pragma unroll
      for (int b = 0; b < K / 32; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) lds[(32 * a + jrow(i, hh)) * K + r + 32 * b] = val[(a * (K / 32) + b) * 16 + i];    // gw[m][k]
      if (hh == 0) lds[M * K + r + 32 * a] = val[NA + a];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < M * K + M; i += 256) slab[i] = lds[i];
}

template <int K, int M, int ACT, typename TG = float, typename TX = float>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const TG* __restrict__ g, int64_t ldg,
                                                           const TX* __restrict__ x, int64_t ldx,
                                                           int64_t n, int64_t n_tiles,
                                                           float* __restrict__ slabs) {
  constexpr int KS = K + 4, MS = M + 4;
  constexpr int PER_WAVE = 32 * KS + 32 * MS;
  constexpr int SLAB = WgradGeo<K, M>::SLAB;
  constexpr int LDS_F = (4 * PER_WAVE > SLAB) ? 4 * PER_WAVE : SLAB;
  __shared__ __attribute__((aligned(16))) float lds[LDS_F];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // SGPR: tile addresses stay scalar
  float* Xt = lds + wave * PER_WAVE;
  float* Gt = Xt + 32 * KS;
  const int r = lane & 31, hh = lane >> 5;
  f32x16 acc[M / 32][K / 32];
  float gbp[M / 32];
#pragma unroll
  for (int a = 0; a < M / 32; ++a) {
    gbp[a] = 0.f;
#pragma unroll
    for (int b = 0; b < K / 32; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  }
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  RowRegs<K, TX> rx;
  RowRegs<M, TG> rgm;
  load_rows<K, TX>(x, ldx, n, tile * 32, lane, rx);
  load_rows<M, TG>(g, ldg, n, tile * 32, lane, rgm);
  for (; tile < n_tiles; tile += stride) {
    store_rows<K, ACT, TX>(rx, lane, Xt);
    store_rows<M, 0, TG>(rgm, lane, Gt);
    load_rows<K, TX>(x, ldx, n, (tile + stride) * 32, lane, rx);
    load_rows<M, TG>(g, ldg, n, (tile + stride) * 32, lane, rgm);
    wave_sync_lds();
    wgrad_tile<K, M>(Xt, Gt, r, hh, acc, gbp);
    wave_sync_lds();
  }
#pragma unroll
  for (int a = 0; a < M / 32; ++a) gbp[a] += __shfl_xor(gbp[a], 32);
  wgrad_finish<K, M, LDS_F>(acc, gbp, lds, slabs + (int64_t)blockIdx.x * SLAB, wave, lane);
}

__global__ __launch_bounds__(kSumThreads) void slab_reduce_kernel(const float* __restrict__ slabs, int n_slabs,
                                                                  int slab_len, int split,
                                                                  float* __restrict__ out0,
                                                                  float* __restrict__ out1) {
  const int i = blockIdx.x * kWave + (threadIdx.x & (kWave - 1));
  const float s = ordered_parts_sum(slabs, n_slabs, slab_len, i, slab_len);
  if (threadIdx.x >= kWave || i >= slab_len) return;
  if (i < split) out0[i] = s;
  else if (out1) out1[i - split] = s;
}

static int num_cus() {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) == hipSuccess &&
      hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
    return v;
  return 256;
}

template <int K, int M, typename TX, typename TY, typename TG>
static int launch_fwd_t(const void* x, int64_t ldx, const float* w, int w_trans, const float* bias, void* y, int64_t ldy,
                        int64_t n, int in_act, const void* gate, int64_t ldgate, hipStream_t s) {
  const int w_sm = w_trans ? 1 : K, w_sk = w_trans ? M : 1;
  constexpr int WAVES = FwdGeo<K, M>::WAVES;
  const int64_t n_tiles = (n + 31) / 32;
  int64_t grid = (n_tiles + WAVES - 1) / WAVES;
  const int64_t cap = (int64_t)num_cus();          // one workgroup per CU fills its LDS
  if (grid > cap) grid = cap;
  const dim3 g((unsigned)grid), b(WAVES * 64);
  const TX* xp = static_cast<const TX*>(x);
  TY* yp = static_cast<TY*>(y);
  const TG* gp = static_cast<const TG*>(gate);
  if (gate)
    hipLaunchKernelGGL((linear_fwd_kernel<K, M, 0, true, TX, TY, TG>), g, b, 0, s, xp, ldx, w, w_sm, w_sk, bias, yp, ldy, n, n_tiles, gp, ldgate);
  else if (in_act)
    hipLaunchKernelGGL((linear_fwd_kernel<K, M, 1, false, TX, TY, float>), g, b, 0, s, xp, ldx, w, w_sm, w_sk, bias, yp, ldy, n, n_tiles,
                       static_cast<const float*>(nullptr), ldgate);
  else
    hipLaunchKernelGGL((linear_fwd_kernel<K, M, 0, false, TX, TY, float>), g, b, 0, s, xp, ldx, w, w_sm, w_sk, bias, yp, ldy, n, n_tiles,
                       static_cast<const float*>(nullptr), ldgate);
  PG_CHECK_LAUNCH("pangnn_linear_fwd");
  return 0;
}

// storage-type dispatch.  A gated product (dL/dx of "dense after ELU") writes the gradient of the gate tensor, so its
// result is stored like the gate: (TG, TY) is (f32, f32) or (H, H).  H: the 2-byte format of this call's 16-bit operands
// (bf16s or f16s — one call never mixes the two; x16 / y16 / gate16 say which operands are stored in it).
template <int K, int M, typename H>
static int launch_fwd(const void* x, int x16, int64_t ldx, const float* w, int w_trans, const float* bias, void* y,
                      int y16, int64_t ldy, int64_t n, int in_act, const void* gate, int gate16, int64_t ldgate,
                      hipStream_t s) {
#define PG_FWD(TX, TY, TG) return launch_fwd_t<K, M, TX, TY, TG>(x, ldx, w, w_trans, bias, y, ldy, n, in_act, gate, ldgate, s)
  if (gate) {
    if (gate16) { if (x16) PG_FWD(H, H, H); PG_FWD(float, H, H); }
    if (x16) PG_FWD(H, float, float);
    PG_FWD(float, float, float);
  }
  if (x16) { if (y16) PG_FWD(H, H, float); PG_FWD(H, float, float); }
  if (y16) PG_FWD(float, H, float);
  PG_FWD(float, float, float);
#undef PG_FWD
}

template <int K, int M, typename TG, typename TX>
static int launch_wgrad_t(const void* g, int64_t ldg, const void* x, int64_t ldx, int64_t n, int in_act, float* gw,
                          float* gb, float* ws, size_t ws_bytes, hipStream_t s) {
  constexpr int SLAB = WgradGeo<K, M>::SLAB;
  const int64_t n_tiles = (n + 31) / 32;
  int64_t grid = (n_tiles + 3) / 4;
  const int64_t cap = (int64_t)num_cus();
  if (grid > cap) grid = cap;
  if (grid < 1) grid = 1;
  PG_CHECK_ARG(ws && ws_bytes >= (size_t)grid * SLAB * sizeof(float), PANGNN_E_WORKSPACE,
               "pangnn_linear_wgrad: workspace too small");
  const TG* gp = static_cast<const TG*>(g);
  const TX* xp = static_cast<const TX*>(x);
  if (in_act)
    hipLaunchKernelGGL((linear_wgrad_kernel<K, M, 1, TG, TX>), dim3((unsigned)grid), dim3(256), 0, s, gp, ldg, xp, ldx, n,
                       n_tiles, ws);
  else
    hipLaunchKernelGGL((linear_wgrad_kernel<K, M, 0, TG, TX>), dim3((unsigned)grid), dim3(256), 0, s, gp, ldg, xp, ldx, n,
                       n_tiles, ws);
  PG_CHECK_LAUNCH("pangnn_linear_wgrad");
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((SLAB + kWave - 1) / kWave), dim3(kSumThreads), 0, s, ws, (int)grid,
                     SLAB, M * K, gw, gb);
  PG_CHECK_LAUNCH("pangnn_linear_wgrad(reduce)");
  return 0;
}

template <int K, int M, typename H>
static int launch_wgrad(const void* g, int g16, int64_t ldg, const void* x, int x16, int64_t ldx, int64_t n,
                        int in_act, float* gw, float* gb, float* ws, size_t ws_bytes, hipStream_t s) {
#define PG_WG(TG, TX) return launch_wgrad_t<K, M, TG, TX>(g, ldg, x, ldx, n, in_act, gw, gb, ws, ws_bytes, s)
  if (g16) { if (x16) PG_WG(H, H); PG_WG(H, float); }
  if (x16) PG_WG(float, H);
  PG_WG(float, float);
#undef PG_WG
}


// ==============================================================================================================
// First layer by linearity, fused into the dense layer that follows it (gnn.py:97,125-166 for the scalar-feature model):
//     h[i][k] = r_i a_k + s_i c_k + b_k          (functional._EmbedConvIn: r = A_hat x, s = A_hat 1, a = W_in w, c = W_in b_emb)
//     y = ELU(h) W_out^T (+ bias)                 (conv_out's / linear_out's dense part)
// The [N, H] rows of h are GENERATED inside the kernels that would read them — 8 bytes per row (r_i, s_i) instead of 4 H:
//   gen_linear_fwd_kernel         y = ELU(h) W_out^T (+ bias)          (h is never written)
//   gen_linear_wgrad_kernel       dL/dW_out = g^T ELU(h), dL/dbias     (h regenerated)
//   gen_linear_dgrad_sums_kernel  [r s 1]^T ((g W_out) * ELU'(h))      (dL/dh is never written: its three weighted column
//                                                                       sums are all the first layer's parameters need)
// The generated values are exactly embed_conv_in_rows_kernel's (same fused multiply-adds), so the forward result equals the
// layer-by-layer evaluation bit for bit; the backward sums differ from it by fp32 re-association only.
// ==============================================================================================================
// acb[0..K) = a, [K..2K) = c, [2K..3K) = b_in (0 when absent): same dot-product order as embed_conv_in_rows_kernel
template <int K>
__device__ __forceinline__ void gen_params(const float* __restrict__ w_emb, const float* __restrict__ b_emb,
                                           const float* __restrict__ w_in, const float* __restrict__ b_in, int D, float* acb,
                                           int n_threads) {
  for (int k = threadIdx.x; k < K; k += n_threads) {
    float a = 0.f, c = 0.f;
    const float* wr = w_in + (int64_t)k * D;
    // sixteen loads in flight (the sums stay in order): one load -> fma round per cache round trip was ~10 us of every
    // kernel that starts here when the launch is a mini-batch's handful of workgroups
#pragma unroll 16
    for (int d = 0; d < D; ++d) { a = fmaf(wr[d], w_emb[d], a); c = fmaf(wr[d], b_emb[d], c); }
    acb[k] = a;
    acb[K + k] = c;
    acb[2 * K + k] = b_in ? b_in[k] : 0.f;
  }
}

// r of rows [base, base + 32) in lanes 0..31, s of the same rows in lanes 32..63; rows >= n read as 0
__device__ __forceinline__ float gen_load(const float* __restrict__ rv, const float* __restrict__ sv, int64_t n, int64_t base,
                                          int lane) {
  const int64_t row = base + (lane & 31);
  const float* p = lane < 32 ? rv : sv;
  return row < n ? p[row] : 0.f;
}

// the [32][C+4] LDS tile of ELU(h) for the 32 rows whose (r, s) sit in `rsv` (gen_load)
template <int C>
__device__ __forceinline__ void gen_store_rows(float rsv, int lane, float* tile, float4 a4, float4 c4, float4 b4) {
  constexpr int LPR = C / 4;
  constexpr int RPI = 64 / LPR;
  constexpr int RS = C + 4;
  const int q = lane % LPR, r0 = lane / LPR;
#pragma unroll
  for (int i = 0; i < 32 / RPI; ++i) {
    const int row = i * RPI + r0;
    const float rv = __shfl(rsv, row), sv = __shfl(rsv, 32 + row);
    float4 v;
    v.x = elu1(gen_h(rv, sv, a4.x, c4.x, b4.x));
    v.y = elu1(gen_h(rv, sv, a4.y, c4.y, b4.y));
    v.z = elu1(gen_h(rv, sv, a4.z, c4.z, b4.z));
    v.w = elu1(gen_h(rv, sv, a4.w, c4.w, b4.w));
    *reinterpret_cast<float4*>(tile + row * RS + 4 * q) = v;
  }
}

template <int K, int M>
__global__ __launch_bounds__((FwdGeo<K, M>::WAVES * 64)) void gen_linear_fwd_kernel(
    const float* __restrict__ rvec, const float* __restrict__ svec, const float* __restrict__ w_emb,
    const float* __restrict__ b_emb, const float* __restrict__ w_in, const float* __restrict__ b_in, int D,
    const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y, int64_t ldy, int64_t n,
    int64_t n_tiles) {
  constexpr int KS = K + 4;
  constexpr int WAVES = FwdGeo<K, M>::WAVES;
  constexpr int WF = (WImg<K, M>::FLOATS + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float lds[WF + WAVES * 32 * KS];
  __shared__ __attribute__((aligned(16))) float acb[3 * K];
  float* Wl = lds;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float* Xt = lds + WF + wave * (32 * KS);
  stage_w<K, M>(w, K, 1, Wl, WAVES * 64);
  gen_params<K>(w_emb, b_emb, w_in, b_in, D, acb, WAVES * 64);
  __syncthreads();
  const int q = lane % (K / 4);
  const float4 a4 = *reinterpret_cast<const float4*>(acb + 4 * q), c4 = *reinterpret_cast<const float4*>(acb + K + 4 * q);
  const float4 b4 = *reinterpret_cast<const float4*>(acb + 2 * K + 4 * q);
  const int r = lane & 31, hh = lane >> 5;
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
  float bv[M / 32];
#pragma unroll
  for (int b = 0; b < M / 32; ++b) bv[b] = bias ? bias[r + 32 * b] : 0.f;
  f32x16 acc[M / 32];
  float rsv = gen_load(rvec, svec, n, tile * 32, lane);
  for (; tile < n_tiles; tile += stride) {
    gen_store_rows<K>(rsv, lane, Xt, a4, c4, b4);
    rsv = gen_load(rvec, svec, n, (tile + stride) * 32, lane);
    wave_sync_lds();
    tile_mult<K, M>(Xt, Wl, r, hh, acc);
    if (tile * 32 + 32 <= n) {                     // wave-uniform: full tile, 128-byte row segments
      uint32_t loff = (4u * hh * (uint32_t)ldy + r) * (uint32_t)sizeof(float);
      int64_t sbase = tile * 32;
      pin(sbase, loff);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float* yr = y + (sbase + (i & 3) + 8 * (i >> 2)) * ldy;
#pragma unroll
        for (int b = 0; b < M / 32; ++b) st_f32(yr + 32 * b, loff, acc[b][i] + bv[b]);
      }
    } else {
#pragma unroll
      for (int b = 0; b < M / 32; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int64_t row = tile * 32 + jrow(i, hh);
          if (row < n) y[row * ldy + r + 32 * b] = acc[b][i] + bv[b];
        }
    }
    wave_sync_lds();
  }
}

// The forward product with the A operand generated IN REGISTERS (split-bf16 product only): lane (row r, half hh) of a
// 32x32x16 step needs ELU(h[r][k]) for 8 consecutive k — from its own row's (r_i, s_i) and the 8 columns' (a, c, b_in),
// read from LDS as three broadcast pairs of ds_read_b128.  No row tile, no shuffles, no wave barrier: the only LDS
// residents are the weight images and the 3 K parameter floats, so 8 waves (two per SIMD) share a workgroup and hide each
// other's exp / split phases.  Same values, same products, same order as gen_linear_fwd_kernel: bit-identical results.
template <int K, int M>
__global__ __launch_bounds__(512) void gen_linear_fwd_reg_kernel(
    const float* __restrict__ rvec, const float* __restrict__ svec, const float* __restrict__ w_emb,
    const float* __restrict__ b_emb, const float* __restrict__ w_in, const float* __restrict__ b_in, int D,
    const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y, int64_t ldy, int64_t n,
    int64_t n_tiles) {
  constexpr int WAVES = 8;
  constexpr int WS = WImg<K, M>::WS, IMG = M * WS;
  constexpr int WF = (WImg<K, M>::FLOATS + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float lds[WF];
  __shared__ __attribute__((aligned(16))) float acb[3 * K];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  stage_w<K, M>(w, K, 1, lds, WAVES * 64);
  gen_params<K>(w_emb, b_emb, w_in, b_in, D, acb, WAVES * 64);
  __syncthreads();
  const int r = lane & 31, hh = lane >> 5;
  const unsigned short* wb = reinterpret_cast<const unsigned short*>(lds) + r * WS + 8 * hh;
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
  float bv[M / 32];
#pragma unroll
  for (int b = 0; b < M / 32; ++b) bv[b] = bias ? bias[r + 32 * b] : 0.f;
  auto row_rs = [&](int64_t t, float& rv, float& sv) {
    const int64_t row = t * 32 + r;
    const bool in = row < n;
    rv = in ? rvec[row] : 0.f;
    sv = in ? svec[row] : 0.f;
  };
  float rv, sv;
  row_rs(tile, rv, sv);
  for (; tile < n_tiles; tile += stride) {
    float rn, sn;
    row_rs(tile + stride, rn, sn);
    f32x16 acc[M / 32];
#pragma unroll
    for (int b = 0; b < M / 32; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
#pragma unroll
    for (int i = 0; i < K / 16; ++i) {
      const int k0 = 16 * i + 8 * hh;
      bf16x8 wf[M / 32][3];
#pragma unroll
      for (int b = 0; b < M / 32; ++b)
#pragma unroll
        for (int t = 0; t < 3; ++t) wf[b][t] = *reinterpret_cast<const bf16x8*>(wb + 32 * b * WS + 16 * i + t * IMG);
      const float4 a0 = *reinterpret_cast<const float4*>(acb + k0), a1 = *reinterpret_cast<const float4*>(acb + k0 + 4);
      const float4 c0 = *reinterpret_cast<const float4*>(acb + K + k0), c1 = *reinterpret_cast<const float4*>(acb + K + k0 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(acb + 2 * K + k0), b1 = *reinterpret_cast<const float4*>(acb + 2 * K + k0 + 4);
      float4 x0, x1;
      x0.x = elu1(gen_h(rv, sv, a0.x, c0.x, b0.x)); x0.y = elu1(gen_h(rv, sv, a0.y, c0.y, b0.y));
      x0.z = elu1(gen_h(rv, sv, a0.z, c0.z, b0.z)); x0.w = elu1(gen_h(rv, sv, a0.w, c0.w, b0.w));
      x1.x = elu1(gen_h(rv, sv, a1.x, c1.x, b1.x)); x1.y = elu1(gen_h(rv, sv, a1.y, c1.y, b1.y));
      x1.z = elu1(gen_h(rv, sv, a1.z, c1.z, b1.z)); x1.w = elu1(gen_h(rv, sv, a1.w, c1.w, b1.w));
      const Split3 a = split8(x0, x1);
#define PG_X3(AT, WT)                                                                                     \
      _Pragma("unroll") for (int b = 0; b < M / 32; ++b)                                                  \
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.AT, wf[b][WT], acc[b], 0, 0, 0);
      PG_X3(lo, 0) PG_X3(hi, 2) PG_X3(mid, 1) PG_X3(mid, 0) PG_X3(hi, 1) PG_X3(hi, 0)
#undef PG_X3
    }
    if (tile * 32 + 32 <= n) {                     // wave-uniform: full tile, 128-byte row segments
      uint32_t loff = (4u * hh * (uint32_t)ldy + r) * (uint32_t)sizeof(float);
      int64_t sbase = tile * 32;
      pin(sbase, loff);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float* yr = y + (sbase + (i & 3) + 8 * (i >> 2)) * ldy;
#pragma unroll
        for (int b = 0; b < M / 32; ++b) st_f32(yr + 32 * b, loff, acc[b][i] + bv[b]);
      }
    } else {
#pragma unroll
      for (int b = 0; b < M / 32; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int64_t row = tile * 32 + jrow(i, hh);
          if (row < n) y[row * ldy + r + 32 * b] = acc[b][i] + bv[b];
        }
    }
    rv = rn;
    sv = sn;
  }
}

// dL/dW_out [M][K] (+ dL/dbias [M]) with x = ELU(h) regenerated: linear_wgrad_kernel's accumulation and slab reduction
template <int K, int M>
__global__ __launch_bounds__(256) void gen_linear_wgrad_kernel(const float* __restrict__ g, int64_t ldg,
                                                               const float* __restrict__ rvec, const float* __restrict__ svec,
                                                               const float* __restrict__ w_emb, const float* __restrict__ b_emb,
                                                               const float* __restrict__ w_in, const float* __restrict__ b_in,
                                                               int D, int64_t n, int64_t n_tiles, float* __restrict__ slabs) {
  constexpr int KS = K + 4, MS = M + 4;
  constexpr int PER_WAVE = 32 * KS + 32 * MS;
  constexpr int SLAB = WgradGeo<K, M>::SLAB;
  constexpr int LDS_F = (4 * PER_WAVE > SLAB) ? 4 * PER_WAVE : SLAB;
  __shared__ __attribute__((aligned(16))) float lds[LDS_F];
  __shared__ __attribute__((aligned(16))) float acb[3 * K];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float* Xt = lds + wave * PER_WAVE;
  float* Gt = Xt + 32 * KS;
  gen_params<K>(w_emb, b_emb, w_in, b_in, D, acb, 256);
  __syncthreads();
  const int q = lane % (K / 4);
  const float4 a4 = *reinterpret_cast<const float4*>(acb + 4 * q), c4 = *reinterpret_cast<const float4*>(acb + K + 4 * q);
  const float4 b4 = *reinterpret_cast<const float4*>(acb + 2 * K + 4 * q);
  const int r = lane & 31, hh = lane >> 5;
  float ca[K / 32], cc[K / 32], cb[K / 32];      // (a, c, b_in) of this lane's columns r + 32 b (split-bf16 form)
#pragma unroll
  for (int b = 0; b < K / 32; ++b) { ca[b] = acb[r + 32 * b]; cc[b] = acb[K + r + 32 * b]; cb[b] = acb[2 * K + r + 32 * b]; }
  f32x16 acc[M / 32][K / 32];
  float gbp[M / 32];
#pragma unroll
  for (int a = 0; a < M / 32; ++a) {
    gbp[a] = 0.f;
#pragma unroll
    for (int b = 0; b < K / 32; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  }
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  RowRegs<M, float> rgm;
  float rsv = gen_load(rvec, svec, n, tile * 32, lane);
  load_rows<M, float>(g, ldg, n, tile * 32, lane, rgm);
  for (; tile < n_tiles; tile += stride) {
    if constexpr (kLinX3) Xt[lane] = rsv;                   // the tile's (r, s): the X operand is formed in registers
    else gen_store_rows<K>(rsv, lane, Xt, a4, c4, b4);
    store_rows<M, 0, float>(rgm, lane, Gt);                 // rows >= n are zero: their generated x contributes nothing
    rsv = gen_load(rvec, svec, n, (tile + stride) * 32, lane);
    load_rows<M, float>(g, ldg, n, (tile + stride) * 32, lane, rgm);
    wave_sync_lds();
    if constexpr (kLinX3) gen_wgrad_tile_x3<K, M>(Gt, Xt, ca, cc, cb, r, hh, acc, gbp);
    else wgrad_tile<K, M>(Xt, Gt, r, hh, acc, gbp);
    wave_sync_lds();
  }
#pragma unroll
  for (int a = 0; a < M / 32; ++a) gbp[a] += __shfl_xor(gbp[a], 32);
  wgrad_finish<K, M, LDS_F>(acc, gbp, lds, slabs + (int64_t)blockIdx.x * SLAB, wave, lane);
}

// sums[3][H] = [r s 1]^T ((g W_out) * ELU'(h)): KG = width of g (the dense layer's output width), H = width of h.
// The product is linear_fwd_kernel's (x = g, w = W_out^T staged transposed out of W_out [KG][H]); the epilogue multiplies
// by ELU'(h) of the regenerated h and accumulates the three weighted column sums per lane over all of the wave's tiles;
// lanes -> waves -> workgroup slab in a fixed order, slabs summed by slab_reduce_kernel (bitwise reproducible).
template <int KG, int H>
__global__ __launch_bounds__((FwdGeo<KG, H>::WAVES * 64)) void gen_linear_dgrad_sums_kernel(
    const float* __restrict__ g, int64_t ldg, const float* __restrict__ w_out, const float* __restrict__ rvec,
    const float* __restrict__ svec, const float* __restrict__ w_emb, const float* __restrict__ b_emb,
    const float* __restrict__ w_in, const float* __restrict__ b_in, int D, int64_t n, int64_t n_tiles,
    float* __restrict__ slabs) {
  constexpr int KS = KG + 4;
  constexpr int WAVES = FwdGeo<KG, H>::WAVES;
  constexpr int WF = (WImg<KG, H>::FLOATS + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float lds[WF + WAVES * 32 * KS];
  __shared__ __attribute__((aligned(16))) float acb[3 * H];
  __shared__ float part[WAVES][3 * H];
  float* Wl = lds;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float* Xt = lds + WF + wave * (32 * KS);
  stage_w<KG, H>(w_out, 1, H, Wl, WAVES * 64);                    // operand [H][KG]: element (h, m) = W_out[m][h]
  gen_params<H>(w_emb, b_emb, w_in, b_in, D, acb, WAVES * 64);
  __syncthreads();
  const int r = lane & 31, hh = lane >> 5;
  float ca[H / 32], cc[H / 32], cb[H / 32];
#pragma unroll
  for (int b = 0; b < H / 32; ++b) { ca[b] = acb[r + 32 * b]; cc[b] = acb[H + r + 32 * b]; cb[b] = acb[2 * H + r + 32 * b]; }
  float s0[H / 32], s1[H / 32], s2[H / 32];
#pragma unroll
  for (int b = 0; b < H / 32; ++b) s0[b] = s1[b] = s2[b] = 0.f;
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
  f32x16 acc[H / 32];
  RowRegs<KG, float> rg;
  load_rows<KG, float>(g, ldg, n, tile * 32, lane, rg);
  float rsv = gen_load(rvec, svec, n, tile * 32, lane);
  for (; tile < n_tiles; tile += stride) {
    store_rows<KG, 0, float>(rg, lane, Xt);
    const float rs_cur = rsv;
    load_rows<KG, float>(g, ldg, n, (tile + stride) * 32, lane, rg);
    rsv = gen_load(rvec, svec, n, (tile + stride) * 32, lane);
    wave_sync_lds();
    tile_mult<KG, H>(Xt, Wl, r, hh, acc);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = jrow(i, hh);
      const float rv = __shfl(rs_cur, row), sv = __shfl(rs_cur, 32 + row);
#pragma unroll
      for (int b = 0; b < H / 32; ++b) {
        const float v = acc[b][i] * elu1_grad(gen_h(rv, sv, ca[b], cc[b], cb[b]));     // rows >= n: g = 0 -> acc = 0
        s0[b] = fmaf(rv, v, s0[b]);
        s1[b] = fmaf(sv, v, s1[b]);
        s2[b] += v;
      }
    }
    wave_sync_lds();
  }
#pragma unroll
  for (int b = 0; b < H / 32; ++b) {
    s0[b] += __shfl_xor(s0[b], 32);
    s1[b] += __shfl_xor(s1[b], 32);
    s2[b] += __shfl_xor(s2[b], 32);
    if (hh == 0) { part[wave][r + 32 * b] = s0[b]; part[wave][H + r + 32 * b] = s1[b]; part[wave][2 * H + r + 32 * b] = s2[b]; }
  }
  __syncthreads();
  float* slab = slabs + (int64_t)blockIdx.x * (3 * H);
  for (int i = threadIdx.x; i < 3 * H; i += WAVES * 64) {
    float t = part[0][i];
#pragma unroll
    for (int wv = 1; wv < WAVES; ++wv) t += part[wv][i];
    slab[i] = t;
  }
}

template <int K, int M>
static int launch_gen_fwd(const float* r, const float* sv, const float* w_emb, const float* b_emb, const float* w_in,
                          const float* b_in, int D, const float* w, const float* bias, float* y, int64_t ldy, int64_t n,
                          hipStream_t s) {
  const int64_t n_tiles = (n + 31) / 32;
  if constexpr (WImg<K, M>::X3) {
    int64_t grid = (n_tiles + 7) / 8;
    const int64_t cap = 2 * (int64_t)num_cus();      // two 8-wave workgroups per CU fit (weight images <= 56 KB each)
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL((gen_linear_fwd_reg_kernel<K, M>), dim3((unsigned)grid), dim3(512), 0, s, r, sv, w_emb, b_emb, w_in, b_in,
                       D, w, bias, y, ldy, n, n_tiles);
    PG_CHECK_LAUNCH("pangnn_embed_linear_fwd");
    return 0;
  }
  constexpr int WAVES = FwdGeo<K, M>::WAVES;
  int64_t grid = (n_tiles + WAVES - 1) / WAVES;
  const int64_t cap = (int64_t)num_cus();
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL((gen_linear_fwd_kernel<K, M>), dim3((unsigned)grid), dim3(WAVES * 64), 0, s, r, sv, w_emb, b_emb, w_in, b_in,
                     D, w, bias, y, ldy, n, n_tiles);
  PG_CHECK_LAUNCH("pangnn_embed_linear_fwd");
  return 0;
}

static int64_t gen_grid(int64_t n, int waves) {
  const int64_t n_tiles = (n + 31) / 32;
  int64_t grid = (n_tiles + waves - 1) / waves;
  const int64_t cap = (int64_t)num_cus();
  if (grid > cap) grid = cap;
  return grid < 1 ? 1 : grid;
}

template <int K, int M>
static int launch_gen_bwd(const float* g, int64_t ldg, const float* r, const float* sv, const float* w_emb, const float* b_emb,
                          const float* w_in, const float* b_in, int D, const float* w_out, int64_t n, float* g_w_out,
                          float* g_b_out, float* sums, float* ws, hipStream_t s) {
  constexpr int SLAB = WgradGeo<K, M>::SLAB;
  const int64_t n_tiles = (n + 31) / 32;
  const int64_t grid_w = gen_grid(n, 4);
  hipLaunchKernelGGL((gen_linear_wgrad_kernel<K, M>), dim3((unsigned)grid_w), dim3(256), 0, s, g, ldg, r, sv, w_emb, b_emb, w_in,
                     b_in, D, n, n_tiles, ws);
  PG_CHECK_LAUNCH("pangnn_embed_linear_bwd(wgrad)");
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((SLAB + kWave - 1) / kWave), dim3(kSumThreads), 0, s, ws, (int)grid_w, SLAB,
                     M * K, g_w_out, g_b_out);
  PG_CHECK_LAUNCH("pangnn_embed_linear_bwd(wgrad reduce)");
  float* ws2 = ws + (size_t)num_cus() * SLAB;
  constexpr int WAVES = FwdGeo<M, K>::WAVES;
  const int64_t grid_d = gen_grid(n, WAVES);
  hipLaunchKernelGGL((gen_linear_dgrad_sums_kernel<M, K>), dim3((unsigned)grid_d), dim3(WAVES * 64), 0, s, g, ldg, w_out, r, sv,
                     w_emb, b_emb, w_in, b_in, D, n, n_tiles, ws2);
  PG_CHECK_LAUNCH("pangnn_embed_linear_bwd(dgrad sums)");
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((3 * K + kWave - 1) / kWave), dim3(kSumThreads), 0, s, ws2, (int)grid_d, 3 * K,
                     3 * K, sums, static_cast<float*>(nullptr));
  PG_CHECK_LAUNCH("pangnn_embed_linear_bwd(sums reduce)");
  return 0;
}

}  // namespace pangnn

using namespace pangnn;

extern "C" int pangnn_linear_supported(int32_t K, int32_t M, int wgrad) {
  const bool km = (K == 64 || K == 128) && (M == 64 || M == 128);
  if (!km) return 0;
  (void)wgrad;   // the 128 x 128 weight gradient (256 accumulator registers as one tile) runs as two 64-column halves of g
  return 1;
}

static bool dtype_ok(int32_t d) { return d == PANGNN_DTYPE_F32 || d == PANGNN_DTYPE_BF16 || d == PANGNN_DTYPE_F16; }
// the one 2-byte format of a call's operands (0 when all are f32), -1 when bfloat16 and float16 operands are mixed
static int fmt16_of(int32_t a, int32_t b, int32_t c = PANGNN_DTYPE_F32) {
  int f = 0;
  for (int32_t d : {a, b, c}) {
    if (d == PANGNN_DTYPE_F32) continue;
    if (f && f != d) return -1;
    f = d;
  }
  return f;
}
static bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

static int linear_fwd_common(const void* x, int32_t x_dtype, int64_t ldx, const float* w, int w_trans, const float* bias,
                             void* y, int32_t y_dtype, int64_t ldy, int64_t n, int32_t K, int32_t M, int32_t in_act,
                             const void* gate, int32_t gate_dtype, int64_t ldgate, pangnn_stream_t stream) {
  PG_CHECK_ARG(n >= 0, PANGNN_E_BADARG, "pangnn_linear_fwd: negative size");
  PG_CHECK_ARG(pangnn_linear_supported(K, M, 0), PANGNN_E_BADARG,
               "pangnn_linear_fwd: K and M must be 64 or 128 (got %d, %d)", (int)K, (int)M);
  PG_CHECK_ARG(dtype_ok(x_dtype) && dtype_ok(y_dtype) && (!gate || dtype_ok(gate_dtype)), PANGNN_E_BADARG,
               "pangnn_linear_fwd: storage types are PANGNN_DTYPE_F32 / _BF16 / _F16");
  PG_CHECK_ARG((in_act == 0 || in_act == 1) && !(in_act && gate) && (!gate || ldgate >= M), PANGNN_E_BADARG,
               "pangnn_linear_act_fwd: in_act must be 0 or 1 (ELU) and excludes gate; ldgate >= M");
  PG_CHECK_ARG(!gate || gate_dtype == y_dtype, PANGNN_E_BADARG,
               "pangnn_linear_act_fwd: a gated product is stored like its gate (it is the gate tensor's gradient)");
  if (n == 0) return 0;
  PG_CHECK_ARG(x && w && y && ldx >= K && ldy >= M, PANGNN_E_BADARG, "pangnn_linear_fwd: bad pointer / ld");
  PG_CHECK_ARG((x_dtype != PANGNN_DTYPE_F32 ? aligned8(x) : aligned16(x)) && aligned16(w) && ldx % 4 == 0,
               PANGNN_E_ALIGN, "pangnn_linear_fwd: x rows must start on 16 bytes (f32) / 8 bytes (bf16, f16), w on 16, ldx % 4 == 0");
  const int fmt = fmt16_of(x_dtype, y_dtype, gate ? gate_dtype : PANGNN_DTYPE_F32);
  PG_CHECK_ARG(fmt >= 0, PANGNN_E_BADARG, "pangnn_linear_fwd: the 2-byte operands of one call are all bfloat16 or all float16");
  hipStream_t s = (hipStream_t)stream;
  const int xb = x_dtype != PANGNN_DTYPE_F32, yb = y_dtype != PANGNN_DTYPE_F32, gb = gate && gate_dtype != PANGNN_DTYPE_F32;
#define PG_SHAPES(H)                                                                                                              \
  do {                                                                                                                            \
    if (K == 64 && M == 64) return launch_fwd<64, 64, H>(x, xb, ldx, w, w_trans, bias, y, yb, ldy, n, in_act, gate, gb, ldgate, s);   \
    if (K == 64 && M == 128) return launch_fwd<64, 128, H>(x, xb, ldx, w, w_trans, bias, y, yb, ldy, n, in_act, gate, gb, ldgate, s); \
    if (K == 128 && M == 64) return launch_fwd<128, 64, H>(x, xb, ldx, w, w_trans, bias, y, yb, ldy, n, in_act, gate, gb, ldgate, s); \
    return launch_fwd<128, 128, H>(x, xb, ldx, w, w_trans, bias, y, yb, ldy, n, in_act, gate, gb, ldgate, s);                         \
  } while (0)
  if (fmt == PANGNN_DTYPE_F16) PG_SHAPES(f16s);
  PG_SHAPES(bf16s);
#undef PG_SHAPES
}

extern "C" int pangnn_linear_act_fwd_mixed(const void* x, int32_t x_dtype, int64_t ldx, const float* w, const float* bias,
                                           void* y, int32_t y_dtype, int64_t ldy, int64_t n, int32_t K, int32_t M,
                                           int32_t in_act, const void* gate, int32_t gate_dtype, int64_t ldgate,
                                           pangnn_stream_t stream) {
  return linear_fwd_common(x, x_dtype, ldx, w, 0, bias, y, y_dtype, ldy, n, K, M, in_act, gate, gate_dtype, ldgate, stream);
}

// dL/dx of the dense layer y = act(x) w^T straight from the layer's own weight w [M][K] (no transposed copy: the kernel
// stages w through the strides of its transpose): gx [n][K] = g [n][M] w  (* ELU'(gate) when a gate is given)
extern "C" int pangnn_linear_dgrad_mixed(const void* g, int32_t g_dtype, int64_t ldg, const float* w, void* gx,
                                         int32_t gx_dtype, int64_t ldgx, int64_t n, int32_t K, int32_t M, const void* gate,
                                         int32_t gate_dtype, int64_t ldgate, pangnn_stream_t stream) {
  return linear_fwd_common(g, g_dtype, ldg, w, 1, nullptr, gx, gx_dtype, ldgx, n, M, K, 0, gate, gate_dtype, ldgate, stream);
}

extern "C" int pangnn_linear_act_fwd_f32(const float* x, int64_t ldx, const float* w, const float* bias, float* y,
                                         int64_t ldy, int64_t n, int32_t K, int32_t M, int32_t in_act,
                                         const float* gate, int64_t ldgate, pangnn_stream_t stream) {
  return pangnn_linear_act_fwd_mixed(x, PANGNN_DTYPE_F32, ldx, w, bias, y, PANGNN_DTYPE_F32, ldy, n, K, M, in_act, gate,
                                     PANGNN_DTYPE_F32, ldgate, stream);
}

extern "C" int pangnn_linear_fwd_f32(const float* x, int64_t ldx, const float* w, const float* bias, float* y,
                                     int64_t ldy, int64_t n, int32_t K, int32_t M, pangnn_stream_t stream) {
  return pangnn_linear_act_fwd_f32(x, ldx, w, bias, y, ldy, n, K, M, 0, nullptr, 0, stream);
}

extern "C" size_t pangnn_linear_wgrad_workspace_bytes(int32_t K, int32_t M) {
  return (size_t)num_cus() * ((size_t)M * K + M) * sizeof(float);
}

extern "C" int pangnn_linear_act_wgrad_mixed(const void* g, int32_t g_dtype, int64_t ldg, const void* x, int32_t x_dtype,
                                             int64_t ldx, int64_t n, int32_t K, int32_t M, int32_t in_act, float* gw,
                                             float* gb, void* workspace, size_t workspace_bytes, pangnn_stream_t stream) {
  PG_CHECK_ARG(n >= 0, PANGNN_E_BADARG, "pangnn_linear_wgrad: negative size");
  PG_CHECK_ARG(pangnn_linear_supported(K, M, 1), PANGNN_E_BADARG,
               "pangnn_linear_wgrad: unsupported (K, M) = (%d, %d)", (int)K, (int)M);
  PG_CHECK_ARG(in_act == 0 || in_act == 1, PANGNN_E_BADARG, "pangnn_linear_act_wgrad: in_act must be 0 or 1");
  PG_CHECK_ARG(dtype_ok(g_dtype) && dtype_ok(x_dtype), PANGNN_E_BADARG,
               "pangnn_linear_wgrad: storage types are PANGNN_DTYPE_F32 / _BF16 / _F16");
  PG_CHECK_ARG(gw && (n == 0 || (g && x)) && ldg >= M && ldx >= K, PANGNN_E_BADARG,
               "pangnn_linear_wgrad: bad pointer / ld");
  PG_CHECK_ARG((g_dtype != PANGNN_DTYPE_F32 ? aligned8(g) : aligned16(g)) &&
                   (x_dtype != PANGNN_DTYPE_F32 ? aligned8(x) : aligned16(x)) && ldg % 4 == 0 && ldx % 4 == 0,
               PANGNN_E_ALIGN, "pangnn_linear_wgrad: rows must start on 16 bytes (f32) / 8 bytes (bf16, f16), ld multiples of 4");
  const int fmt = fmt16_of(g_dtype, x_dtype);
  PG_CHECK_ARG(fmt >= 0, PANGNN_E_BADARG, "pangnn_linear_wgrad: the 2-byte operands of one call are all bfloat16 or all float16");
  hipStream_t s = (hipStream_t)stream;
  float* ws = static_cast<float*>(workspace);
  const int gbf = g_dtype != PANGNN_DTYPE_F32, xbf = x_dtype != PANGNN_DTYPE_F32;
#define PG_SHAPES(H)                                                                                                                 \
  do {                                                                                                                               \
    if (K == 64 && M == 64) return launch_wgrad<64, 64, H>(g, gbf, ldg, x, xbf, ldx, n, in_act, gw, gb, ws, workspace_bytes, s);      \
    if (K == 64 && M == 128) return launch_wgrad<64, 128, H>(g, gbf, ldg, x, xbf, ldx, n, in_act, gw, gb, ws, workspace_bytes, s);    \
    if (K == 128 && M == 128) {                                                                                                      \
      /* dW[0:64) from g[:, 0:64), dW[64:128) from g[:, 64:128): the <128, 64> kernel twice over column windows of g (same ldg), */  \
      /* stream-ordered on one workspace.  x is read twice (N * 512 B more than a single pass would need). */                        \
      const char* g_hi = static_cast<const char*>(g) + (size_t)64 * (gbf ? 2 : 4);                                                   \
      int rc = launch_wgrad<128, 64, H>(g, gbf, ldg, x, xbf, ldx, n, in_act, gw, gb, ws, workspace_bytes, s);                         \
      if (rc) return rc;                                                                                                             \
      return launch_wgrad<128, 64, H>(g_hi, gbf, ldg, x, xbf, ldx, n, in_act, gw + (size_t)64 * 128, gb ? gb + 64 : nullptr, ws,      \
                                      workspace_bytes, s);                                                                           \
    }                                                                                                                                \
    return launch_wgrad<128, 64, H>(g, gbf, ldg, x, xbf, ldx, n, in_act, gw, gb, ws, workspace_bytes, s);                             \
  } while (0)
  if (fmt == PANGNN_DTYPE_F16) PG_SHAPES(f16s);
  PG_SHAPES(bf16s);
#undef PG_SHAPES
}

extern "C" int pangnn_linear_act_wgrad_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t n,
                                           int32_t K, int32_t M, int32_t in_act, float* gw, float* gb,
                                           void* workspace, size_t workspace_bytes, pangnn_stream_t stream) {
  return pangnn_linear_act_wgrad_mixed(g, PANGNN_DTYPE_F32, ldg, x, PANGNN_DTYPE_F32, ldx, n, K, M, in_act, gw, gb,
                                       workspace, workspace_bytes, stream);
}

extern "C" int pangnn_linear_wgrad_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t n,
                                       int32_t K, int32_t M, float* gw, float* gb, void* workspace,
                                       size_t workspace_bytes, pangnn_stream_t stream) {
  return pangnn_linear_act_wgrad_f32(g, ldg, x, ldx, n, K, M, 0, gw, gb, workspace, workspace_bytes, stream);
}

// ---- first layer by linearity fused into the following dense layer (see gen_linear_*_kernel above) ----
extern "C" int pangnn_embed_linear_supported(int32_t H, int32_t M) {
  return pangnn_linear_supported(H, M, 1) && !(H == 128 && M == 128);   // no generated-rows kernels for the square 128 shape
}

extern "C" int pangnn_embed_linear_fwd(const float* r, const float* s, int64_t n, const float* w_emb, const float* b_emb,
                                       const float* w_in, const float* b_in, int32_t D, int32_t H, const float* w_out,
                                       const float* bias_out, int32_t M, float* y, int64_t ldy, pangnn_stream_t stream) {
  const char* who = "pangnn_embed_linear_fwd";
  PG_CHECK_ARG(n >= 0 && D > 0, PANGNN_E_BADARG, "%s: negative size", who);
  PG_CHECK_ARG(pangnn_embed_linear_supported(H, M), PANGNN_E_BADARG, "%s: unsupported (H, M) = (%d, %d)", who, (int)H, (int)M);
  if (n == 0) return 0;
  PG_CHECK_ARG(r && s && w_emb && b_emb && w_in && w_out && y && ldy >= M, PANGNN_E_BADARG, "%s: bad pointer / ld", who);
  PG_CHECK_ARG(aligned16(w_out), PANGNN_E_ALIGN, "%s: w_out must be 16-byte aligned", who);
  hipStream_t st = (hipStream_t)stream;
  if (H == 64 && M == 64) return launch_gen_fwd<64, 64>(r, s, w_emb, b_emb, w_in, b_in, D, w_out, bias_out, y, ldy, n, st);
  if (H == 64 && M == 128) return launch_gen_fwd<64, 128>(r, s, w_emb, b_emb, w_in, b_in, D, w_out, bias_out, y, ldy, n, st);
  return launch_gen_fwd<128, 64>(r, s, w_emb, b_emb, w_in, b_in, D, w_out, bias_out, y, ldy, n, st);
}

extern "C" size_t pangnn_embed_linear_bwd_workspace_bytes(int32_t H, int32_t M) {
  return (size_t)num_cus() * ((size_t)M * H + M + 3 * (size_t)H) * sizeof(float);
}

// g [n, M] = dL/dy.  Outputs: g_w_out [M, H], g_b_out [M] (nullable), sums [3, H] = [r s 1]^T dL/dh (feed
// pangnn_embed_conv_in_grads_from_sums).
extern "C" int pangnn_embed_linear_bwd(const float* g, int64_t ldg, const float* r, const float* s, int64_t n,
                                       const float* w_emb, const float* b_emb, const float* w_in, const float* b_in, int32_t D,
                                       int32_t H, const float* w_out, int32_t M, float* g_w_out, float* g_b_out, float* sums,
                                       void* workspace, size_t workspace_bytes, pangnn_stream_t stream) {
  const char* who = "pangnn_embed_linear_bwd";
  PG_CHECK_ARG(n >= 0 && D > 0, PANGNN_E_BADARG, "%s: negative size", who);
  PG_CHECK_ARG(pangnn_embed_linear_supported(H, M), PANGNN_E_BADARG, "%s: unsupported (H, M) = (%d, %d)", who, (int)H, (int)M);
  PG_CHECK_ARG(w_emb && b_emb && w_in && w_out && g_w_out && sums && (n == 0 || (g && r && s)) && ldg >= M && ldg % 4 == 0,
               PANGNN_E_BADARG, "%s: bad pointer / ld", who);
  PG_CHECK_ARG(workspace && workspace_bytes >= pangnn_embed_linear_bwd_workspace_bytes(H, M), PANGNN_E_WORKSPACE,
               "%s: workspace too small", who);
  PG_CHECK_ARG(n == 0 || aligned16(g), PANGNN_E_ALIGN, "%s: g rows must start on 16 bytes", who);
  hipStream_t st = (hipStream_t)stream;
  float* ws = static_cast<float*>(workspace);
#define PG_GB(HH, MM) return launch_gen_bwd<HH, MM>(g, ldg, r, s, w_emb, b_emb, w_in, b_in, D, w_out, n, g_w_out, g_b_out, sums, ws, st)
  if (H == 64 && M == 64) PG_GB(64, 64);
  if (H == 64 && M == 128) PG_GB(64, 128);
  PG_GB(128, 64);
#undef PG_GB
}
