// Fused link decoder for D = 64 (src/gnn.py:110-116,171-177), fp32, gfx950.
//
//   h1[e]   = relu(P[src_e] + Q[dst_e] (+ w_e * c))         P = z W1a^T, Q = z W1b^T + b1  (node level)
//   h2[e]   = relu(W2 h1[e] + b2)
//   logit_e = w3 . h2[e] + b3
//
// One wavefront owns a tile of 32 consecutive edges (caller's edge order, so logits are written
// coalesced and no permutation is needed).  The 32 x 64 h1 tile is gathered with row-contiguous
// 16-byte loads (16 lanes per 256-B node row, 4 rows per wave-instruction) into a padded LDS
// image and multiplied with W2 (LDS resident, loaded once per workgroup) on the f32 MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, so results match an fp32 reference to rounding).
// The product is oriented C[j][e] = sum_k W2[j][k] h1[e][k] (A = W2, B = h1^T): the edge index stays
// on the lane, so bias + relu + the dot with w3 are in-lane and one xor-32 shuffle finishes a logit.
//
// Backward recomputes the forward per tile and chains three more MFMA products without leaving the
// CU:  G = dL/dh2pre (in the accumulator registers of the first product, used directly as the A
// operand of the second), gH1 = G^T W2 (written out as dL/dh1pre [E,64]), gW2 += G h1 (operands
// re-read from LDS).  Parameter gradients are accumulated in registers across all tiles of a wave,
// reduced over the workgroup's waves in a fixed order and written as one partial slab per workgroup;
// a second tiny kernel sums the slabs in index order => bitwise reproducible, no float atomics.
#include "common.h"

#include <type_traits>

namespace pangnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef PANGNN_ABLATE_MFMA   // diagnostic builds only (hipcc -DPANGNN_ABLATE_MFMA, loaded through PANGNN_HIP_LIB): one VALU op per MFMA
__device__ __forceinline__ f32x16 fake_mfma(float a, float b, f32x16 c) { c[0] = fmaf(a, b, c[0]); return c; }
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) fake_mfma(a, b, c)
#endif

constexpr int DD = 64;        // decoder width (node_dim)
constexpr int TE = 32;        // edges per wave tile
constexpr int GS = 33;        // row stride of the G tile (floats): conflict-free column reads
constexpr int SLAB = 64 * 64 + 64 + 64 + 64 + 16;   // gW2 | gb2 | gw3 | gcvec | gb3(+pad)

// LDS images of W2 [64][64] and of the h1 tile [32][64] use a padded row stride of 68 floats
// (272 B): 16-byte aligned, ds_read_b128 down a column of rows is conflict-free (bank = 4*row + 4*k4
// mod 64 is distinct over each 16-lane service group), row-wise b32/b128 accesses are contiguous, and
// every address is `lane base + compile-time offset` (an XOR swizzle costs a VGPR per address).
constexpr int RS = 68;
__device__ __forceinline__ constexpr int swz(int row, int k) { return row * RS + k; }
__device__ __forceinline__ constexpr int swz4(int row, int k4) { return row * RS + 4 * k4; }

__device__ __forceinline__ void wave_lds_sync() {
  // LDS operations of one wave execute in issue order; this only pins the compiler's ordering.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// row of the 32x32 MFMA accumulator held in register r by a lane of half hh
__device__ __forceinline__ constexpr int jr(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

struct DecParams {
  const float* p; const float* q; uint32_t ldp4; uint32_t ldq4;   // row strides in float4 units
  const int64_t* ei; int64_t ld; int64_t E;
  const float* extra; const float* cvec;
  const float* w2; const float* b2; const float* w3; const float* b3;
};

// ---- stage W2 / b2 / w3 (/ cvec) into LDS, once per workgroup
__device__ __forceinline__ void stage_weights(const DecParams& a, float* Wl, float* b2l, float* w3l,
                                              float* cvl, int nthreads) {
  for (int i = threadIdx.x; i < 64 * 16; i += nthreads) {
    const int j = i >> 4, k4 = i & 15;
    const float4 v = reinterpret_cast<const float4*>(a.w2)[i];
    *reinterpret_cast<float4*>(Wl + swz4(j, k4)) = v;
  }
  for (int i = threadIdx.x; i < 64; i += nthreads) {
    b2l[i] = a.b2[i];
    w3l[i] = a.w3[i];
    cvl[i] = a.cvec ? a.cvec[i] : 0.f;
  }
}

// ---- gather the tile's 32 h1 rows into the wave's LDS image.  Returns w_e / validity per lane e.
// FULL: all 32 edges exist (every tile but the last): no bounds predicate on the id loads.
// max(x, 0), compiler-visible (not inline asm: an asm statement that defines a VGPR is invisible to the MFMA hazard
// recognizer and may overwrite a source register of an in-flight MFMA — see decoder16.hip)
__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff()); }

// 16-byte row piece at `table + byte_off`: uniform base pointer + 32-bit per-lane byte offset (node tables are
// < 4 GiB, checked by the host wrapper) — `global_load_dwordx4 v, v_off, s[base]`, no 64-bit vector address
__device__ __forceinline__ float4 ld_row16(const float* table, uint32_t byte_off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(table) + byte_off);
}

template <bool FULL = false>
__device__ __forceinline__ void gather_tile(const DecParams& a, int64_t ebase, int lane, float* Ht,
                                            const float* cvl, float& w_e, int& id) {
  const int64_t e = ebase + (lane & 31);
  id = 0;
  w_e = 0.f;
  if (FULL || e < a.E) {
    id = (int)a.ei[(int64_t)(lane >> 5) * a.ld + e];     // lanes 0-31: source, 32-63: target
    if (a.extra && lane < 32) w_e = a.extra[e];
  }
  const int c4 = lane & 15, r4 = lane >> 4;
  // byte offset of this lane's node row in its table (P for the source half, Q for the target half): one
  // multiply per lane and tile; the 16 row gathers below shuffle the finished offset instead of the id
  const uint32_t row_off = (uint32_t)id * ((lane >> 5) ? a.ldq4 : a.ldp4) * 16u;
  const uint32_t col_off = 16u * c4;
  const bool has_extra = a.extra != nullptr;             // uniform: the w_e * c term only exists with skip connections
  float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (has_extra) cv = reinterpret_cast<const float4*>(cvl)[c4];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    float4 pv[4], qv[4];
    float wv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * (4 * half + i) + r4;
      pv[i] = ld_row16(a.p, (uint32_t)__shfl((int)row_off, row) + col_off);
      qv[i] = ld_row16(a.q, (uint32_t)__shfl((int)row_off, 32 + row) + col_off);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                         // same association as the forward kernel: (p + q) + w c
      pv[i].x += qv[i].x; pv[i].y += qv[i].y; pv[i].z += qv[i].z; pv[i].w += qv[i].w;
    }
    if (has_extra) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wv[i] = __shfl(w_e, 4 * (4 * half + i) + r4);
        pv[i].x = fmaf(wv[i], cv.x, pv[i].x);
        pv[i].y = fmaf(wv[i], cv.y, pv[i].y);
        pv[i].z = fmaf(wv[i], cv.z, pv[i].z);
        pv[i].w = fmaf(wv[i], cv.w, pv[i].w);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * (4 * half + i) + r4;
      float4 h;
      h.x = relu1(pv[i].x);
      h.y = relu1(pv[i].y);
      h.z = relu1(pv[i].z);
      h.w = relu1(pv[i].w);
      *reinterpret_cast<float4*>(Ht + swz4(row, c4)) = h;
    }
  }
}

// ---- C[j][e] = sum_k W2[j][k] h1[e][k]; lane (e = lane&31, hh = lane>>5) gets rows jr(r,hh)+32b
__device__ __forceinline__ void gemm1(const float* Wl, const float* Ht, int lane, f32x16 (&acc)[2]) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k4 = 8 * h + i;
    const float4 bf = *reinterpret_cast<const float4*>(Ht + swz4(r, k4));
    const float4 a0 = *reinterpret_cast<const float4*>(Wl + swz4(r, k4));
    const float4 a1 = *reinterpret_cast<const float4*>(Wl + swz4(r + 32, k4));
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bf.x, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bf.x, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bf.y, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bf.y, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bf.z, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bf.z, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bf.w, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bf.w, acc[1], 0, 0, 0);
  }
}

constexpr int FWD_WAVES = 8;    // 512 threads, 1 workgroup / CU, 2 waves / SIMD (prefetch registers)
constexpr int BWD_WAVES = 8;    // 512 threads, 1 workgroup / CU, 2 waves / SIMD

// ---- software-pipelined gather: edge ids two tiles ahead, node rows one tile ahead, so a wave never
// waits on HBM between two MFMA phases (the rows of tile t+1 land while tile t is in the matrix pipe).
struct TileIds { int id; float w_e; uint32_t row_off; };     // row_off: byte offset of the lane's node row in P / Q
struct TileRows { float4 pv[8]; float4 qv[8]; float wv[8]; };

__device__ __forceinline__ TileIds load_ids(const DecParams& a, int64_t tile, int64_t n_tiles, int lane) {
  TileIds t;
  t.id = 0;
  t.w_e = 0.f;
  const int64_t e = tile * TE + (lane & 31);
  if (tile < n_tiles && e < a.E) {
    t.id = (int)a.ei[(int64_t)(lane >> 5) * a.ld + e];   // lanes 0-31: source, 32-63: target
    if (a.extra && lane < 32) t.w_e = a.extra[e];
  }
  t.row_off = (uint32_t)t.id * ((lane >> 5) ? a.ldq4 : a.ldp4) * 16u;
  return t;
}

__device__ __forceinline__ void issue_rows(const DecParams& a, const TileIds& t, int lane, TileRows& rw) {
  const int r4 = lane >> 4;
  const uint32_t col_off = 16u * (lane & 15);
  const bool has_extra = a.extra != nullptr;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = 4 * i + r4;
    rw.wv[i] = has_extra ? __shfl(t.w_e, row) : 0.f;
    rw.pv[i] = ld_row16(a.p, (uint32_t)__shfl((int)t.row_off, row) + col_off);
    rw.qv[i] = ld_row16(a.q, (uint32_t)__shfl((int)t.row_off, 32 + row) + col_off);
  }
}

__device__ __forceinline__ void commit_rows(const DecParams& a, const TileRows& rw, int lane, const float* cvl,
                                            float* Ht) {
  const int c4 = lane & 15, r4 = lane >> 4;
  const bool has_extra = a.extra != nullptr;
  float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (has_extra) cv = reinterpret_cast<const float4*>(cvl)[c4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = 4 * i + r4;
    float4 h;
    h.x = rw.pv[i].x + rw.qv[i].x;
    h.y = rw.pv[i].y + rw.qv[i].y;
    h.z = rw.pv[i].z + rw.qv[i].z;
    h.w = rw.pv[i].w + rw.qv[i].w;
    if (has_extra) {
      h.x = fmaf(rw.wv[i], cv.x, h.x);
      h.y = fmaf(rw.wv[i], cv.y, h.y);
      h.z = fmaf(rw.wv[i], cv.z, h.z);
      h.w = fmaf(rw.wv[i], cv.w, h.w);
    }
    h.x = relu1(h.x); h.y = relu1(h.y); h.z = relu1(h.z); h.w = relu1(h.w);
    *reinterpret_cast<float4*>(Ht + swz4(row, c4)) = h;
  }
}

__global__ __launch_bounds__(FWD_WAVES * 64) void decoder_fwd_kernel(DecParams a, float* __restrict__ logits,
                                                                    int64_t n_tiles) {
  __shared__ __attribute__((aligned(16))) float lds[64 * RS + 3 * 64 + FWD_WAVES * TE * RS];
  float* Wl = lds;
  float* b2l = lds + 64 * RS;
  float* w3l = b2l + 64;
  float* cvl = w3l + 64;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // SGPR: tile index and its addresses stay scalar
  float* Ht = cvl + 64 + wave * (TE * RS);
  stage_weights(a, Wl, b2l, w3l, cvl, FWD_WAVES * 64);
  __syncthreads();
  const float b3 = a.b3[0];
  const int hh = lane >> 5;
  const int64_t stride = (int64_t)gridDim.x * FWD_WAVES;
  int64_t tile = (int64_t)blockIdx.x * FWD_WAVES + wave;
  TileIds cur = load_ids(a, tile, n_tiles, lane);
  TileIds nxt = load_ids(a, tile + stride, n_tiles, lane);
  TileRows rw;
  issue_rows(a, cur, lane, rw);
  for (; tile < n_tiles; tile += stride) {
    const int64_t ebase = tile * TE;
    commit_rows(a, rw, lane, cvl, Ht);                       // waits for this tile's rows only
    const TileIds nn = load_ids(a, tile + 2 * stride, n_tiles, lane);
    issue_rows(a, nxt, lane, rw);                            // next tile's rows fly during the MFMAs
    nxt = nn;
    wave_lds_sync();
    f32x16 acc[2];
    gemm1(Wl, Ht, lane, acc);
    float part = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int j0 = 32 * b + 8 * qd + 4 * hh;
        const float4 bb = *reinterpret_cast<const float4*>(b2l + j0);
        const float4 ww = *reinterpret_cast<const float4*>(w3l + j0);
        part = fmaf(fmaxf(acc[b][4 * qd + 0] + bb.x, 0.f), ww.x, part);
        part = fmaf(fmaxf(acc[b][4 * qd + 1] + bb.y, 0.f), ww.y, part);
        part = fmaf(fmaxf(acc[b][4 * qd + 2] + bb.z, 0.f), ww.z, part);
        part = fmaf(fmaxf(acc[b][4 * qd + 3] + bb.w, 0.f), ww.w, part);
      }
    part += __shfl_xor(part, 32);
    if (lane < 32 && ebase + lane < a.E) logits[ebase + lane] = part + b3;
    wave_lds_sync();   // the next commit overwrites Ht
  }
}

// FUSED_LOSS: the upstream gradient is not read but produced in place — the tile's logits come out of the
// first product anyway, so BCEWithLogits(pos_weight) and its derivative are evaluated right there
// (pangnn.py:200-207 in one pass): a training step then runs this kernel only, not forward + loss +
// backward, and the forward's 64 MFMAs per tile are not spent twice.
struct LossParams {
  const float* y; const float* pos_weight; float inv_denom; float* logits; 
};

// RUNSUM: when the caller's edge order is sorted by source (every edge list pangnn_amd/construct.py emits
// is), the rows of dL/dh1 that belong to one source are consecutive, so their sum — dL/dP[source] — is
// taken right here per (tile, source) run and written as one 256-byte "part" row; a short contiguous
// segment sum over the parts replaces a 19 GB pass over dL/dh1.  part_off[tile] = index of the tile's first
// part (precomputed with the structure); the number of parts a tile writes is fixed by the edge list, so
// the layout is deterministic.
struct RunSumParams {
  float* part; const int32_t* part_off;
};

template <bool FUSED_LOSS, bool RUNSUM>
__global__ __launch_bounds__(BWD_WAVES * 64) void decoder_bwd_kernel(
    DecParams a, const float* __restrict__ g_logits, LossParams lp, RunSumParams rs, float* __restrict__ g_h1,
    float* __restrict__ slabs, int64_t n_tiles) {
  constexpr int PER_WAVE = TE * RS + 64 * GS + 64;   // Ht | H2t | w_e | g_e
  __shared__ __attribute__((aligned(16))) float lds[64 * RS + 3 * 64 + BWD_WAVES * PER_WAVE];
  float* Wl = lds;
  float* b2l = lds + 64 * RS;
  float* w3l = b2l + 64;
  float* cvl = w3l + 64;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // SGPR: tile index and its addresses stay scalar
  float* Ht = cvl + 64 + wave * PER_WAVE;
  float* Gt = Ht + TE * RS;
  float* wl = Gt + 64 * GS;
  float* gl = wl + 32;
  stage_weights(a, Wl, b2l, w3l, cvl, BWD_WAVES * 64);
  __syncthreads();
  const int hh = lane >> 5, r = lane & 31;

  f32x16 acc3[2][2];   // gW2[j = jr(i,hh)+32bj][k = r+32bk]
#pragma unroll
  for (int x = 0; x < 2; ++x) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc3[x][0][i] = 0.f; acc3[x][1][i] = 0.f; }
  }
  const float w3j[2] = {w3l[r], w3l[r + 32]};
  float gw3p[2] = {0.f, 0.f};   // lane (j = r+32bj, h): partial of gw3[j] over edges e = h mod 2
  float gb2p[2] = {0.f, 0.f};   // same lanes: partial of gb2[j]
  float gcv[2] = {0.f, 0.f};    // lane (k = r+32bp, hh): partial of gcvec[k]
  float gb3p = 0.f;
  float lossp = 0.f;            // FUSED_LOSS: lanes < 32, partial of the (already 1/denom-scaled) loss
  const float b3v = a.b3[0];
  const float pw = (FUSED_LOSS && lp.pos_weight) ? lp.pos_weight[0] : 1.f;

  // One tile.  FULL (every tile but the last of the edge list): all 32 edges exist, so the bounds predicates on
  // loads / stores and the `live` selects fold away; the instruction count of this body is what the kernel's
  // time follows (f32 MFMA and vector instructions share the SIMD's lanes), so the common case carries none.
  auto body = [&](const int64_t tile, auto full_c) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_c)::value;
    const int64_t ebase = tile * TE;
    float w_e;
    int id;
    gather_tile<FULL>(a, ebase, lane, Ht, cvl, w_e, id);
    float g_e = 0.f;
    float y_e = 0.f;
    const bool live = FULL || ebase + r < a.E;
    if (FUSED_LOSS) {
      if (live) y_e = lp.y[ebase + r];
      if (lane < 32) wl[lane] = w_e;
    } else {
      if (live) g_e = g_logits[ebase + r];
      if (lane < 32) { wl[lane] = w_e; gl[lane] = g_e; gb3p += g_e; }
    }
    wave_lds_sync();

    f32x16 acc[2];
    gemm1(Wl, Ht, lane, acc);

    if (FUSED_LOSS) {
      // h2 = relu(C + b2) -> LDS (the weight-gradient product reads it); the tile's logits; the accumulator is
      // overwritten with the masked w3 ([h2 > 0] w3[j]) while w3 is in registers anyway, so that once the
      // edge's dL/dlogit is known G = g_e * (masked w3) is one multiply per element — no second pass over w3,
      // no compare / select after the reduction
      float part = 0.f;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int j0 = 32 * b + 8 * qd + 4 * hh;
          const float4 bb = *reinterpret_cast<const float4*>(b2l + j0);
          const float4 ww = *reinterpret_cast<const float4*>(w3l + j0);
          const float bbv[4] = {bb.x, bb.y, bb.z, bb.w};
          const float wwv[4] = {ww.x, ww.y, ww.z, ww.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int i = 4 * qd + c;
            const float h2 = fmaxf(acc[b][i] + bbv[c], 0.f);
            part = fmaf(h2, wwv[c], part);
            Gt[(j0 + c) * GS + r] = h2;
            acc[b][i] = h2 > 0.f ? wwv[c] : 0.f;
          }
        }
      part += __shfl_xor(part, 32);                 // both halves of the wave now hold edge r's logit
      const float xv = part + b3v;
      const float lw = 1.f + (pw - 1.f) * y_e;
      const float t = expf(-fabsf(xv));             // in (0, 1]
      const float u = 1.f + t;
      float ru = __builtin_amdgcn_rcpf(u);
      ru = ru * (2.f - u * ru);                     // 1 / (1 + t), one Newton step on the hardware reciprocal
      const float sig_neg = xv >= 0.f ? t * ru : ru;                              // sigmoid(-x)
      g_e = live ? ((1.f - y_e) - lw * sig_neg) * lp.inv_denom : 0.f;
      // softplus(-x) = log1p(t) + max(-x, 0);  log1p(t) = log(u) * t / (u - 1) with u = fl(1 + t) (exact u - 1),
      // = t when u == 1
      const float um1 = u - 1.f;
      float rm = __builtin_amdgcn_rcpf(um1);
      rm = rm * (2.f - um1 * rm);
      const float l1p = um1 == 0.f ? t : logf(u) * (t * rm);
      gl[r] = g_e;                                   // the two halves write the same value
      gb3p += g_e;                                   // per-half partials; lane 0's half is the one read out
      lossp += live ? ((1.f - y_e) * xv + lw * (l1p + fmaxf(-xv, 0.f))) * lp.inv_denom : 0.f;   // select, not a branch
      if (hh == 0 && live) lp.logits[ebase + r] = xv;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] *= g_e;   // G[j][e]: A operand of the next product
    } else {
      // G[j][e] = g_e * w3[j] * [h2pre > 0]  (in place in acc; A operand of the next product);
      // h2[j][e] goes to LDS: the weight-gradient product rebuilds G and g_e * h2 from it
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int j0 = 32 * b + 8 * qd + 4 * hh;
          const float4 bb = *reinterpret_cast<const float4*>(b2l + j0);
          const float4 ww = *reinterpret_cast<const float4*>(w3l + j0);
          const float bbv[4] = {bb.x, bb.y, bb.z, bb.w};
          const float wwv[4] = {ww.x, ww.y, ww.z, ww.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int i = 4 * qd + c;
            const float pre = acc[b][i] + bbv[c];
            const bool on = pre > 0.f;
            acc[b][i] = on ? g_e * wwv[c] : 0.f;
            Gt[(j0 + c) * GS + r] = on ? pre : 0.f;
          }
        }
    }
    wave_lds_sync();

    // gH1[e][k] = sum_j G[j][e] W2[j][k]  : A = G from the accumulator registers, B = W2 rows
    f32x16 acc2[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc2[0][i] = 0.f; acc2[1][i] = 0.f; }
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int j = 32 * b + jr(i, hh);
        const float w0 = Wl[swz(j, r)];
        const float w1 = Wl[swz(j, r + 32)];
        acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[b][i], w0, acc2[0], 0, 0, 0);
        acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[b][i], w1, acc2[1], 0, 0, 0);
      }
    // mask by h1 > 0, write dL/dh1pre (kept in acc2 for the run sums), accumulate gcvec.  Full tiles store
    // through one lane base pointer + compile-time offsets: no per-element bounds test or address arithmetic.
    {
      float* gout = g_h1 + (ebase + 4 * hh) * DD + r;
      const bool full = FULL || ebase + TE <= a.E;
#pragma unroll
      for (int bp = 0; bp < 2; ++bp)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int e = jr(i, hh);
          const int k = r + 32 * bp;
          const float hval = Ht[swz(e, k)];
          const float v = hval > 0.f ? acc2[bp][i] : 0.f;
          if (full || ebase + e < a.E) __builtin_nontemporal_store(v, &gout[jr(i, 0) * DD + 32 * bp]);   // 19 GB written once
          acc2[bp][i] = v;
        }
      if (a.extra) {                                 // uniform; only with skip connections
#pragma unroll
        for (int bp = 0; bp < 2; ++bp)
#pragma unroll
          for (int i = 0; i < 16; ++i) gcv[bp] = fmaf(wl[jr(i, hh)], acc2[bp][i], gcv[bp]);
      }
    }

    // gW2[j][k] += sum_e G[j][e] h1[e][k]  : both operands from LDS, reduction over the tile's edges
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int e = 2 * s + hh;
      const float ge = gl[e];
      const float x0 = Gt[r * GS + e];           // h2[j][e]
      const float x1 = Gt[(r + 32) * GS + e];
      const float a0 = x0 > 0.f ? ge * w3j[0] : 0.f;
      const float a1 = x1 > 0.f ? ge * w3j[1] : 0.f;
      gw3p[0] = fmaf(ge, x0, gw3p[0]);
      gw3p[1] = fmaf(ge, x1, gw3p[1]);
      gb2p[0] += a0;
      gb2p[1] += a1;
      const float h0 = Ht[swz(e, r)];
      const float h1v = Ht[swz(e, r + 32)];
      acc3[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, h0, acc3[0][0], 0, 0, 0);
      acc3[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, h1v, acc3[0][1], 0, 0, 0);
      acc3[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, h0, acc3[1][0], 0, 0, 0);
      acc3[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, h1v, acc3[1][1], 0, 0, 0);
    }
    if (RUNSUM) {
      // close a part at the last edge of every source run of the tile
      const int id_nxt = __shfl(id, (lane + 1) & 63);
      const bool ok = lane < 32 && (FULL || ebase + lane < a.E);
      const bool ok_nxt = lane < 31 && (FULL || ebase + lane + 1 < a.E);
      const unsigned long long mask = __ballot(ok && (!ok_nxt || id != id_nxt));
      const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(mask & 0xffffffffull));
      int64_t pidx = rs.part_off[tile];
      if ((m & (m - 1u)) == 0u) {
        // one run covers the tile (the common case at degree >> 32): column sums straight from registers
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0 += acc2[0][i]; s1 += acc2[1][i]; }
        s0 += __shfl_xor(s0, 32);
        s1 += __shfl_xor(s1, 32);
        if (hh == 0) {
          rs.part[pidx * DD + r] = s0;
          rs.part[pidx * DD + 32 + r] = s1;
        }
      } else if (__builtin_popcount(m) == 2) {
        // two runs: both column sums from the registers, rows selected by their position relative to the boundary
        const int bnd = __builtin_ctz(m);            // last edge of the first run
        float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const bool first = jr(i, hh) <= bnd;
          a0 += first ? acc2[0][i] : 0.f;
          b0 += first ? 0.f : acc2[0][i];
          a1 += first ? acc2[1][i] : 0.f;
          b1 += first ? 0.f : acc2[1][i];
        }
        a0 += __shfl_xor(a0, 32);
        a1 += __shfl_xor(a1, 32);
        b0 += __shfl_xor(b0, 32);
        b1 += __shfl_xor(b1, 32);
        if (hh == 0) {
          rs.part[pidx * DD + r] = a0;
          rs.part[pidx * DD + 32 + r] = a1;
          rs.part[(pidx + 1) * DD + r] = b0;
          rs.part[(pidx + 1) * DD + 32 + r] = b1;
        }
      } else {
        // three or more runs: dL/dh1 tile -> LDS (the h1 image is no longer needed); every lane owns one of the
        // 64 columns and walks the 32 rows
        wave_lds_sync();
#pragma unroll
        for (int bp = 0; bp < 2; ++bp)
#pragma unroll
          for (int i = 0; i < 16; ++i) Ht[swz(jr(i, hh), r + 32 * bp)] = acc2[bp][i];
        wave_lds_sync();
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < TE; ++e) {
          sum += Ht[swz(e, lane)];
          if ((m >> e) & 1u) {
            rs.part[pidx * DD + lane] = sum;
            sum = 0.f;
            ++pidx;
          }
        }
      }
    }
    wave_lds_sync();   // next tile overwrites Ht / Gt / wl
  };
  const int64_t n_full = a.E / TE, stride = (int64_t)gridDim.x * BWD_WAVES;
  int64_t tile = (int64_t)blockIdx.x * BWD_WAVES + wave;
  for (; tile < n_full; tile += stride) body(tile, std::true_type{});
  if (tile < n_tiles) body(tile, std::false_type{});     // tile == n_full: the partial tile, one wave's turn

  // ---- fold the per-lane partials, then reduce the workgroup's waves in wave order
  gw3p[0] += __shfl_xor(gw3p[0], 32);
  gw3p[1] += __shfl_xor(gw3p[1], 32);
  gb2p[0] += __shfl_xor(gb2p[0], 32);
  gb2p[1] += __shfl_xor(gb2p[1], 32);
  gcv[0] += __shfl_xor(gcv[0], 32);
  gcv[1] += __shfl_xor(gcv[1], 32);
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) {
    gb3p += __shfl_xor(gb3p, off);
    lossp += __shfl_xor(lossp, off);
  }

  __syncthreads();
  float* red = cvl + 64;   // reuse the tile area: SLAB floats
  for (int w = 0; w < BWD_WAVES; ++w) {
    if (wave == w) {
      const bool first = (w == 0);
#pragma unroll
      for (int bj = 0; bj < 2; ++bj)
#pragma unroll
        for (int bk = 0; bk < 2; ++bk)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int idx = (32 * bj + jr(i, hh)) * 64 + r + 32 * bk;
            red[idx] = (first ? 0.f : red[idx]) + acc3[bj][bk][i];
          }
      if (hh == 0) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          red[4096 + r + 32 * x] = (first ? 0.f : red[4096 + r + 32 * x]) + gb2p[x];
          red[4096 + 64 + r + 32 * x] = (first ? 0.f : red[4096 + 64 + r + 32 * x]) + gw3p[x];
          red[4096 + 128 + r + 32 * x] = (first ? 0.f : red[4096 + 128 + r + 32 * x]) + gcv[x];
        }
      }
      if (lane == 0) {
        red[4096 + 192] = (first ? 0.f : red[4096 + 192]) + gb3p;
        red[4096 + 193] = (first ? 0.f : red[4096 + 193]) + lossp;
      }
    }
    __syncthreads();
  }
  float* slab = slabs + (int64_t)blockIdx.x * SLAB;
  for (int i = threadIdx.x; i < 4096 + 194; i += BWD_WAVES * 64) slab[i] = red[i];
}

// out[i] = sum over workgroup slabs in index order (fixed => reproducible)
__global__ __launch_bounds__(kSumThreads) void decoder_reduce_kernel(const float* __restrict__ slabs, int n_slabs,
                                                                float* __restrict__ g_w2,
                                                                float* __restrict__ g_b2,
                                                                float* __restrict__ g_w3,
                                                                float* __restrict__ g_cvec,
                                                                float* __restrict__ g_b3,
                                                                float* __restrict__ loss) {
  const int i = blockIdx.x * kWave + (threadIdx.x & (kWave - 1));
  const float s = ordered_parts_sum(slabs, n_slabs, SLAB, i, 4096 + 194);
  if (threadIdx.x >= kWave || i >= 4096 + 194) return;
  if (i < 4096) g_w2[i] = s;
  else if (i < 4096 + 64) { if (g_b2) g_b2[i - 4096] = s; }
  else if (i < 4096 + 128) g_w3[i - 4096 - 64] = s;
  else if (i < 4096 + 192) { if (g_cvec) g_cvec[i - 4096 - 128] = s; }
  else if (i == 4096 + 192) g_b3[0] = s;
  else if (loss) loss[0] = s;
}

// decoder16.hip: inference form of the bf16 matrix-pipe mode
int launch_decoder_infer16(const float* p, int64_t ldp, const float* q, int64_t ldq, const int64_t* edge_index, int64_t ld,
                           int64_t num_edges, const float* extra, const float* cvec, const float* w2, const float* b2,
                           const float* w3, const float* b3, float* logits, hipStream_t s);

// for decoder16.hip: the same finishing reduction over its slabs (same layout)
int launch_decoder_reduce(const float* slabs, int n_slabs, float* g_w2, float* g_b2, float* g_w3, float* g_cvec,
                          float* g_b3, float* loss, hipStream_t s) {
  hipLaunchKernelGGL(decoder_reduce_kernel, dim3((4096 + 194 + kWave - 1) / kWave), dim3(kSumThreads), 0, s, slabs,
                     n_slabs, g_w2, g_b2, g_w3, g_cvec, g_b3, loss);
  PG_CHECK_LAUNCH("decoder_reduce");
  return 0;
}

static int grid_cus() {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  return cus;
}

static int check_common(const char* who, const float* p, const float* q, int64_t ldp, int64_t ldq,
                        int64_t num_nodes,
                        const int64_t* ei, int64_t ld, int64_t E, const float* extra,
                        const float* cvec, const float* w2, const float* b2, const float* w3,
                        const float* b3, int32_t D) {
  PG_CHECK_ARG(D == DD, PANGNN_E_BADARG, "%s: fused decoder is built for node_dim 64, got %d", who, (int)D);
  PG_CHECK_ARG(E >= 0 && ld >= E && num_nodes >= 0, PANGNN_E_BADARG, "%s: bad size", who);
  PG_CHECK_ARG(ldp >= DD && ldq >= DD && ldp % 4 == 0 && ldq % 4 == 0, PANGNN_E_BADARG,
               "%s: ldp / ldq must be multiples of 4 and >= 64", who);
  PG_CHECK_ARG((double)num_nodes * (double)(ldp > ldq ? ldp : ldq) * 4.0 < 4294967296.0, PANGNN_E_TOOLARGE,
               "%s: node tables must stay under 4 GiB (32-bit gather offsets)", who);
  if (E == 0) return 0;
  PG_CHECK_ARG(p && q && ei && w2 && b2 && w3 && b3 && (!extra || cvec), PANGNN_E_BADARG,
               "%s: null pointer", who);
  PG_CHECK_ARG(aligned16(p) && aligned16(q) && aligned16(w2), PANGNN_E_ALIGN,
               "%s: p / q / w2 must be 16-byte aligned", who);
  return 0;
}

}  // namespace pangnn

using namespace pangnn;

extern "C" int pangnn_decoder_mlp_infer_f32(const float* p, int64_t ldp, const float* q, int64_t ldq,
                                            int64_t num_nodes, const int64_t* edge_index, int64_t ld,
                                            int64_t num_edges, const float* extra, const float* cvec,
                                            const float* w2, const float* b2, const float* w3, const float* b3,
                                            int32_t D, float* logits, int32_t precision, pangnn_stream_t stream) {
  int rc = check_common("pangnn_decoder_mlp_fwd_f32", p, q, ldp, ldq, num_nodes, edge_index, ld, num_edges, extra,
                        cvec, w2, b2, w3, b3, D);
  if (rc) return rc;
  PG_CHECK_ARG(precision == 0 || precision == 1, PANGNN_E_BADARG, "pangnn_decoder_mlp_infer_f32: precision must be 0 or 1");
  if (num_edges == 0) return 0;
  PG_CHECK_ARG(logits, PANGNN_E_BADARG, "pangnn_decoder_mlp_fwd_f32: null logits");
  if (precision)   // bf16 matrix pipe, split operands: decoder16.hip (the first product of the training kernel)
    return launch_decoder_infer16(p, ldp, q, ldq, edge_index, ld, num_edges, extra, cvec, w2, b2, w3, b3, logits,
                                  (hipStream_t)stream);
  const int64_t n_tiles = (num_edges + TE - 1) / TE;
  int64_t grid = (n_tiles + FWD_WAVES - 1) / FWD_WAVES;
  const int cus = grid_cus();
  if (grid > cus) grid = cus;
  DecParams a{p, q, (uint32_t)(ldp / 4), (uint32_t)(ldq / 4), edge_index, ld, num_edges, extra, cvec, w2, b2, w3, b3};
  hipLaunchKernelGGL(decoder_fwd_kernel, dim3((unsigned)grid), dim3(FWD_WAVES * 64), 0, (hipStream_t)stream, a, logits,
                     n_tiles);
  PG_CHECK_LAUNCH("pangnn_decoder_mlp_fwd_f32");
  return 0;
}

extern "C" int pangnn_decoder_mlp_fwd_f32(const float* p, int64_t ldp, const float* q, int64_t ldq,
                                          int64_t num_nodes, const int64_t* edge_index, int64_t ld, int64_t num_edges,
                                          const float* extra, const float* cvec, const float* w2,
                                          const float* b2, const float* w3, const float* b3,
                                          int32_t D, float* logits, pangnn_stream_t stream) {
  return pangnn_decoder_mlp_infer_f32(p, ldp, q, ldq, num_nodes, edge_index, ld, num_edges, extra, cvec, w2, b2, w3, b3,
                                      D, logits, 0, stream);
}

extern "C" size_t pangnn_decoder_mlp_bwd_workspace_bytes(int64_t num_edges) {
  (void)num_edges;
  return (size_t)grid_cus() * SLAB * sizeof(float);
}

static int launch_bwd(const char* who, const DecParams& a, int64_t num_nodes, int32_t D, const float* g_logits,
                      const LossParams* lp, int precision, float* part_buf, const int32_t* part_off, float* g_h1, float* g_w2, float* g_b2, float* g_w3, float* g_b3,
                      float* g_cvec, float* loss, void* workspace, size_t workspace_bytes, hipStream_t s) {
  PG_CHECK_ARG(g_w2 && g_b2 && g_w3 && g_b3, PANGNN_E_BADARG, "%s: null gradient output", who);
  const int64_t num_edges = a.E;
  PG_CHECK_ARG(precision == 0, PANGNN_E_BADARG,
               "%s: only precision 0 (f32 MFMA) here; the bf16 matrix-pipe mode is pangnn_decoder_train_f32 + "
               "pangnn_decoder_dgrad_f32", who);
  const int waves = BWD_WAVES;
  const int64_t n_tiles = (num_edges + TE - 1) / TE;
  int64_t grid = (n_tiles + waves - 1) / waves;
  const int cus = grid_cus();
  if (grid > cus) grid = cus;
  if (grid < 1) grid = 1;
  PG_CHECK_ARG(workspace && workspace_bytes >= (size_t)grid * SLAB * sizeof(float), PANGNN_E_WORKSPACE,
               "%s: workspace too small", who);
  PG_CHECK_ARG(num_edges == 0 || g_h1, PANGNN_E_BADARG, "%s: null g_h1", who);
  if (num_edges == 0) {
    hipError_t e = hipMemsetAsync(workspace, 0, (size_t)grid * SLAB * sizeof(float), s);
    PG_CHECK_ARG(e == hipSuccess, (int)e, "%s: memset failed", who);
  } else {
    PG_CHECK_ARG((part_buf == nullptr) == (part_off == nullptr), PANGNN_E_BADARG,
                 "%s: part_buf and part_off go together", who);
    const LossParams none{nullptr, nullptr, 0.f, nullptr};
    const LossParams l = lp ? *lp : none;
    const RunSumParams rs{part_buf, part_off};
    const dim3 g((unsigned)grid), b(waves * 64);
    float* ws = static_cast<float*>(workspace);
    if (lp && part_buf) hipLaunchKernelGGL((decoder_bwd_kernel<true, true>), g, b, 0, s, a, g_logits, l, rs, g_h1, ws, n_tiles);
    else if (lp) hipLaunchKernelGGL((decoder_bwd_kernel<true, false>), g, b, 0, s, a, g_logits, l, rs, g_h1, ws, n_tiles);
    else if (part_buf) hipLaunchKernelGGL((decoder_bwd_kernel<false, true>), g, b, 0, s, a, g_logits, l, rs, g_h1, ws, n_tiles);
    else hipLaunchKernelGGL((decoder_bwd_kernel<false, false>), g, b, 0, s, a, g_logits, l, rs, g_h1, ws, n_tiles);
    PG_CHECK_LAUNCH(who);
  }
  hipLaunchKernelGGL(decoder_reduce_kernel, dim3((4096 + 194 + kWave - 1) / kWave), dim3(kSumThreads), 0, s,
                     static_cast<const float*>(workspace), (int)grid, g_w2, g_b2, g_w3, g_cvec, g_b3, loss);
  PG_CHECK_LAUNCH(who);
  return 0;
}

extern "C" int pangnn_decoder_mlp_bwd_f32(const float* p, int64_t ldp, const float* q, int64_t ldq,
                                          int64_t num_nodes, const int64_t* edge_index, int64_t ld, int64_t num_edges,
                                          const float* extra, const float* cvec, const float* w2,
                                          const float* b2, const float* w3, const float* b3, int32_t D,
                                          const float* g_logits, float* g_h1, float* g_w2, float* g_b2,
                                          float* g_w3, float* g_b3, float* g_cvec, float* part_buf,
                                          const int32_t* part_off, int32_t precision, void* workspace,
                                          size_t workspace_bytes, pangnn_stream_t stream) {
  int rc = check_common("pangnn_decoder_mlp_bwd_f32", p, q, ldp, ldq, num_nodes, edge_index, ld, num_edges, extra,
                        cvec, w2, b2, w3, b3, D);
  if (rc) return rc;
  PG_CHECK_ARG(num_edges == 0 || g_logits, PANGNN_E_BADARG, "pangnn_decoder_mlp_bwd_f32: null g_logits");
  DecParams a{p, q, (uint32_t)(ldp / 4), (uint32_t)(ldq / 4), edge_index, ld, num_edges, extra, cvec, w2, b2, w3, b3};
  return launch_bwd("pangnn_decoder_mlp_bwd_f32", a, num_nodes, D, g_logits, nullptr, precision, part_buf, part_off, g_h1, g_w2, g_b2, g_w3, g_b3,
                    g_cvec, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int pangnn_decoder_mlp_loss_f32(const float* p, int64_t ldp, const float* q, int64_t ldq,
                                           int64_t num_nodes, const int64_t* edge_index, int64_t ld,
                                           int64_t num_edges, const float* extra, const float* cvec,
                                           const float* w2, const float* b2, const float* w3, const float* b3,
                                           int32_t D, const float* y, const float* pos_weight, int64_t denom,
                                           float* logits, float* loss, float* g_h1, float* g_w2, float* g_b2,
                                           float* g_w3, float* g_b3, float* g_cvec, float* part_buf,
                                           const int32_t* part_off, int32_t precision, void* workspace,
                                           size_t workspace_bytes, pangnn_stream_t stream) {
  int rc = check_common("pangnn_decoder_mlp_loss_f32", p, q, ldp, ldq, num_nodes, edge_index, ld, num_edges, extra,
                        cvec, w2, b2, w3, b3, D);
  if (rc) return rc;
  PG_CHECK_ARG(denom > 0 && loss && (num_edges == 0 || (y && logits)), PANGNN_E_BADARG,
               "pangnn_decoder_mlp_loss_f32: bad denom / null pointer");
  DecParams a{p, q, (uint32_t)(ldp / 4), (uint32_t)(ldq / 4), edge_index, ld, num_edges, extra, cvec, w2, b2, w3, b3};
  LossParams lp{y, pos_weight, 1.0f / (float)denom, logits};
  return launch_bwd("pangnn_decoder_mlp_loss_f32", a, num_nodes, D, nullptr, &lp, precision, part_buf, part_off, g_h1, g_w2, g_b2, g_w3, g_b3,
                    g_cvec, loss, workspace, workspace_bytes, (hipStream_t)stream);
}
