// propagate (k5/k6/k5^T): CSR gather - scale - per-destination reduce, fp32, gfx950.
//
// Layout: one 64-lane wavefront owns one destination row.  A source row of F floats is covered by
// G = F/4 lanes with one 16-byte load each (global_load_dwordx4), so a wave-instruction gathers
// EPS = 64/G different source rows = 1 KiB of row data.  The row's (idx, val) list is read 64 entries
// at a time with one coalesced load per array and handed to the G-lane groups by ds_bpermute
// (cross-lane, no memory); the next 64 entries are prefetched while the current ones are consumed.
// U independent gathers per lane are kept in flight (U KiB per wave).  The EPS partial sums are
// combined with log2(EPS) xor-shuffles at the end of the row; no atomics, fixed summation order.
#include "common.h"

namespace pangnn {

// XF: storage format of the gathered rows — 0 float32, 1 bfloat16, 2 float16 (PANGNN_DTYPE_*; the 2-byte formats: 8 bytes per
// lane instead of 16; SURVEY.md §8b `X f32/bf16`, the storage format of config 5, and `--mixed_precision fp16`).  Weights,
// accumulation and the result stay fp32 — what PyG's propagate computes under autocast (a 16-bit x_j times an fp32 edge
// weight promotes to fp32, scatter-add into an fp32 output).
__device__ __forceinline__ float4 row_piece(const char* p, int xf) {
  if (!xf) return *reinterpret_cast<const float4*>(p);
  return rows16_to_f32(*reinterpret_cast<const uint2*>(p), xf);
}

template <int F, int U, bool BIG, int XF = 0>
__global__ __launch_bounds__(kBlock) void spmm_row_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ idx,
    const float* __restrict__ val, const float* __restrict__ x, int64_t ldx,
    const float* __restrict__ bias, float* __restrict__ out, int64_t ldo, int64_t n_rows,
    int accumulate) {
  constexpr int G = F / 4;        // lanes per source row
  constexpr int EPS = kWave / G;  // source rows per wave-instruction
  static_assert(G >= 1 && G <= 64 && (G & (G - 1)) == 0, "F must be 4 * power of two, <= 256");
  static_assert(EPS * U <= 64, "unroll too deep for this feature width");

  const int lane = threadIdx.x & (kWave - 1);
  const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // SGPR
  if (row >= n_rows) return;  // whole wave exits together (row is wave-uniform)
  const int sub = lane / G;   // which of the EPS rows of a step this lane helps with
  const int fl = lane % G;    // which float4 of that row

  const int64_t beg = rowptr[row];
  const int64_t end = rowptr[row + 1];

  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int ES = XF ? 2 : 4;                      // bytes per stored element
  const char* xbase = reinterpret_cast<const char*>(x) + fl * 4 * ES;
  const uint32_t ldx_b32 = (uint32_t)(ldx * ES);

  // prefetch first block of 64 list entries
  int c_nxt = 0;
  float v_nxt = 0.f;
  if (beg + lane < end) {
    // the entry lists are streamed once per launch: non-temporal, so that they do not displace gathered rows
    c_nxt = idx ? __builtin_nontemporal_load(idx + beg + lane) : (int)(beg + lane);     // idx == NULL: identity
    v_nxt = val ? __builtin_nontemporal_load(val + beg + lane) : 1.f;
  }
  for (int64_t e0 = beg; e0 < end; e0 += kWave) {
    const int c_cur = c_nxt;
    const float v_cur = v_nxt;
    const int64_t rem = end - e0;
    const int cnt = rem < kWave ? (int)rem : kWave;
    {
      const int64_t en = e0 + kWave + lane;
      c_nxt = 0;
      v_nxt = 0.f;
      if (en < end) {
        c_nxt = idx ? __builtin_nontemporal_load(idx + en) : (int)en;
        v_nxt = val ? __builtin_nontemporal_load(val + en) : 1.f;
      }
    }
    for (int s = 0; s < cnt; s += EPS * U) {
      float4 xv[U];
      float vv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = s + u * EPS + sub;  // < 64 by construction
        const int c = __shfl(c_cur, k);
        vv[u] = __shfl(v_cur, k);
        if (k < cnt) {
          if (BIG && !XF) {
            // > 4 GiB tables are the per-edge gradient rows of the decoder: each row is read exactly once
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xbase + (int64_t)c * ldx * 4));
            xv[u] = make_float4(t[0], t[1], t[2], t[3]);
          } else if (BIG) {
            xv[u] = row_piece(xbase + (int64_t)c * ldx * ES, XF);
          } else {
            xv[u] = row_piece(xbase + (uint32_t)c * ldx_b32, XF);
          }
        } else {
          xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          vv[u] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        acc.x = fmaf(vv[u], xv[u].x, acc.x);
        acc.y = fmaf(vv[u], xv[u].y, acc.y);
        acc.z = fmaf(vv[u], xv[u].z, acc.z);
        acc.w = fmaf(vv[u], xv[u].w, acc.w);
      }
    }
  }
  // combine the EPS partial rows (fixed tree order)
#pragma unroll
  for (int off = 32; off >= G; off >>= 1) {
    acc.x += __shfl_xor(acc.x, off);
    acc.y += __shfl_xor(acc.y, off);
    acc.z += __shfl_xor(acc.z, off);
    acc.w += __shfl_xor(acc.w, off);
  }
  if (lane < G) {
    if (bias) {
      const float4 b = reinterpret_cast<const float4*>(bias)[fl];
      acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    }
    float4* o = reinterpret_cast<float4*>(out + row * ldo) + fl;
    if (accumulate) {
      const float4 p = *o;
      acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
    }
    *o = acc;
  }
}

// Thin rows (the positional-neighbour graph has 3 entries per row, the per-source run parts of the decoder
// 2-3): one wave per row would use 3 of the 64 / G row slots of a wave-instruction and retire after one step.
// Here a group of G = F/4 lanes owns a row (64/G rows per wave) and walks its entries serially; the index and
// weight loads of a group hit one address (broadcast), the row gather is the same 16-byte-per-lane access.
template <int F, bool BIG>
__global__ __launch_bounds__(kBlock) void spmm_thin_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ idx, const float* __restrict__ val,
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ bias, float* __restrict__ out,
    int64_t ldo, int64_t n_rows, int accumulate) {
  constexpr int G = F / 4;
  constexpr int RPB = kBlock / G;       // rows per workgroup
  const int fl = threadIdx.x % G;
  const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / G;
  if (row >= n_rows) return;
  const int64_t beg = rowptr[row], end = rowptr[row + 1];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const char* xbase = reinterpret_cast<const char*>(x) + fl * 16;
  const uint32_t ldx_b32 = (uint32_t)(ldx * 4);
  for (int64_t e = beg; e < end; ++e) {
    const int c = idx ? idx[e] : (int)e;
    const float v = val ? val[e] : 1.f;
    const float4 xv = BIG ? *reinterpret_cast<const float4*>(xbase + (int64_t)c * ldx * 4)
                          : *reinterpret_cast<const float4*>(xbase + (uint32_t)c * ldx_b32);
    acc.x = fmaf(v, xv.x, acc.x);
    acc.y = fmaf(v, xv.y, acc.y);
    acc.z = fmaf(v, xv.z, acc.z);
    acc.w = fmaf(v, xv.w, acc.w);
  }
  if (bias) {
    const float4 b = reinterpret_cast<const float4*>(bias)[fl];
    acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
  }
  float4* o = reinterpret_cast<float4*>(out + row * ldo) + fl;
  if (accumulate) {
    const float4 p = *o;
    acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
  }
  *o = acc;
}

// Any feature width: lanes stride over features, edges serial.  Correctness path for odd F.
__global__ __launch_bounds__(kBlock) void spmm_row_generic_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ idx,
    const float* __restrict__ val, const float* __restrict__ x, int64_t ldx,
    const float* __restrict__ bias, float* __restrict__ out, int64_t ldo, int64_t n_rows, int F,
    int accumulate) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // SGPR
  if (row >= n_rows) return;
  const int64_t beg = rowptr[row], end = rowptr[row + 1];
  for (int f = lane; f < F; f += kWave) {
    float acc = 0.f;
    for (int64_t e = beg; e < end; ++e) {
      const float v = val ? val[e] : 1.f;
      acc = fmaf(v, x[(int64_t)(idx ? idx[e] : (int)e) * ldx + f], acc);
    }
    if (bias) acc += bias[f];
    if (accumulate) acc += out[row * ldo + f];
    out[row * ldo + f] = acc;
  }
}

template <int F, int U, int XF>
static int launch_spmm_16(const int64_t* rowptr, const int32_t* idx, const float* val, const void* x,
                          int64_t ldx, int64_t n_src_rows, const float* bias, float* out, int64_t ldo,
                          int64_t n_rows, int accumulate, hipStream_t s) {
  const int64_t blocks = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
  const bool big = (double)n_src_rows * (double)ldx * 2.0 >= 4294967296.0;
  const float* xf = static_cast<const float*>(x);
  if (big)
    hipLaunchKernelGGL((spmm_row_kernel<F, U, true, XF>), dim3((unsigned)blocks), dim3(kBlock), 0, s,
                       rowptr, idx, val, xf, ldx, bias, out, ldo, n_rows, accumulate);
  else
    hipLaunchKernelGGL((spmm_row_kernel<F, U, false, XF>), dim3((unsigned)blocks), dim3(kBlock), 0, s,
                       rowptr, idx, val, xf, ldx, bias, out, ldo, n_rows, accumulate);
  PG_CHECK_LAUNCH("pangnn_spmm_csr_16");
  return 0;
}

template <int F, int U>
static int launch_spmm(const int64_t* rowptr, const int32_t* idx, const float* val, const float* x,
                       int64_t ldx, int64_t n_src_rows, const float* bias, float* out, int64_t ldo,
                       int64_t n_rows, int accumulate, hipStream_t s) {
  const int64_t blocks = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
  const bool big = (double)n_src_rows * (double)ldx * 4.0 >= 4294967296.0;
  if (big)
    hipLaunchKernelGGL((spmm_row_kernel<F, U, true>), dim3((unsigned)blocks), dim3(kBlock), 0, s,
                       rowptr, idx, val, x, ldx, bias, out, ldo, n_rows, accumulate);
  else
    hipLaunchKernelGGL((spmm_row_kernel<F, U, false>), dim3((unsigned)blocks), dim3(kBlock), 0, s,
                       rowptr, idx, val, x, ldx, bias, out, ldo, n_rows, accumulate);
  PG_CHECK_LAUNCH("pangnn_spmm_csr_f32");
  return 0;
}

template <int F>
static int launch_thin(const int64_t* rowptr, const int32_t* idx, const float* val, const float* x,
                       int64_t ldx, int64_t n_src_rows, const float* bias, float* out, int64_t ldo,
                       int64_t n_rows, int accumulate, hipStream_t s) {
  constexpr int RPB = kBlock / (F / 4);
  const int64_t blocks = (n_rows + RPB - 1) / RPB;
  const bool big = (double)n_src_rows * (double)ldx * 4.0 >= 4294967296.0;
  if (big)
    hipLaunchKernelGGL((spmm_thin_kernel<F, true>), dim3((unsigned)blocks), dim3(kBlock), 0, s,
                       rowptr, idx, val, x, ldx, bias, out, ldo, n_rows, accumulate);
  else
    hipLaunchKernelGGL((spmm_thin_kernel<F, false>), dim3((unsigned)blocks), dim3(kBlock), 0, s,
                       rowptr, idx, val, x, ldx, bias, out, ldo, n_rows, accumulate);
  PG_CHECK_LAUNCH("pangnn_spmm_csr_f32(thin)");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Propagate over the positional-neighbour graph of a whole genome set (src/dataset.py:356-361: edges (i, j) for
// j in [i - k, i + k] ∩ [0, N), self loop included, unit weights): a BAND matrix — row t gathers rows t - k .. t + k.
//   out[t] = bias + sum_{s = t-k .. t+k} (dis[s] * dis[t]) x[s]        dis = deg^-1/2 (GcnNorm.deg_inv_sqrt)
// No index or weight arrays, rows read as a contiguous window: the same sums in the same order as the generic
// kernel over that edge list (bit-identical), at streaming speed.  The band is symmetric, so the transposed
// propagate is this kernel too; COLSUM also emits per-workgroup column sums of x (the bias gradient when x is the
// upstream gradient), finished in fixed order by band_colsum_finish_kernel.
// ------------------------------------------------------------------------------------------------------------------
template <int F, typename TX, bool COLSUM>
__global__ __launch_bounds__(kBlock) void band_propagate_kernel(const TX* __restrict__ x, int64_t ldx,
                                                                const float* __restrict__ dis,
                                                                const float* __restrict__ bias, float* __restrict__ out,
                                                                int64_t ldo, int64_t n, int k,
                                                                float* __restrict__ colsum_partial) {
  constexpr int G = F / 4;              // lanes per row (4 adjacent columns each)
  constexpr int RPB = kBlock / G;       // rows per workgroup pass
  __shared__ float red[COLSUM ? kBlock * 4 : 4];
  const int fl = threadIdx.x % G, rg = threadIdx.x / G;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) b = reinterpret_cast<const float4*>(bias)[fl];
  for (int64_t t = (int64_t)blockIdx.x * RPB + rg; t < n; t += (int64_t)gridDim.x * RPB) {
    const float dt = dis[t];
    const int64_t lo = t - k < 0 ? 0 : t - k, hi = t + k >= n ? n - 1 : t + k;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t sidx = lo; sidx <= hi; ++sidx) {
      float4 xv;
      if constexpr (sizeof(TX) == 4) {
        xv = *reinterpret_cast<const float4*>(x + sidx * ldx + 4 * fl);
      } else {       // TX = unsigned short: bfloat16 bits; TX = _Float16: IEEE half
        xv = rows16_to_f32(*reinterpret_cast<const uint2*>(x + sidx * ldx + 4 * fl), RowFmt<TX>::value);
      }
      const float v = dis[sidx] * dt;                       // gcn_norm: (dis[src] * 1) * dis[dst]
      acc.x = fmaf(v, xv.x, acc.x);
      acc.y = fmaf(v, xv.y, acc.y);
      acc.z = fmaf(v, xv.z, acc.z);
      acc.w = fmaf(v, xv.w, acc.w);
      if (COLSUM && sidx == t) { cs.x += xv.x; cs.y += xv.y; cs.z += xv.z; cs.w += xv.w; }
    }
    acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    *reinterpret_cast<float4*>(out + t * ldo + 4 * fl) = acc;
  }
  if (COLSUM) {
    red[rg * F + 4 * fl + 0] = cs.x; red[rg * F + 4 * fl + 1] = cs.y;
    red[rg * F + 4 * fl + 2] = cs.z; red[rg * F + 4 * fl + 3] = cs.w;
    __syncthreads();
    if (threadIdx.x < F) {
      float s = 0.f;
      for (int q = 0; q < RPB; ++q) s += red[q * F + threadIdx.x];
      colsum_partial[(int64_t)blockIdx.x * F + threadIdx.x] = s;
    }
  }
}

__global__ __launch_bounds__(kSumThreads) void band_colsum_finish_kernel(const float* __restrict__ partial, int nblocks,
                                                                         int F, float* __restrict__ out) {
  const int i = blockIdx.x * kWave + (threadIdx.x & (kWave - 1));
  const float t = ordered_parts_sum(partial, nblocks, F, i, F);
  if (threadIdx.x < kWave && i < F) out[i] = t;
}

constexpr int kBandBlocks = 2048;

template <int F>
static int launch_band(const void* x, int x_fmt, int64_t ldx, const float* dis, const float* bias, float* out, int64_t ldo,
                       int64_t n, int k, float* colsum, float* ws, hipStream_t s) {
  constexpr int RPB = kBlock / (F / 4);
  int64_t blocks = (n + RPB - 1) / RPB;
  if (blocks > kBandBlocks) blocks = kBandBlocks;
  const dim3 g((unsigned)blocks), b(kBlock);
#define PG_BAND(T, CS) hipLaunchKernelGGL((band_propagate_kernel<F, T, CS>), g, b, 0, s, static_cast<const T*>(x), ldx, dis, bias, out, ldo, n, k, ws)
  if (x_fmt == PANGNN_DTYPE_F16) { if (colsum) PG_BAND(_Float16, true); else PG_BAND(_Float16, false); }
  else if (x_fmt) { if (colsum) PG_BAND(unsigned short, true); else PG_BAND(unsigned short, false); }
  else { if (colsum) PG_BAND(float, true); else PG_BAND(float, false); }
#undef PG_BAND
  PG_CHECK_LAUNCH("pangnn_band_propagate");
  if (colsum) {
    hipLaunchKernelGGL(band_colsum_finish_kernel, dim3((F + kWave - 1) / kWave), dim3(kSumThreads), 0, s, ws, (int)blocks, F,
                       colsum);
    PG_CHECK_LAUNCH("pangnn_band_propagate(colsum)");
  }
  return 0;
}

}  // namespace pangnn

using namespace pangnn;

extern "C" int pangnn_spmm_csr_f32(const int64_t* rowptr, const int32_t* idx, const float* val,
                                   const float* x, int64_t ldx, int64_t n_src_rows,
                                   const float* bias, float* out, int64_t ldo, int64_t n_rows,
                                   int64_t nnz, int32_t F, int accumulate, pangnn_stream_t stream) {
  PG_CHECK_ARG(n_rows >= 0 && n_src_rows >= 0 && F > 0, PANGNN_E_BADARG,
               "pangnn_spmm_csr_f32: negative size (n_rows=%lld n_src_rows=%lld F=%d)",
               (long long)n_rows, (long long)n_src_rows, (int)F);
  if (n_rows == 0) return 0;
  PG_CHECK_ARG(rowptr && out && (x || n_src_rows == 0), PANGNN_E_BADARG,
               "pangnn_spmm_csr_f32: null pointer");
  PG_CHECK_ARG(ldx >= F && ldo >= F, PANGNN_E_BADARG,
               "pangnn_spmm_csr_f32: leading dimension smaller than F");
  PG_CHECK_ARG((n_rows + kWavesPerBlock - 1) / kWavesPerBlock < 2147483647LL, PANGNN_E_TOOLARGE,
               "pangnn_spmm_csr_f32: too many rows for one launch");
  hipStream_t s = (hipStream_t)stream;
  const bool vec_ok = aligned16(x) && aligned16(out) && (!bias || aligned16(bias)) &&
                      (ldx % 4 == 0) && (ldo % 4 == 0);
  // Rows that average fewer than 8 entries (the positional-neighbour graph has 3 per row) go to the kernel
  // that packs 64/G rows into a wave; everything else to the wave-per-row kernel with 4 gathers in flight.
  const bool thin = nnz >= 0 && nnz < 8 * n_rows;
  if (vec_ok) {
    switch (F) {
      case 16:  return launch_spmm<16, 4>(rowptr, idx, val, x, ldx, n_src_rows, bias, out, ldo, n_rows, accumulate, s);
      case 32:  return launch_spmm<32, 4>(rowptr, idx, val, x, ldx, n_src_rows, bias, out, ldo, n_rows, accumulate, s);
      case 64:
        if (thin) return launch_thin<64>(rowptr, idx, val, x, ldx, n_src_rows, bias, out, ldo, n_rows, accumulate, s);
        return launch_spmm<64, 4>(rowptr, idx, val, x, ldx, n_src_rows, bias, out, ldo, n_rows, accumulate, s);
      case 128:
        if (thin) return launch_thin<128>(rowptr, idx, val, x, ldx, n_src_rows, bias, out, ldo, n_rows, accumulate, s);
        return launch_spmm<128, 4>(rowptr, idx, val, x, ldx, n_src_rows, bias, out, ldo, n_rows, accumulate, s);
      case 256: return launch_spmm<256, 4>(rowptr, idx, val, x, ldx, n_src_rows, bias, out, ldo, n_rows, accumulate, s);
      default: break;
    }
  }
  const int64_t blocks = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
  hipLaunchKernelGGL(spmm_row_generic_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, rowptr,
                     idx, val, x, ldx, bias, out, ldo, n_rows, (int)F, accumulate);
  PG_CHECK_LAUNCH("pangnn_spmm_csr_f32(generic)");
  return 0;
}

extern "C" int pangnn_segment_sum_rows_f32(const int64_t* rowptr, const int32_t* perm,
                                           const float* m, int64_t ldm, int64_t n_m_rows,
                                           int64_t col_off, float* out,
                                           int64_t ldo, int64_t n_rows, int32_t F, int accumulate,
                                           pangnn_stream_t stream) {
  PG_CHECK_ARG(col_off >= 0 && col_off + F <= ldm, PANGNN_E_BADARG,
               "pangnn_segment_sum_rows_f32: column window [%lld,%lld) outside ldm=%lld",
               (long long)col_off, (long long)(col_off + F), (long long)ldm);
  return pangnn_spmm_csr_f32(rowptr, perm, nullptr, m + col_off, ldm, n_m_rows, nullptr,
                             out, ldo, n_rows, n_m_rows, F, accumulate, stream);
}

static int spmm_csr_16(const char* who, int fmt, const int64_t* rowptr, const int32_t* idx, const float* val, const void* x16,
                       int64_t ldx, int64_t n_src_rows, const float* bias, float* out, int64_t ldo, int64_t n_rows, int32_t F,
                       int accumulate, pangnn_stream_t stream) {
  PG_CHECK_ARG(n_rows >= 0 && n_src_rows >= 0 && F > 0, PANGNN_E_BADARG, "%s: negative size", who);
  if (n_rows == 0) return 0;
  PG_CHECK_ARG(rowptr && out && (x16 || n_src_rows == 0) && ldx >= F && ldo >= F, PANGNN_E_BADARG,
               "%s: null pointer / leading dimension smaller than F", who);
  PG_CHECK_ARG((reinterpret_cast<uintptr_t>(x16) & 7u) == 0 && aligned16(out) && (!bias || aligned16(bias)) &&
                   ldx % 4 == 0 && ldo % 4 == 0,
               PANGNN_E_ALIGN, "%s: x must be 8-byte aligned, out / bias 16-byte, ld multiples of 4", who);
  PG_CHECK_ARG((n_rows + kWavesPerBlock - 1) / kWavesPerBlock < 2147483647LL, PANGNN_E_TOOLARGE,
               "%s: too many rows for one launch", who);
  hipStream_t s = (hipStream_t)stream;
#define PG_W(FW)                                                                                                              \
  case FW:                                                                                                                    \
    return fmt == PANGNN_DTYPE_F16                                                                                            \
               ? launch_spmm_16<FW, 4, 2>(rowptr, idx, val, x16, ldx, n_src_rows, bias, out, ldo, n_rows, accumulate, s)       \
               : launch_spmm_16<FW, 4, 1>(rowptr, idx, val, x16, ldx, n_src_rows, bias, out, ldo, n_rows, accumulate, s)
  switch (F) {
    PG_W(32);
    PG_W(64);
    PG_W(128);
    PG_W(256);
    default: break;
  }
#undef PG_W
  set_error("%s: F must be 32, 64, 128 or 256 (got %d)", who, (int)F);
  return PANGNN_E_BADARG;
}

extern "C" int pangnn_spmm_csr_bf16(const int64_t* rowptr, const int32_t* idx, const float* val,
                                    const void* x_bf16, int64_t ldx, int64_t n_src_rows,
                                    const float* bias, float* out, int64_t ldo, int64_t n_rows,
                                    int64_t nnz, int32_t F, int accumulate, pangnn_stream_t stream) {
  (void)nnz;
  return spmm_csr_16("pangnn_spmm_csr_bf16", PANGNN_DTYPE_BF16, rowptr, idx, val, x_bf16, ldx, n_src_rows, bias, out, ldo, n_rows,
                     F, accumulate, stream);
}

extern "C" int pangnn_spmm_csr_f16(const int64_t* rowptr, const int32_t* idx, const float* val,
                                   const void* x_f16, int64_t ldx, int64_t n_src_rows,
                                   const float* bias, float* out, int64_t ldo, int64_t n_rows,
                                   int64_t nnz, int32_t F, int accumulate, pangnn_stream_t stream) {
  (void)nnz;
  return spmm_csr_16("pangnn_spmm_csr_f16", PANGNN_DTYPE_F16, rowptr, idx, val, x_f16, ldx, n_src_rows, bias, out, ldo, n_rows,
                     F, accumulate, stream);
}

extern "C" size_t pangnn_band_propagate_workspace_bytes(int32_t F) {
  return (size_t)kBandBlocks * (size_t)(F > 0 ? F : 1) * sizeof(float);
}

extern "C" int pangnn_band_propagate(const void* x, int32_t x_dtype, int64_t ldx, const float* dis, const float* bias,
                                     float* out, int64_t ldo, int64_t n, int32_t F, int32_t k, float* colsum,
                                     void* workspace, size_t workspace_bytes, pangnn_stream_t stream) {
  const char* who = "pangnn_band_propagate";
  PG_CHECK_ARG(n >= 0 && k >= 0 && (F == 64 || F == 128), PANGNN_E_BADARG, "%s: F must be 64 or 128, n >= 0, k >= 0 (F=%d k=%d)",
               who, (int)F, (int)k);
  PG_CHECK_ARG(x_dtype == PANGNN_DTYPE_F32 || x_dtype == PANGNN_DTYPE_BF16 || x_dtype == PANGNN_DTYPE_F16, PANGNN_E_BADARG,
               "%s: x_dtype is PANGNN_DTYPE_F32 / _BF16 / _F16", who);
  PG_CHECK_ARG(!colsum || (workspace && workspace_bytes >= pangnn_band_propagate_workspace_bytes(F)), PANGNN_E_WORKSPACE,
               "%s: colsum needs the workspace", who);
  if (n == 0) {
    if (colsum) {
      hipError_t e = hipMemsetAsync(colsum, 0, (size_t)F * sizeof(float), (hipStream_t)stream);
      PG_CHECK_ARG(e == hipSuccess, (int)e, "%s: memset failed", who);
    }
    return 0;
  }
  PG_CHECK_ARG(x && dis && out && ldx >= F && ldo >= F && ldx % 4 == 0 && ldo % 4 == 0, PANGNN_E_BADARG,
               "%s: null pointer / leading dimension", who);
  PG_CHECK_ARG((x_dtype == PANGNN_DTYPE_F32 ? aligned16(x) : (reinterpret_cast<uintptr_t>(x) & 7u) == 0) && aligned16(out) &&
                   (!bias || aligned16(bias)),
               PANGNN_E_ALIGN, "%s: rows must start on 16 bytes (bf16 / f16 x: 8)", who);
  hipStream_t s = (hipStream_t)stream;
  const int xb = x_dtype;
  if (F == 64) return launch_band<64>(x, xb, ldx, dis, bias, out, ldo, n, (int)k, colsum, static_cast<float*>(workspace), s);
  return launch_band<128>(x, xb, ldx, dis, bias, out, ldo, n, (int)k, colsum, static_cast<float*>(workspace), s);
}
