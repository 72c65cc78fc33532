// gcn_norm (k1-k3), per-edge gathers for the link decoder (k7) and 'max' segment reduction.
#include <stdarg.h>
#include "common.h"

namespace pangnn {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// deg[i] = sum_{in-edges} w ; dis = deg^-1/2 (0 when deg == 0).  One wave per target row; lanes
// stride the row (fixed order => reproducible).
__global__ __launch_bounds__(kBlock) void degree_kernel(const int64_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ perm,
                                                        const float* __restrict__ w,
                                                        float* __restrict__ dis, int64_t n) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // SGPR
  if (row >= n) return;
  const int64_t beg = rowptr[row], end = rowptr[row + 1];
  float s = 0.f;
  if (w) {
    for (int64_t e = beg + lane; e < end; e += kWave) s += w[perm[e]];
    s = wave_sum(s);
  } else {
    s = (float)(end - beg);
  }
  if (lane == 0) {
    // PyG: deg.pow(-0.5); masked_fill(== inf, 0).  pow(x,-0.5) == 1/sqrt(x) for x > 0.
    float d = 1.0f / sqrtf(s);
    if (isinf(d)) d = 0.f;
    dis[row] = d;
  }
}

__global__ __launch_bounds__(kBlock) void edge_norm_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ src,
    const int32_t* __restrict__ perm, const float* __restrict__ w, const float* __restrict__ dis_src,
    const float* __restrict__ dis,
    float* __restrict__ norm_sorted, float* __restrict__ norm_orig, int64_t n) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // SGPR
  if (row >= n) return;
  const int64_t beg = rowptr[row], end = rowptr[row + 1];
  const float dc = dis[row];
  for (int64_t e = beg + lane; e < end; e += kWave) {
    const int32_t o = perm[e];
    const float we = w ? w[o] : 1.f;
    const float v = dis_src[src[e]] * we * dc;  // PyG order: (dis[row] * w) * dis[col]
    norm_sorted[e] = v;
    if (norm_orig) norm_orig[o] = v;
  }
}

__global__ __launch_bounds__(kBlock) void permute_kernel(const float* __restrict__ in,
                                                         const int32_t* __restrict__ perm,
                                                         float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock)
    out[i] = in[perm[i]];
}

// out[e, 0:D] = z[src], out[e, D:2D] = z[dst], (out[e,2D] = extra[e]).  VEC: float4 per lane.
template <bool VEC>
__global__ __launch_bounds__(kBlock) void gather_concat_kernel(
    const float* __restrict__ z, int64_t ldz, const int64_t* __restrict__ ei, int64_t ld,
    int64_t e_begin, int64_t n_edges, const float* __restrict__ extra, float* __restrict__ out,
    int64_t ldo, int D) {
  const int per_edge = VEC ? (2 * D / 4) : (2 * D);
  const int64_t total = n_edges * per_edge;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * kBlock) {
    const int64_t el = t / per_edge;
    const int c = (int)(t - el * per_edge);
    const int64_t e = e_begin + el;
    if (VEC) {
      const int half = c >= D / 4;
      const int64_t node = ei[half ? ld + e : e];
      const int cc = half ? c - D / 4 : c;
      const float4 v = reinterpret_cast<const float4*>(z + node * ldz)[cc];
      reinterpret_cast<float4*>(out + el * ldo)[c] = v;
    } else {
      const int half = c >= D;
      const int64_t node = ei[half ? ld + e : e];
      out[el * ldo + c] = z[node * ldz + (half ? c - D : c)];
      if (extra && c == 0) out[el * ldo + 2 * D] = extra[e];
    }
  }
}

// out[e, 0:D] = p[src] + q[dst] (+ extra[e] * cvec).  D % 4 == 0, float4 per lane.
__global__ __launch_bounds__(kBlock) void pair_add_kernel(
    const float* __restrict__ p, const float* __restrict__ q, int64_t ldpq,
    const int64_t* __restrict__ ei, int64_t ld, int64_t e_begin, int64_t n_edges,
    const float* __restrict__ extra, const float* __restrict__ cvec, float* __restrict__ out,
    int64_t ldo, int D) {
  const int per_edge = D / 4;
  const int64_t total = n_edges * per_edge;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * kBlock) {
    const int64_t el = t / per_edge;
    const int c = (int)(t - el * per_edge);
    const int64_t e = e_begin + el;
    const int64_t s = ei[e], d = ei[ld + e];
    const float4 a = reinterpret_cast<const float4*>(p + s * ldpq)[c];
    const float4 b = reinterpret_cast<const float4*>(q + d * ldpq)[c];
    float4 r = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    if (extra) {
      const float w = extra[e];
      const float4 cv = reinterpret_cast<const float4*>(cvec)[c];
      r.x = fmaf(w, cv.x, r.x); r.y = fmaf(w, cv.y, r.y);
      r.z = fmaf(w, cv.z, r.z); r.w = fmaf(w, cv.w, r.w);
    }
    reinterpret_cast<float4*>(out + el * ldo)[c] = r;
  }
}

// 'max' aggregation over the edges of a target row (convolution.py:7).  Wave per row, lanes stride
// the feature dim, edges serial in ascending original id (first maximum wins, as a sequential
// scatter would).
__global__ __launch_bounds__(kBlock) void segment_max_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ perm,
    const float* __restrict__ m, int64_t ldm, float* __restrict__ out, int32_t* __restrict__ arg,
    int64_t ldo, int64_t n_rows, int F) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // SGPR
  if (row >= n_rows) return;
  const int64_t beg = rowptr[row], end = rowptr[row + 1];
  for (int f = lane; f < F; f += kWave) {
    float best = 0.f;
    int32_t bi = -1;
    for (int64_t k = beg; k < end; ++k) {
      const int32_t o = perm[k];
      const float v = m[(int64_t)o * ldm + f];
      if (bi < 0 || v > best || (v != v && best == best)) { best = v; bi = o; }  // NaN propagates
    }
    out[row * ldo + f] = best;
    if (arg) arg[row * ldo + f] = bi;
  }
}

__global__ __launch_bounds__(kBlock) void segment_max_bwd_kernel(
    const float* __restrict__ g, const int32_t* __restrict__ arg, float* __restrict__ gm,
    int64_t ldm, int64_t ldo, int64_t n_rows, int F) {
  const int64_t total = n_rows * F;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * kBlock) {
    const int64_t r = t / F;
    const int f = (int)(t - r * F);
    const int32_t a = arg[r * ldo + f];
    if (a >= 0) gm[(int64_t)a * ldm + f] = g[r * ldo + f];  // every (edge, f) has one owner row
  }
}

// BCEWithLogits(pos_weight), mean over `denom` edges (pangnn.py:98,203), forward AND gradient in one
// pass:  l = (1-y) x + (1 + (pw-1) y) softplus(-x),   dl/dx = (1-y) - (1 + (pw-1) y) sigmoid(-x).
// Each block sums a fixed slice in a fixed order; a second single-block pass adds the block partials in
// index order (reproducible).
constexpr int kBceBlocks = 1024;
__global__ __launch_bounds__(kBlock) void bce_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                     const float* __restrict__ pos_weight, int64_t n,
                                                     float inv_denom, float* __restrict__ g,
                                                     float* __restrict__ partial) {
  __shared__ float red[kBlock / kWave];
  const float pw = pos_weight ? pos_weight[0] : 1.f;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const float xv = x[i], yv = y[i];
    const float lw = 1.f + (pw - 1.f) * yv;
    const float ax = fabsf(xv);
    const float t = expf(-ax);
    const float sp = log1pf(t) + fmaxf(-xv, 0.f);            // softplus(-x)
    acc += (1.f - yv) * xv + lw * sp;
    const float sig_neg = xv >= 0.f ? t / (1.f + t) : 1.f / (1.f + t);   // sigmoid(-x)
    g[i] = ((1.f - yv) - lw * sig_neg) * inv_denom;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < kBlock / kWave; ++w) s += red[w];
    partial[blockIdx.x] = s * inv_denom;
  }
}

__global__ void bce_finish_kernel(const float* __restrict__ partial, int n, float* __restrict__ loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += partial[i];
    loss[0] = s;
  }
}

// counts[2*label + prediction] += 1 with prediction = (score >= threshold), score = sigmoid(x) or x itself.
// pangnn.py:218-222,257-262: probabilities = sigmoid(output); (probabilities >= binary_th).int();
// BinaryConfusionMatrix.update(prediction, labels).  Integer atomics: the result is order independent.
__global__ __launch_bounds__(kBlock) void confusion_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           int64_t n, float threshold, int apply_sigmoid,
                                                           unsigned long long* __restrict__ counts) {
  __shared__ unsigned int red[kBlock / kWave][4];
  unsigned int c[4] = {0u, 0u, 0u, 0u};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const float xv = x[i];
    const float score = apply_sigmoid ? 1.0f / (1.0f + expf(-xv)) : xv;
    const int pred = score >= threshold ? 1 : 0, lab = y[i] > 0.5f ? 1 : 0;
    c[2 * lab + pred] += 1u;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned int v = c[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    unsigned long long t = 0;
    for (int w = 0; w < kBlock / kWave; ++w) t += red[w][threadIdx.x];
    if (t) atomicAdd(&counts[threadIdx.x], t);
  }
}

// out[0, c] = sum_n r[n] g[n, c],  out[1, c] = sum_n s[n] g[n, c]   (c < F <= 256, F | 256).
// Parameter gradients of the scalar-feature embedding that feeds the first GCN layer (src/gnn.py:97,125,158):
// with h0 = x w^T + 1 b^T one has dL/dw = (A_hat x)^T g and dL/db = (A_hat 1)^T g, so no transposed propagate
// is needed for a layer whose input carries no other gradient.  Two-stage fixed-order sum.
constexpr int kColsumBlocks = 1024;
// VEC = 4: a thread owns 4 adjacent columns (16-byte loads, F/4 lanes per row); VEC = 1 for F < 4 or odd strides
template <int VEC>
__global__ __launch_bounds__(kBlock) void weighted_colsum_kernel(const float* __restrict__ g, int64_t ldg,
                                                                 const float* __restrict__ r,
                                                                 const float* __restrict__ sv, int64_t n, int F,
                                                                 float* __restrict__ partial) {
  __shared__ float red[2][kBlock * VEC];
  const int lpr = F / VEC;                        // lanes per row
  const int c = threadIdx.x % lpr, rg = threadIdx.x / lpr, groups = kBlock / lpr;
  float a0[VEC], a1[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) a0[v] = a1[v] = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * groups + rg; row < n; row += (int64_t)gridDim.x * groups) {
    float v[VEC];
    if constexpr (VEC == 4) {
      const float4 t = *reinterpret_cast<const float4*>(g + row * ldg + 4 * c);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
      v[0] = g[row * ldg + c];
    }
    const float rv = r[row], sw = sv[row];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { a0[k] = fmaf(rv, v[k], a0[k]); a1[k] = fmaf(sw, v[k], a1[k]); }
  }
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    red[0][rg * F + VEC * c + k] = a0[k];
    red[1][rg * F + VEC * c + k] = a1[k];
  }
  __syncthreads();
  if (threadIdx.x < F) {
    float t0 = 0.f, t1 = 0.f;
    for (int k = 0; k < groups; ++k) { t0 += red[0][k * F + threadIdx.x]; t1 += red[1][k * F + threadIdx.x]; }
    partial[((int64_t)blockIdx.x * 2 + 0) * F + threadIdx.x] = t0;
    partial[((int64_t)blockIdx.x * 2 + 1) * F + threadIdx.x] = t1;
  }
}

__global__ __launch_bounds__(kSumThreads) void colsum_finish_kernel(const float* __restrict__ partial, int nblocks,
                                                                    int F, float* __restrict__ out) {
  const int i = blockIdx.x * kWave + (threadIdx.x & (kWave - 1));
  const float t = ordered_parts_sum(partial, nblocks, 2 * F, i, 2 * F);
  if (threadIdx.x < kWave && i < 2 * F) out[i] = t;
}

// ------------------------------------------------------------------------------------------------------------------
// First GCN layer of the scalar-feature model by linearity (src/gnn.py:97,125,158: h0 = x w^T + 1 b^T, one scalar per
// node):  conv_in(embedding(x)) = A_hat (x w^T + 1 b^T) W^T + b_in = r a^T + s c^T + b_in  with the node vectors
// r = A_hat x, s = A_hat 1 (once per graph) and a = W w, c = W b (per step, H-vectors).  rank2_rows_kernel writes
// that [N, H] matrix (HBM-write bound); its backward needs only [r s 1]^T g = three weighted column sums of g.
// ------------------------------------------------------------------------------------------------------------------
template <typename TY>
__global__ __launch_bounds__(kBlock) void rank2_rows_kernel(const float* __restrict__ r, const float* __restrict__ sv,
                                                            const float* __restrict__ a, const float* __restrict__ c,
                                                            const float* __restrict__ bias, TY* __restrict__ out,
                                                            int64_t ldo, int64_t n, int F) {
  const int lpr = F / 4;                           // lanes per row (4 adjacent columns each)
  const int col = 4 * (threadIdx.x % lpr), rg = threadIdx.x / lpr, groups = kBlock / lpr;
  const float4 av = *reinterpret_cast<const float4*>(a + col), cv = *reinterpret_cast<const float4*>(c + col);
  const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t row = (int64_t)blockIdx.x * groups + rg; row < n; row += (int64_t)gridDim.x * groups) {
    const float rv = r[row], sw = sv[row];
    const float v0 = fmaf(rv, av.x, fmaf(sw, cv.x, bv.x)), v1 = fmaf(rv, av.y, fmaf(sw, cv.y, bv.y));
    const float v2 = fmaf(rv, av.z, fmaf(sw, cv.z, bv.z)), v3 = fmaf(rv, av.w, fmaf(sw, cv.w, bv.w));
    if constexpr (sizeof(TY) == 4) {
      *reinterpret_cast<float4*>(out + row * ldo + col) = make_float4(v0, v1, v2, v3);
    } else {
      *reinterpret_cast<uint2*>(out + row * ldo + col) = f32_to_rows16(v0, v1, v2, v3, RowFmt<TY>::value);   // round to nearest even
    }
  }
}

// out[0] = sum_n r[n] g[n,:], out[1] = sum_n s[n] g[n,:], out[2] = sum_n g[n,:]; g stored as f32 or bf16 (4 adjacent
// columns per thread).  Two-stage fixed-order sum like weighted_colsum_kernel.
template <typename TG>
__global__ __launch_bounds__(kBlock) void weighted_colsum3_kernel(const TG* __restrict__ g, int64_t ldg,
                                                                  const float* __restrict__ r,
                                                                  const float* __restrict__ sv, int64_t n, int F,
                                                                  float* __restrict__ partial) {
  __shared__ float red[3][kBlock * 4];
  const int lpr = F / 4;
  const int c = threadIdx.x % lpr, rg = threadIdx.x / lpr, groups = kBlock / lpr;
  float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t row = (int64_t)blockIdx.x * groups + rg; row < n; row += (int64_t)gridDim.x * groups) {
    float v[4];
    if constexpr (sizeof(TG) == 4) {
      const float4 t = *reinterpret_cast<const float4*>(g + row * ldg + 4 * c);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
      const float4 t = rows16_to_f32(*reinterpret_cast<const uint2*>(g + row * ldg + 4 * c), RowFmt<TG>::value);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    const float rv = r[row], sw = sv[row];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a0[k] = fmaf(rv, v[k], a0[k]); a1[k] = fmaf(sw, v[k], a1[k]); a2[k] += v[k]; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    red[0][rg * F + 4 * c + k] = a0[k];
    red[1][rg * F + 4 * c + k] = a1[k];
    red[2][rg * F + 4 * c + k] = a2[k];
  }
  __syncthreads();
  if (threadIdx.x < F) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    for (int k = 0; k < groups; ++k) {
      t0 += red[0][k * F + threadIdx.x]; t1 += red[1][k * F + threadIdx.x]; t2 += red[2][k * F + threadIdx.x];
    }
    float* p = partial + (int64_t)blockIdx.x * 3 * F;
    p[threadIdx.x] = t0; p[F + threadIdx.x] = t1; p[2 * F + threadIdx.x] = t2;
  }
}

__global__ __launch_bounds__(kSumThreads) void colsum3_finish_kernel(const float* __restrict__ partial, int nblocks,
                                                                     int F, float* __restrict__ out) {
  const int i = blockIdx.x * kWave + (threadIdx.x & (kWave - 1));
  const float t = ordered_parts_sum(partial, nblocks, 3 * F, i, 3 * F);
  if (threadIdx.x < kWave && i < 3 * F) out[i] = t;
}

// rank2_rows_kernel with a = W w and c = W b formed in the kernel (per workgroup, in LDS): no host-side parameter
// algebra, one launch for the whole layer.  W [H, D] row-major, H <= 256.
template <typename TY>
__global__ __launch_bounds__(kBlock) void embed_conv_in_rows_kernel(const float* __restrict__ r, const float* __restrict__ sv,
                                                                    const float* __restrict__ w_emb,
                                                                    const float* __restrict__ b_emb,
                                                                    const float* __restrict__ w_in,
                                                                    const float* __restrict__ bias, int D,
                                                                    TY* __restrict__ out, int64_t ldo, int64_t n, int F) {
  __shared__ __attribute__((aligned(16))) float ac[2][kBlock];
  if ((int)threadIdx.x < F) {
    float a = 0.f, c = 0.f;
    const float* wr = w_in + (int64_t)threadIdx.x * D;
#pragma unroll 16
    for (int d = 0; d < D; ++d) { a = fmaf(wr[d], w_emb[d], a); c = fmaf(wr[d], b_emb[d], c); }
    ac[0][threadIdx.x] = a;
    ac[1][threadIdx.x] = c;
  }
  __syncthreads();
  const int lpr = F / 4;
  const int col = 4 * (threadIdx.x % lpr), rg = threadIdx.x / lpr, groups = kBlock / lpr;
  const float4 av = *reinterpret_cast<const float4*>(&ac[0][col]), cv = *reinterpret_cast<const float4*>(&ac[1][col]);
  const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t row = (int64_t)blockIdx.x * groups + rg; row < n; row += (int64_t)gridDim.x * groups) {
    const float rv = r[row], sw = sv[row];
    const float v0 = fmaf(rv, av.x, fmaf(sw, cv.x, bv.x)), v1 = fmaf(rv, av.y, fmaf(sw, cv.y, bv.y));
    const float v2 = fmaf(rv, av.z, fmaf(sw, cv.z, bv.z)), v3 = fmaf(rv, av.w, fmaf(sw, cv.w, bv.w));
    if constexpr (sizeof(TY) == 4) {
      *reinterpret_cast<float4*>(out + row * ldo + col) = make_float4(v0, v1, v2, v3);
    } else {
      *reinterpret_cast<uint2*>(out + row * ldo + col) = f32_to_rows16(v0, v1, v2, v3, RowFmt<TY>::value);   // round to nearest even
    }
  }
}

// the layer's parameter gradients from sums = [r s 1]^T g  ([3, H], finished):  dL/da = sums[0], dL/dc = sums[1]
//   g_w_in[h][d] = sums[0][h] w[d] + sums[1][h] b[d]     g_w_emb[d] = sum_h W[h][d] sums[0][h]
//   g_b_in[h]    = sums[2][h]                            g_b_emb[d] = sum_h W[h][d] sums[1][h]      (fixed order)
__global__ __launch_bounds__(kBlock) void embed_conv_in_param_grads_kernel(const float* __restrict__ sums,
                                                                           const float* __restrict__ w_emb,
                                                                           const float* __restrict__ b_emb,
                                                                           const float* __restrict__ w_in, int D, int H,
                                                                           float* __restrict__ g_w_emb,
                                                                           float* __restrict__ g_b_emb,
                                                                           float* __restrict__ g_w_in,
                                                                           float* __restrict__ g_b_in) {
  // (loads of several iterations in flight in both loops: this one-workgroup kernel is a chain of cache round trips otherwise)
#pragma unroll 8
  for (int i = threadIdx.x; i < H * D; i += kBlock) {
    const int h = i / D, d = i % D;
    g_w_in[i] = fmaf(sums[h], w_emb[d], sums[H + h] * b_emb[d]);
  }
  for (int d = threadIdx.x; d < D; d += kBlock) {
    float u = 0.f, v = 0.f;
#pragma unroll 16
    for (int h = 0; h < H; ++h) { u = fmaf(w_in[(int64_t)h * D + d], sums[h], u); v = fmaf(w_in[(int64_t)h * D + d], sums[H + h], v); }
    g_w_emb[d] = u;
    g_b_emb[d] = v;
  }
  if (g_b_in)
    for (int h = threadIdx.x; h < H; h += kBlock) g_b_in[h] = sums[2 * H + h];
}

static inline unsigned grid_for(int64_t total) {
  int64_t b = (total + kBlock - 1) / kBlock;
  const int64_t cap = 256 * 16;  // 256 CUs x 16 blocks, grid-stride the rest
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ------------------------------------------------------------------------------------------------------------------
// normalize_sim_scores (src/preprocessing.py:454-548) as ONE segmented pass over the (source, candidate genome)-sorted
// relation: per segment  p = softmax(score / t)  (a single candidate: p = 1),  q = -10 log10(clip(1 - p, eps, 1 - eps))
// + pseudo_count, in float64 like the reference's numpy.  One wavefront per segment (segments hold ~2 .. 500
// candidates): lanes stride over the segment for the maximum, then for sum exp(x - max) with a fixed xor-shuffle
// tree (deterministic), then write q.  Replaces four float64 ATen scatter / gather passes.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void softmax_qscore_kernel(const int64_t* __restrict__ rowptr,
                                                               const double* __restrict__ score, int64_t nseg, double t,
                                                               double eps, double pseudo, double* __restrict__ q) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t seg = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (seg >= nseg) return;                       // wave-uniform: whole waves leave
  const int64_t beg = rowptr[seg], end = rowptr[seg + 1];
  const double q_one = -10.0 * log10(eps) + pseudo;                     // p = 1: clip(0, eps, 1 - eps) = eps
  if (end - beg == 1) {
    if (lane == 0) q[beg] = q_one;
    return;
  }
  double mx = -INFINITY;
  for (int64_t i = beg + lane; i < end; i += kWave) mx = fmax(mx, score[i] / t);     // divide like the reference (t = 0.8 is not a binary fraction)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
  double sm = 0.0;
  for (int64_t i = beg + lane; i < end; i += kWave) sm += exp(score[i] / t - mx);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sm += __shfl_xor(sm, off);
  const double lse = log(sm) + mx;                                      // scipy.special.logsumexp
  for (int64_t i = beg + lane; i < end; i += kWave) {
    const double p = exp(score[i] / t - lse);
    double c = 1.0 - p;
    c = c < eps ? eps : (c > 1.0 - eps ? 1.0 - eps : c);
    const double v = -10.0 * log10(c);
    q[i] = (p != p ? -10.0 * log10(1.0 - eps) : v) + pseudo;            // NaN p: the reference's nan_to_num branch
  }
}

// (r, s) = (A_hat x, A_hat 1) of one CSR order: r[i] = sum_k val[k] x[other[k]], s[i] = sum_k val[k] over row i — the two
// node vectors through which a scalar-feature embedding acts after one propagate (functional._node_actions).  One wave per
// row, lanes stride the row, xor-shuffle finish: one launch where the generic propagate on a 16-column table took three
// (table build, propagate, transpose-copy) — what a fresh mini-batch pays every step.
__global__ __launch_bounds__(kBlock) void node_actions_kernel(const int64_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ other,
                                                              const float* __restrict__ val, const float* __restrict__ x,
                                                              int64_t n, float* __restrict__ r, float* __restrict__ sv) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (row >= n) return;
  const int64_t b = rowptr[row], e = rowptr[row + 1];
  float ar = 0.f, as = 0.f;
  for (int64_t k = b + lane; k < e; k += kWave) {
    const float v = val ? val[k] : 1.f;
    as += v;
    ar = fmaf(v, x[other[k]], ar);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    ar += __shfl_xor(ar, off);
    as += __shfl_xor(as, off);
  }
  if (lane == 0) {
    r[row] = ar;
    sv[row] = as;
  }
}

// Operands of the decoder's re-associated first layer from mlp[0] = Linear(2D (+1), D) (src/gnn.py:110,173-175) in ONE launch:
// w_pq [2D][D] = [W[:, :D] ; W[:, D:2D]], b_pq [2D] = [0 ; b], cvec [D] = W[:, 2D] (skip connections).  Plain data movement.
__global__ __launch_bounds__(kBlock) void pq_operands_kernel(const float* __restrict__ w, int64_t ldw,
                                                             const float* __restrict__ b, int d, float* __restrict__ w_pq,
                                                             float* __restrict__ b_pq, float* __restrict__ cvec) {
  const int total = 2 * d * d;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
    const int row = i / d, col = i - row * d;                 // row of w_pq: half * d + j
    const int half = row >= d, j = row - half * d;
    w_pq[i] = w[(int64_t)j * ldw + half * d + col];
  }
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < 2 * d; i += kBlock) b_pq[i] = i < d ? 0.f : b[i - d];
    if (cvec)
      for (int i = threadIdx.x; i < d; i += kBlock) cvec[i] = w[(int64_t)i * ldw + 2 * d];
  }
}

// column sums of a SHORT matrix (a mini-batch's node rows) in ONE one-workgroup launch.  A thread owns 4 adjacent columns
// (16-byte loads, f32; 8-byte, bf16); the F / 4 threads of a row group take rows q, q + R, q + 2R, ... (R = 1024 / (F / 4) row
// groups) with eight loads in flight — on 900 rows a dependent load-add per L2 round trip took 89 us — and the R partial
// rows are added in group order: fixed order of additions, reproducible.  VEC = 1: any F <= 1024 / alignment, scalar columns.
template <typename TG, int VEC>
__global__ __launch_bounds__(kSumThreads) void colsum_small_kernel(const TG* __restrict__ g, int64_t ldg, int64_t n, int f,
                                                                   float* __restrict__ out) {
  __shared__ float red[kSumThreads * VEC];
  const int lpr = (f + VEC - 1) / VEC;                      // threads per row
  const int groups = kSumThreads / lpr;                     // row groups of the block (>= 1: F <= 1024 VEC)
  const int c = threadIdx.x % lpr, q = threadIdx.x / lpr;
  float s[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) s[v] = 0.f;
  auto ld = [&](int64_t row, float (&v)[VEC]) {
    const TG* p = g + row * ldg + VEC * c;
    if constexpr (VEC == 4 && sizeof(TG) == 4) {
      const float4 t = *reinterpret_cast<const float4*>(p);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else if constexpr (VEC == 4) {
      const float4 t = rows16_to_f32(*reinterpret_cast<const uint2*>(p), RowFmt<TG>::value);   // four 2-byte elements: exact in f32
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else if constexpr (sizeof(TG) == 4) {
      v[0] = *p;
    } else {
      v[0] = row16_to_f32(__builtin_bit_cast(unsigned short, *p), RowFmt<TG>::value);
    }
  };
  if (q < groups) {
    int64_t row = q;
    for (; row + 7 * (int64_t)groups < n; row += 8 * (int64_t)groups) {
      float v[8][VEC];
#pragma unroll
      for (int u = 0; u < 8; ++u) ld(row + u * (int64_t)groups, v[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < VEC; ++k) s[k] += v[u][k];
    }
    for (; row < n; row += groups) {
      float v[VEC];
      ld(row, v);
#pragma unroll
      for (int k = 0; k < VEC; ++k) s[k] += v[k];
    }
  }
#pragma unroll
  for (int k = 0; k < VEC; ++k) red[threadIdx.x * VEC + k] = s[k];
  __syncthreads();
  for (int col = threadIdx.x; col < f; col += kSumThreads) {
    float t = 0.f;
    for (int r = 0; r < groups; ++r) t += red[(r * lpr + col / VEC) * VEC + col % VEC];
    out[col] = t;
  }
}

// Gradients of the one-pass training decoder times the upstream gradient of the loss (pangnn_scale_unless_one_f32): every
// workgroup reads the device scalar first and leaves when it is exactly 1 — what `loss.backward()` hands over — so the usual
// step pays one launch and no memory traffic; any other value (a GradScaler's scale, a loss divided for gradient
// accumulation) is applied in place, 16 bytes per lane where the buffer allows.
struct ScaleList {
  float* p[PANGNN_SCALE_MAX_TENSORS];
  int64_t n[PANGNN_SCALE_MAX_TENSORS];
};

__global__ __launch_bounds__(kBlock) void scale_unless_one_kernel(ScaleList l, const float* __restrict__ scale) {
  const float s = *scale;
  if (s == 1.0f) return;
  float* p = l.p[blockIdx.y];
  const int64_t n = l.n[blockIdx.y];
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
    float4* p4 = reinterpret_cast<float4*>(p);
    const int64_t n4 = n >> 2;
    for (int64_t i = t; i < n4; i += stride) {
      float4 v = p4[i];
      v.x *= s; v.y *= s; v.z *= s; v.w *= s;
      p4[i] = v;
    }
    for (int64_t i = (n4 << 2) + t; i < n; i += stride) p[i] *= s;
  } else {
    for (int64_t i = t; i < n; i += stride) p[i] *= s;
  }
}

}  // namespace pangnn

using namespace pangnn;

extern "C" int pangnn_abi_version(void) { return PANGNN_ABI_VERSION; }
extern "C" const char* pangnn_last_error(void) { return g_err; }

extern "C" int pangnn_gcn_norm_f32(const int64_t* rowptr_dst, const int32_t* src_sorted,
                                   const int32_t* perm_dst, const float* edge_weight,
                                   int64_t num_nodes, int64_t num_edges, float* deg_inv_sqrt,
                                   float* norm_sorted, float* norm_orig, pangnn_stream_t stream) {
  PG_CHECK_ARG(num_nodes >= 0 && num_edges >= 0, PANGNN_E_BADARG, "pangnn_gcn_norm_f32: negative size");
  if (num_nodes == 0) return 0;
  PG_CHECK_ARG(rowptr_dst && deg_inv_sqrt && (num_edges == 0 || (src_sorted && perm_dst && norm_sorted)),
               PANGNN_E_BADARG, "pangnn_gcn_norm_f32: null pointer");
  const int64_t blocks = (num_nodes + kWavesPerBlock - 1) / kWavesPerBlock;
  PG_CHECK_ARG(blocks < 2147483647LL, PANGNN_E_TOOLARGE, "pangnn_gcn_norm_f32: too many nodes");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(degree_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, rowptr_dst, perm_dst,
                     edge_weight, deg_inv_sqrt, num_nodes);
  PG_CHECK_LAUNCH("pangnn_gcn_norm_f32(degree)");
  if (num_edges > 0) {
    hipLaunchKernelGGL(edge_norm_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, rowptr_dst,
                       src_sorted, perm_dst, edge_weight, deg_inv_sqrt, deg_inv_sqrt, norm_sorted, norm_orig,
                       num_nodes);
    PG_CHECK_LAUNCH("pangnn_gcn_norm_f32(norm)");
  }
  return 0;
}

extern "C" int pangnn_gcn_degree_f32(const int64_t* rowptr_dst, const int32_t* perm_dst,
                                     const float* edge_weight, int64_t n_rows, float* deg_inv_sqrt,
                                     pangnn_stream_t stream) {
  PG_CHECK_ARG(n_rows >= 0, PANGNN_E_BADARG, "pangnn_gcn_degree_f32: negative size");
  if (n_rows == 0) return 0;
  PG_CHECK_ARG(rowptr_dst && deg_inv_sqrt && (!edge_weight || perm_dst), PANGNN_E_BADARG,
               "pangnn_gcn_degree_f32: null pointer");
  const int64_t blocks = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
  PG_CHECK_ARG(blocks < 2147483647LL, PANGNN_E_TOOLARGE, "pangnn_gcn_degree_f32: too many rows");
  hipLaunchKernelGGL(degree_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, rowptr_dst,
                     perm_dst, edge_weight, deg_inv_sqrt, n_rows);
  PG_CHECK_LAUNCH("pangnn_gcn_degree_f32");
  return 0;
}

extern "C" int pangnn_gcn_edge_norm_f32(const int64_t* rowptr_dst, const int32_t* src_sorted,
                                        const int32_t* perm_dst, const float* edge_weight,
                                        const float* dis_src, const float* dis_dst, int64_t n_rows,
                                        int64_t num_edges, float* norm_sorted, float* norm_orig,
                                        pangnn_stream_t stream) {
  PG_CHECK_ARG(n_rows >= 0 && num_edges >= 0, PANGNN_E_BADARG, "pangnn_gcn_edge_norm_f32: negative size");
  if (n_rows == 0 || num_edges == 0) return 0;
  PG_CHECK_ARG(rowptr_dst && src_sorted && perm_dst && dis_src && dis_dst && norm_sorted, PANGNN_E_BADARG,
               "pangnn_gcn_edge_norm_f32: null pointer");
  const int64_t blocks = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
  PG_CHECK_ARG(blocks < 2147483647LL, PANGNN_E_TOOLARGE, "pangnn_gcn_edge_norm_f32: too many rows");
  hipLaunchKernelGGL(edge_norm_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, rowptr_dst,
                     src_sorted, perm_dst, edge_weight, dis_src, dis_dst, norm_sorted, norm_orig, n_rows);
  PG_CHECK_LAUNCH("pangnn_gcn_edge_norm_f32");
  return 0;
}

extern "C" int pangnn_permute_f32(const float* in, const int32_t* perm, float* out, int64_t n,
                                  pangnn_stream_t stream) {
  PG_CHECK_ARG(n >= 0, PANGNN_E_BADARG, "pangnn_permute_f32: negative size");
  if (n == 0) return 0;
  PG_CHECK_ARG(in && perm && out, PANGNN_E_BADARG, "pangnn_permute_f32: null pointer");
  hipLaunchKernelGGL(permute_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, in,
                     perm, out, n);
  PG_CHECK_LAUNCH("pangnn_permute_f32");
  return 0;
}

extern "C" int pangnn_edge_gather_concat_f32(const float* z, int64_t ldz, int64_t num_nodes,
                                             const int64_t* edge_index, int64_t ld, int64_t e_begin,
                                             int64_t n_edges, const float* extra, float* out,
                                             int64_t ldo, int32_t D, pangnn_stream_t stream) {
  PG_CHECK_ARG(n_edges >= 0 && e_begin >= 0 && D > 0 && num_nodes >= 0, PANGNN_E_BADARG,
               "pangnn_edge_gather_concat_f32: bad size");
  if (n_edges == 0) return 0;
  PG_CHECK_ARG(z && edge_index && out, PANGNN_E_BADARG, "pangnn_edge_gather_concat_f32: null pointer");
  PG_CHECK_ARG(e_begin + n_edges <= ld, PANGNN_E_BADARG,
               "pangnn_edge_gather_concat_f32: edge window [%lld,%lld) outside ld=%lld",
               (long long)e_begin, (long long)(e_begin + n_edges), (long long)ld);
  PG_CHECK_ARG(ldo >= 2 * D + (extra ? 1 : 0) && ldz >= D, PANGNN_E_BADARG,
               "pangnn_edge_gather_concat_f32: leading dimension too small");
  const bool vec = !extra && D % 4 == 0 && ldz % 4 == 0 && ldo % 4 == 0 && aligned16(z) && aligned16(out);
  hipStream_t s = (hipStream_t)stream;
  if (vec)
    hipLaunchKernelGGL((gather_concat_kernel<true>), dim3(grid_for(n_edges * (2 * D / 4))), dim3(kBlock),
                       0, s, z, ldz, edge_index, ld, e_begin, n_edges, extra, out, ldo, (int)D);
  else
    hipLaunchKernelGGL((gather_concat_kernel<false>), dim3(grid_for(n_edges * 2 * D)), dim3(kBlock), 0,
                       s, z, ldz, edge_index, ld, e_begin, n_edges, extra, out, ldo, (int)D);
  PG_CHECK_LAUNCH("pangnn_edge_gather_concat_f32");
  return 0;
}

extern "C" int pangnn_edge_pair_add_f32(const float* p, const float* q, int64_t ldpq,
                                        int64_t num_nodes, const int64_t* edge_index, int64_t ld,
                                        int64_t e_begin, int64_t n_edges, const float* extra,
                                        const float* cvec, float* out, int64_t ldo, int32_t D,
                                        pangnn_stream_t stream) {
  PG_CHECK_ARG(n_edges >= 0 && e_begin >= 0 && D > 0 && num_nodes >= 0, PANGNN_E_BADARG,
               "pangnn_edge_pair_add_f32: bad size");
  if (n_edges == 0) return 0;
  PG_CHECK_ARG(p && q && edge_index && out && (!extra || cvec), PANGNN_E_BADARG,
               "pangnn_edge_pair_add_f32: null pointer");
  PG_CHECK_ARG(e_begin + n_edges <= ld, PANGNN_E_BADARG, "pangnn_edge_pair_add_f32: edge window outside ld");
  PG_CHECK_ARG(D % 4 == 0 && ldpq % 4 == 0 && ldo % 4 == 0 && ldpq >= D && ldo >= D, PANGNN_E_BADARG,
               "pangnn_edge_pair_add_f32: D / leading dimensions must be multiples of 4");
  PG_CHECK_ARG(aligned16(p) && aligned16(q) && aligned16(out) && (!cvec || aligned16(cvec)),
               PANGNN_E_ALIGN, "pangnn_edge_pair_add_f32: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(pair_add_kernel, dim3(grid_for(n_edges * (D / 4))), dim3(kBlock), 0,
                     (hipStream_t)stream, p, q, ldpq, edge_index, ld, e_begin, n_edges, extra, cvec,
                     out, ldo, (int)D);
  PG_CHECK_LAUNCH("pangnn_edge_pair_add_f32");
  return 0;
}

extern "C" int pangnn_segment_max_rows_f32(const int64_t* rowptr, const int32_t* perm, const float* m,
                                           int64_t ldm, float* out, int32_t* arg, int64_t ldo,
                                           int64_t n_rows, int32_t F, pangnn_stream_t stream) {
  PG_CHECK_ARG(n_rows >= 0 && F > 0 && ldm >= F && ldo >= F, PANGNN_E_BADARG,
               "pangnn_segment_max_rows_f32: bad size");
  if (n_rows == 0) return 0;
  PG_CHECK_ARG(rowptr && out, PANGNN_E_BADARG, "pangnn_segment_max_rows_f32: null pointer");
  const int64_t blocks = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
  PG_CHECK_ARG(blocks < 2147483647LL, PANGNN_E_TOOLARGE, "pangnn_segment_max_rows_f32: too many rows");
  hipLaunchKernelGGL(segment_max_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream,
                     rowptr, perm, m, ldm, out, arg, ldo, n_rows, (int)F);
  PG_CHECK_LAUNCH("pangnn_segment_max_rows_f32");
  return 0;
}

extern "C" int pangnn_segment_max_bwd_f32(const float* g, const int32_t* arg, const int64_t* rowptr,
                                          float* gm, int64_t ldm, int64_t ldo, int64_t n_rows,
                                          int32_t F, pangnn_stream_t stream) {
  (void)rowptr;
  PG_CHECK_ARG(n_rows >= 0 && F > 0, PANGNN_E_BADARG, "pangnn_segment_max_bwd_f32: bad size");
  if (n_rows == 0) return 0;
  PG_CHECK_ARG(g && arg && gm, PANGNN_E_BADARG, "pangnn_segment_max_bwd_f32: null pointer");
  hipLaunchKernelGGL(segment_max_bwd_kernel, dim3(grid_for(n_rows * F)), dim3(kBlock), 0,
                     (hipStream_t)stream, g, arg, gm, ldm, ldo, n_rows, (int)F);
  PG_CHECK_LAUNCH("pangnn_segment_max_bwd_f32");
  return 0;
}

extern "C" size_t pangnn_bce_logits_workspace_bytes(void) { return kBceBlocks * sizeof(float); }

extern "C" int pangnn_bce_logits_f32(const float* logits, const float* y, const float* pos_weight, int64_t n,
                                     int64_t denom, float* loss, float* g_logits, void* workspace,
                                     size_t workspace_bytes, pangnn_stream_t stream) {
  PG_CHECK_ARG(n >= 0 && denom > 0, PANGNN_E_BADARG, "pangnn_bce_logits_f32: bad size");
  PG_CHECK_ARG(loss && (n == 0 || (logits && y && g_logits)), PANGNN_E_BADARG, "pangnn_bce_logits_f32: null pointer");
  PG_CHECK_ARG(workspace && workspace_bytes >= kBceBlocks * sizeof(float), PANGNN_E_WORKSPACE,
               "pangnn_bce_logits_f32: workspace too small (%zu < %zu)", workspace_bytes, (size_t)(kBceBlocks * sizeof(float)));
  hipStream_t s = (hipStream_t)stream;
  int blocks = (int)((n + kBlock - 1) / kBlock);
  if (blocks > kBceBlocks) blocks = kBceBlocks;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(bce_kernel, dim3(blocks), dim3(kBlock), 0, s, logits, y, pos_weight, n,
                     1.0f / (float)denom, g_logits, static_cast<float*>(workspace));
  PG_CHECK_LAUNCH("pangnn_bce_logits_f32");
  hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(64), 0, s, static_cast<const float*>(workspace), blocks,
                     loss);
  PG_CHECK_LAUNCH("pangnn_bce_logits_f32(finish)");
  return 0;
}

extern "C" int pangnn_node_actions_f32(const int64_t* rowptr, const int32_t* other, const float* val, const float* x,
                                       int64_t n_rows, float* r, float* s, pangnn_stream_t stream) {
  PG_CHECK_ARG(n_rows >= 0, PANGNN_E_BADARG, "pangnn_node_actions_f32: negative size");
  if (n_rows == 0) return 0;
  // `other` is NULL for an edge-less list (every row empty); E is not an argument, so that is the caller's contract
  PG_CHECK_ARG(rowptr && x && r && s, PANGNN_E_BADARG, "pangnn_node_actions_f32: null pointer");
  const int64_t blocks = (n_rows + kWavesPerBlock - 1) / kWavesPerBlock;
  hipLaunchKernelGGL(node_actions_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, rowptr, other, val, x,
                     n_rows, r, s);
  PG_CHECK_LAUNCH("pangnn_node_actions_f32");
  return 0;
}

extern "C" int pangnn_pq_operands_f32(const float* w, int64_t ldw, const float* b, int32_t d, int skip, float* w_pq,
                                      float* b_pq, float* cvec, pangnn_stream_t stream) {
  PG_CHECK_ARG(d > 0 && d <= 4096 && ldw >= 2 * (int64_t)d + (skip ? 1 : 0), PANGNN_E_BADARG,
               "pangnn_pq_operands_f32: w must be [D][2D (+1)] (D = %d, ldw = %lld)", (int)d, (long long)ldw);
  PG_CHECK_ARG(w && b && w_pq && b_pq && (!skip || cvec), PANGNN_E_BADARG, "pangnn_pq_operands_f32: null pointer");
  int blocks = (2 * d * d + kBlock - 1) / kBlock;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(pq_operands_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, w, ldw, b, (int)d, w_pq, b_pq,
                     skip ? cvec : static_cast<float*>(nullptr));
  PG_CHECK_LAUNCH("pangnn_pq_operands_f32");
  return 0;
}

extern "C" int pangnn_colsum_small(const void* g, int32_t g_dtype, int64_t ldg, int64_t n, int32_t F, float* out,
                                   pangnn_stream_t stream) {
  PG_CHECK_ARG(n >= 0 && F > 0 && F <= kSumThreads && ldg >= F && out && (n == 0 || g), PANGNN_E_BADARG,
               "pangnn_colsum_small: bad argument (F = %d: 1 .. %d)", (int)F, kSumThreads);
  PG_CHECK_ARG(g_dtype == PANGNN_DTYPE_F32 || g_dtype == PANGNN_DTYPE_BF16 || g_dtype == PANGNN_DTYPE_F16, PANGNN_E_BADARG,
               "pangnn_colsum_small: storage types are PANGNN_DTYPE_F32 / _BF16 / _F16");
  const bool bf = g_dtype != PANGNN_DTYPE_F32;
  const bool vec = F % 4 == 0 && ldg % 4 == 0 && (reinterpret_cast<uintptr_t>(g) & (bf ? 7u : 15u)) == 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(1), block(kSumThreads);
#define PG_CS(T, V) hipLaunchKernelGGL((colsum_small_kernel<T, V>), grid, block, 0, st, static_cast<const T*>(g), ldg, n, (int)F, out)
  if (g_dtype == PANGNN_DTYPE_F16) { if (vec) PG_CS(_Float16, 4); else PG_CS(_Float16, 1); }
  else if (bf) { if (vec) PG_CS(unsigned short, 4); else PG_CS(unsigned short, 1); }
  else if (vec) PG_CS(float, 4);
  else PG_CS(float, 1);
#undef PG_CS
  PG_CHECK_LAUNCH("pangnn_colsum_small");
  return 0;
}

extern "C" size_t pangnn_weighted_colsum_workspace_bytes(int32_t F) {
  return (size_t)kColsumBlocks * 2 * (size_t)(F > 0 ? F : 1) * sizeof(float);
}

extern "C" int pangnn_weighted_colsum_f32(const float* g, int64_t ldg, const float* r, const float* s, int64_t n,
                                          int32_t F, float* out, void* workspace, size_t workspace_bytes,
                                          pangnn_stream_t stream) {
  PG_CHECK_ARG(n >= 0 && F > 0 && F <= kBlock && kBlock % F == 0 && ldg >= F, PANGNN_E_BADARG,
               "pangnn_weighted_colsum_f32: F must divide 256 (got %d)", (int)F);
  PG_CHECK_ARG(out && (n == 0 || (g && r && s)), PANGNN_E_BADARG, "pangnn_weighted_colsum_f32: null pointer");
  PG_CHECK_ARG(workspace && workspace_bytes >= (size_t)kColsumBlocks * 2 * F * sizeof(float), PANGNN_E_WORKSPACE,
               "pangnn_weighted_colsum_f32: workspace too small (%zu < %zu)", workspace_bytes,
               (size_t)kColsumBlocks * 2 * F * sizeof(float));
  hipStream_t st = (hipStream_t)stream;
  const int groups = kBlock / F;            // rows per block pass of the scalar form; the float4 form takes 4x
  int blocks = (int)((n + 4 * groups - 1) / (4 * groups));
  if (blocks > kColsumBlocks) blocks = kColsumBlocks;
  if (blocks < 1) blocks = 1;
  if (F % 4 == 0 && ldg % 4 == 0 && aligned16(g))
    hipLaunchKernelGGL(weighted_colsum_kernel<4>, dim3(blocks), dim3(kBlock), 0, st, g, ldg, r, s, n, (int)F,
                       static_cast<float*>(workspace));
  else
    hipLaunchKernelGGL(weighted_colsum_kernel<1>, dim3(blocks), dim3(kBlock), 0, st, g, ldg, r, s, n, (int)F,
                       static_cast<float*>(workspace));
  PG_CHECK_LAUNCH("pangnn_weighted_colsum_f32");
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((2 * F + kWave - 1) / kWave), dim3(kSumThreads), 0, st,
                     static_cast<const float*>(workspace), blocks, (int)F, out);
  PG_CHECK_LAUNCH("pangnn_weighted_colsum_f32(finish)");
  return 0;
}

extern "C" int pangnn_rank2_rows(const float* r, const float* s, const float* a, const float* c, const float* bias,
                                 void* out, int32_t out_dtype, int64_t ldo, int64_t n, int32_t F, pangnn_stream_t stream) {
  const char* who = "pangnn_rank2_rows";
  PG_CHECK_ARG(n >= 0 && F > 0 && F % 4 == 0 && F <= 4 * kBlock && (4 * kBlock) % F == 0 && ldo >= F && ldo % 4 == 0,
               PANGNN_E_BADARG, "%s: F must be a multiple of 4 dividing 1024, ldo >= F and a multiple of 4 (got %d)", who, (int)F);
  PG_CHECK_ARG(out_dtype == PANGNN_DTYPE_F32 || out_dtype == PANGNN_DTYPE_BF16 || out_dtype == PANGNN_DTYPE_F16, PANGNN_E_BADARG,
               "%s: out_dtype is PANGNN_DTYPE_F32 / _BF16 / _F16", who);
  if (n == 0) return 0;
  PG_CHECK_ARG(r && s && a && c && out, PANGNN_E_BADARG, "%s: null pointer", who);
  PG_CHECK_ARG(aligned16(a) && aligned16(c) && (!bias || aligned16(bias)) &&
                   (out_dtype == PANGNN_DTYPE_F32 ? aligned16(out) : (reinterpret_cast<uintptr_t>(out) & 7u) == 0),
               PANGNN_E_ALIGN, "%s: a / c / bias / out rows must be 16-byte (bf16 out: 8-byte) aligned", who);
  const int groups = kBlock / (F / 4);
  int64_t blocks = (n + groups - 1) / groups;
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (out_dtype == PANGNN_DTYPE_F32)
    hipLaunchKernelGGL(rank2_rows_kernel<float>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, r, s, a, c, bias,
                       static_cast<float*>(out), ldo, n, (int)F);
  else if (out_dtype == PANGNN_DTYPE_F16)
    hipLaunchKernelGGL(rank2_rows_kernel<_Float16>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, r, s, a,
                       c, bias, static_cast<_Float16*>(out), ldo, n, (int)F);
  else
    hipLaunchKernelGGL(rank2_rows_kernel<unsigned short>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, r, s, a,
                       c, bias, static_cast<unsigned short*>(out), ldo, n, (int)F);
  PG_CHECK_LAUNCH(who);
  return 0;
}

extern "C" int pangnn_embed_conv_in_rows(const float* r, const float* s, const float* w_emb, const float* b_emb,
                                         const float* w_in, const float* b_in, int32_t D, void* out, int32_t out_dtype,
                                         int64_t ldo, int64_t n, int32_t H, pangnn_stream_t stream) {
  const char* who = "pangnn_embed_conv_in_rows";
  PG_CHECK_ARG(n >= 0 && D > 0 && H > 0 && H % 4 == 0 && H <= kBlock && (4 * kBlock) % H == 0 && ldo >= H && ldo % 4 == 0,
               PANGNN_E_BADARG, "%s: H must be a multiple of 4 dividing 1024, at most 256; ldo >= H (H=%d)", who, (int)H);
  PG_CHECK_ARG(out_dtype == PANGNN_DTYPE_F32 || out_dtype == PANGNN_DTYPE_BF16 || out_dtype == PANGNN_DTYPE_F16, PANGNN_E_BADARG,
               "%s: out_dtype is PANGNN_DTYPE_F32 / _BF16 / _F16", who);
  if (n == 0) return 0;
  PG_CHECK_ARG(r && s && w_emb && b_emb && w_in && out, PANGNN_E_BADARG, "%s: null pointer", who);
  PG_CHECK_ARG((!b_in || aligned16(b_in)) &&
                   (out_dtype == PANGNN_DTYPE_F32 ? aligned16(out) : (reinterpret_cast<uintptr_t>(out) & 7u) == 0),
               PANGNN_E_ALIGN, "%s: b_in / out rows must be 16-byte (bf16 out: 8-byte) aligned", who);
  const int groups = kBlock / (H / 4);
  int64_t blocks = (n + groups - 1) / groups;
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (out_dtype == PANGNN_DTYPE_F32)
    hipLaunchKernelGGL(embed_conv_in_rows_kernel<float>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, r, s, w_emb,
                       b_emb, w_in, b_in, (int)D, static_cast<float*>(out), ldo, n, (int)H);
  else if (out_dtype == PANGNN_DTYPE_F16)
    hipLaunchKernelGGL(embed_conv_in_rows_kernel<_Float16>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, r,
                       s, w_emb, b_emb, w_in, b_in, (int)D, static_cast<_Float16*>(out), ldo, n, (int)H);
  else
    hipLaunchKernelGGL(embed_conv_in_rows_kernel<unsigned short>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, r,
                       s, w_emb, b_emb, w_in, b_in, (int)D, static_cast<unsigned short*>(out), ldo, n, (int)H);
  PG_CHECK_LAUNCH(who);
  return 0;
}

extern "C" size_t pangnn_weighted_colsum3_workspace_bytes(int32_t F) {
  return (size_t)kColsumBlocks * 3 * (size_t)(F > 0 ? F : 1) * sizeof(float);
}

extern "C" int pangnn_weighted_colsum3(const void* g, int32_t g_dtype, int64_t ldg, const float* r, const float* s,
                                       int64_t n, int32_t F, float* out, void* workspace, size_t workspace_bytes,
                                       pangnn_stream_t stream) {
  const char* who = "pangnn_weighted_colsum3";
  PG_CHECK_ARG(n >= 0 && F > 0 && F % 4 == 0 && F <= kBlock && (4 * kBlock) % F == 0 && ldg >= F && ldg % 4 == 0,
               PANGNN_E_BADARG, "%s: F must be a multiple of 4 dividing 1024, at most 256; ldg >= F and a multiple of 4 (got %d)", who, (int)F);
  PG_CHECK_ARG(g_dtype == PANGNN_DTYPE_F32 || g_dtype == PANGNN_DTYPE_BF16 || g_dtype == PANGNN_DTYPE_F16, PANGNN_E_BADARG,
               "%s: g_dtype is PANGNN_DTYPE_F32 / _BF16 / _F16", who);
  PG_CHECK_ARG(out && (n == 0 || (g && r && s)), PANGNN_E_BADARG, "%s: null pointer", who);
  PG_CHECK_ARG(workspace && workspace_bytes >= pangnn_weighted_colsum3_workspace_bytes(F), PANGNN_E_WORKSPACE,
               "%s: workspace too small (%zu < %zu)", who, workspace_bytes, pangnn_weighted_colsum3_workspace_bytes(F));
  PG_CHECK_ARG(n == 0 || (g_dtype == PANGNN_DTYPE_F32 ? aligned16(g) : (reinterpret_cast<uintptr_t>(g) & 7u) == 0),
               PANGNN_E_ALIGN, "%s: g rows must be 16-byte (bf16: 8-byte) aligned", who);
  hipStream_t st = (hipStream_t)stream;
  const int groups = kBlock / (F / 4);
  int blocks = (int)((n + 4 * groups - 1) / (4 * groups));
  if (blocks > kColsumBlocks) blocks = kColsumBlocks;
  if (blocks < 1) blocks = 1;
  if (g_dtype == PANGNN_DTYPE_F32)
    hipLaunchKernelGGL(weighted_colsum3_kernel<float>, dim3(blocks), dim3(kBlock), 0, st, static_cast<const float*>(g), ldg, r,
                       s, n, (int)F, static_cast<float*>(workspace));
  else if (g_dtype == PANGNN_DTYPE_F16)
    hipLaunchKernelGGL(weighted_colsum3_kernel<_Float16>, dim3(blocks), dim3(kBlock), 0, st,
                       static_cast<const _Float16*>(g), ldg, r, s, n, (int)F, static_cast<float*>(workspace));
  else
    hipLaunchKernelGGL(weighted_colsum3_kernel<unsigned short>, dim3(blocks), dim3(kBlock), 0, st,
                       static_cast<const unsigned short*>(g), ldg, r, s, n, (int)F, static_cast<float*>(workspace));
  PG_CHECK_LAUNCH(who);
  hipLaunchKernelGGL(colsum3_finish_kernel, dim3((3 * F + kWave - 1) / kWave), dim3(kSumThreads), 0, st,
                     static_cast<const float*>(workspace), blocks, (int)F, out);
  PG_CHECK_LAUNCH("pangnn_weighted_colsum3(finish)");
  return 0;
}

// the first layer's parameter gradients from the finished sums [3, H] = [r s 1]^T dL/dh (device), however they were formed
// (pangnn_weighted_colsum3, or pangnn_embed_linear_bwd which never writes dL/dh)
extern "C" int pangnn_embed_conv_in_grads_from_sums(const float* sums, const float* w_emb, const float* b_emb,
                                                    const float* w_in, int32_t D, int32_t H, float* g_w_emb, float* g_b_emb,
                                                    float* g_w_in, float* g_b_in, pangnn_stream_t stream) {
  const char* who = "pangnn_embed_conv_in_grads_from_sums";
  PG_CHECK_ARG(D > 0 && H > 0 && sums && w_emb && b_emb && w_in && g_w_emb && g_b_emb && g_w_in, PANGNN_E_BADARG,
               "%s: null pointer / size", who);
  hipLaunchKernelGGL(embed_conv_in_param_grads_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, sums, w_emb, b_emb, w_in,
                     (int)D, (int)H, g_w_emb, g_b_emb, g_w_in, g_b_in);
  PG_CHECK_LAUNCH(who);
  return 0;
}

extern "C" size_t pangnn_embed_conv_in_grads_workspace_bytes(int32_t H) {
  return pangnn_weighted_colsum3_workspace_bytes(H) + 3 * (size_t)(H > 0 ? H : 1) * sizeof(float);
}

extern "C" int pangnn_embed_conv_in_grads(const void* g, int32_t g_dtype, int64_t ldg, const float* r, const float* s,
                                          int64_t n, const float* w_emb, const float* b_emb, const float* w_in, int32_t D,
                                          int32_t H, float* g_w_emb, float* g_b_emb, float* g_w_in, float* g_b_in,
                                          void* workspace, size_t workspace_bytes, pangnn_stream_t stream) {
  const char* who = "pangnn_embed_conv_in_grads";
  PG_CHECK_ARG(D > 0 && H > 0 && w_emb && b_emb && w_in && g_w_emb && g_b_emb && g_w_in, PANGNN_E_BADARG,
               "%s: null pointer / size", who);
  PG_CHECK_ARG(workspace && workspace_bytes >= pangnn_embed_conv_in_grads_workspace_bytes(H), PANGNN_E_WORKSPACE,
               "%s: workspace too small", who);
  // sums [3, H] at the head of the workspace, the column-sum partials behind it
  float* sums = static_cast<float*>(workspace);
  char* rest = static_cast<char*>(workspace) + 3 * (size_t)H * sizeof(float);
  const int rc = pangnn_weighted_colsum3(g, g_dtype, ldg, r, s, n, H, sums, rest, workspace_bytes - 3 * (size_t)H * sizeof(float),
                                         stream);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(embed_conv_in_param_grads_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, sums, w_emb, b_emb, w_in,
                     (int)D, (int)H, g_w_emb, g_b_emb, g_w_in, g_b_in);
  PG_CHECK_LAUNCH(who);
  return 0;
}

extern "C" int pangnn_confusion_update_f32(const float* scores, const float* labels, int64_t n, float threshold,
                                           int apply_sigmoid, int64_t* counts, pangnn_stream_t stream) {
  PG_CHECK_ARG(n >= 0 && counts && (n == 0 || (scores && labels)), PANGNN_E_BADARG,
               "pangnn_confusion_update_f32: null pointer / negative size");
  // one thread counts at most 2^32-1 elements of its grid-stride slice
  PG_CHECK_ARG(n < ((int64_t)1 << 40), PANGNN_E_TOOLARGE, "pangnn_confusion_update_f32: n too large");
  if (n == 0) return 0;
  int blocks = (int)((n + kBlock - 1) / kBlock);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(confusion_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, scores, labels, n,
                     threshold, apply_sigmoid, reinterpret_cast<unsigned long long*>(counts));
  PG_CHECK_LAUNCH("pangnn_confusion_update_f32");
  return 0;
}

extern "C" int pangnn_softmax_qscore_f64(const int64_t* rowptr, const double* score, int64_t num_segments,
                                         int64_t num_items, double t, double epsilon, double pseudo_count, double* q,
                                         pangnn_stream_t stream) {
  PG_CHECK_ARG(num_segments >= 0 && num_items >= 0 && t > 0.0 && epsilon > 0.0, PANGNN_E_BADARG,
               "pangnn_softmax_qscore_f64: bad size / temperature / epsilon");
  if (num_segments == 0 || num_items == 0) return 0;
  PG_CHECK_ARG(rowptr && score && q, PANGNN_E_BADARG, "pangnn_softmax_qscore_f64: null pointer");
  const int64_t blocks = (num_segments + kWavesPerBlock - 1) / kWavesPerBlock;
  PG_CHECK_ARG(blocks < 2147483647LL, PANGNN_E_TOOLARGE, "pangnn_softmax_qscore_f64: too many segments");
  hipLaunchKernelGGL(softmax_qscore_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, rowptr, score,
                     num_segments, t, epsilon, pseudo_count, q);
  PG_CHECK_LAUNCH("pangnn_softmax_qscore_f64");
  return 0;
}

extern "C" int pangnn_scale_unless_one_f32(float* const* ptrs, const int64_t* counts, int32_t n_tensors, const float* scale,
                                           pangnn_stream_t stream) {
  PG_CHECK_ARG(n_tensors >= 0 && n_tensors <= PANGNN_SCALE_MAX_TENSORS, PANGNN_E_BADARG,
               "pangnn_scale_unless_one_f32: n_tensors = %d (0 .. %d)", (int)n_tensors, PANGNN_SCALE_MAX_TENSORS);
  if (n_tensors == 0) return 0;
  PG_CHECK_ARG(ptrs && counts && scale, PANGNN_E_BADARG, "pangnn_scale_unless_one_f32: null pointer");
  ScaleList l;
  int64_t longest = 0;
  for (int i = 0; i < PANGNN_SCALE_MAX_TENSORS; ++i) {
    l.p[i] = i < n_tensors ? ptrs[i] : nullptr;
    l.n[i] = i < n_tensors ? counts[i] : 0;
    PG_CHECK_ARG(l.n[i] >= 0 && (l.n[i] == 0 || l.p[i]), PANGNN_E_BADARG,
                 "pangnn_scale_unless_one_f32: tensor %d: null pointer or negative count", i);
    PG_CHECK_ARG((reinterpret_cast<uintptr_t>(l.p[i]) & 3u) == 0, PANGNN_E_ALIGN,
                 "pangnn_scale_unless_one_f32: tensor %d is not 4-byte aligned", i);
    if (l.n[i] > longest) longest = l.n[i];
  }
  if (longest == 0) return 0;
  int64_t blocks = (longest / 4 + kBlock - 1) / kBlock;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(scale_unless_one_kernel, dim3((unsigned)blocks, (unsigned)n_tensors), dim3(kBlock), 0,
                     (hipStream_t)stream, l, scale);
  PG_CHECK_LAUNCH("pangnn_scale_unless_one_f32");
  return 0;
}
