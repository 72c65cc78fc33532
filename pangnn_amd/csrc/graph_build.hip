// One-time structure build: COO edge_index[2][E] (int64) -> CSR grouped by target or by source.
// Stable LSD radix sort (rocPRIM) of (node key, edge id) pairs, so inside a row the edges keep
// ascending original id: the per-row summation order of every downstream kernel is therefore a
// function of the input alone (reproducible across runs and across GPU counts).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "common.h"

namespace pangnn {

__global__ __launch_bounds__(kBlock) void make_keys_kernel(const int64_t* __restrict__ key_row,
                                                           int32_t* __restrict__ keys,
                                                           int32_t* __restrict__ ids, int64_t e,
                                                           int64_t n, int* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < e;
       i += (int64_t)gridDim.x * kBlock) {
    const int64_t k = key_row[i];
    if (k < 0 || k >= n) *bad = 1;  // reported by the host wrapper on its next sync, never faults
    keys[i] = (int32_t)(k < 0 ? 0 : (k >= n ? n - 1 : k));
    ids[i] = (int32_t)i;
  }
}

__global__ __launch_bounds__(kBlock) void other_end_kernel(const int64_t* __restrict__ other_row,
                                                           const int32_t* __restrict__ perm,
                                                           int32_t* __restrict__ other, int64_t e,
                                                           int64_t n, int* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < e;
       i += (int64_t)gridDim.x * kBlock) {
    int64_t v = other_row[perm[i]];
    if (v < 0 || v >= n) *bad = 1;
    v = v < 0 ? 0 : (v >= n ? n - 1 : v);  // clamp: an out-of-range id must never become a wild gather
    other[i] = (int32_t)v;
  }
}

// rowptr[r] = first position whose key >= r  (keys sorted ascending), r in [0, n]
__global__ __launch_bounds__(kBlock) void rowptr_kernel(const int32_t* __restrict__ keys,
                                                        int64_t* __restrict__ rowptr, int64_t e,
                                                        int64_t n) {
  for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r <= n;
       r += (int64_t)gridDim.x * kBlock) {
    int64_t lo = 0, hi = e;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)keys[mid] < r) lo = mid + 1; else hi = mid;
    }
    rowptr[r] = lo;
  }
}

static inline unsigned grid_for(int64_t total) {
  int64_t b = (total + kBlock - 1) / kBlock;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int key_bits(int64_t n) {
  int b = 1;
  while (((int64_t)1 << b) < n && b < 31) ++b;
  return b;
}

static size_t sort_temp_bytes(int64_t e, int64_t n) {
  size_t bytes = 0;
  int32_t* nul = nullptr;
  // size query only: no kernel is launched when temporary_storage == nullptr
  hipError_t err = rocprim::radix_sort_pairs(nullptr, bytes, nul, nul, nul, nul, (size_t)e, 0u,
                                             (unsigned)key_bits(n), (hipStream_t)0);
  if (err != hipSuccess) return (size_t)-1;
  return bytes;
}

}  // namespace pangnn

using namespace pangnn;

// workspace layout: [keys_in E*4][keys_out E*4][ids_in E*4][flag 256][rocPRIM temp]
extern "C" size_t pangnn_csr_build_workspace_bytes(int64_t num_edges, int64_t num_nodes) {
  if (num_edges <= 0 || num_nodes <= 0) return 256;
  const size_t t = sort_temp_bytes(num_edges, num_nodes);
  if (t == (size_t)-1) return 0;
  return 3 * align_up((size_t)num_edges * 4, 256) + 256 + align_up(t, 256);
}

extern "C" int pangnn_csr_build(const int64_t* edge_index, int64_t ld, int64_t num_edges,
                                int64_t num_nodes, int group_by, int64_t* rowptr, int32_t* other,
                                int32_t* perm, void* workspace, size_t workspace_bytes,
                                pangnn_stream_t stream) {
  PG_CHECK_ARG(num_edges >= 0 && num_nodes >= 0 && ld >= num_edges, PANGNN_E_BADARG,
               "pangnn_csr_build: bad size (E=%lld N=%lld ld=%lld)", (long long)num_edges,
               (long long)num_nodes, (long long)ld);
  PG_CHECK_ARG(group_by == 0 || group_by == 1, PANGNN_E_BADARG, "pangnn_csr_build: group_by must be 0|1");
  PG_CHECK_ARG(num_edges < 2147483647LL && num_nodes < 2147483647LL, PANGNN_E_TOOLARGE,
               "pangnn_csr_build: E and N must fit int32 (partition the graph first)");
  PG_CHECK_ARG(rowptr, PANGNN_E_BADARG, "pangnn_csr_build: null rowptr");
  hipStream_t s = (hipStream_t)stream;
  if (num_edges == 0 || num_nodes == 0) {
    hipError_t e = hipMemsetAsync(rowptr, 0, (size_t)(num_nodes + 1) * sizeof(int64_t), s);
    PG_CHECK_ARG(e == hipSuccess, (int)e, "pangnn_csr_build: memset failed: %s", hipGetErrorString(e));
    return 0;
  }
  PG_CHECK_ARG(edge_index && other && perm && workspace, PANGNN_E_BADARG, "pangnn_csr_build: null pointer");
  const size_t seg = align_up((size_t)num_edges * 4, 256);
  const size_t temp = sort_temp_bytes(num_edges, num_nodes);
  PG_CHECK_ARG(temp != (size_t)-1, PANGNN_E_BADARG, "pangnn_csr_build: rocPRIM size query failed");
  PG_CHECK_ARG(workspace_bytes >= 3 * seg + 256 + align_up(temp, 256), PANGNN_E_WORKSPACE,
               "pangnn_csr_build: workspace too small (%zu < %zu)", workspace_bytes,
               3 * seg + 256 + align_up(temp, 256));
  PG_CHECK_ARG(aligned16(workspace), PANGNN_E_ALIGN, "pangnn_csr_build: workspace must be 16-byte aligned");
  char* ws = static_cast<char*>(workspace);
  int32_t* keys_in = reinterpret_cast<int32_t*>(ws);
  int32_t* keys_out = reinterpret_cast<int32_t*>(ws + seg);
  int32_t* ids_in = reinterpret_cast<int32_t*>(ws + 2 * seg);
  int* bad = reinterpret_cast<int*>(ws + 3 * seg);
  void* tmp = ws + 3 * seg + 256;

  const int64_t* key_row = edge_index + (group_by ? ld : 0);
  const int64_t* other_row = edge_index + (group_by ? 0 : ld);
  hipError_t e = hipMemsetAsync(bad, 0, 256, s);
  PG_CHECK_ARG(e == hipSuccess, (int)e, "pangnn_csr_build: memset failed: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(make_keys_kernel, dim3(grid_for(num_edges)), dim3(kBlock), 0, s, key_row, keys_in,
                     ids_in, num_edges, num_nodes, bad);
  PG_CHECK_LAUNCH("pangnn_csr_build(keys)");
  size_t tb = temp;
  e = rocprim::radix_sort_pairs(tmp, tb, keys_in, keys_out, ids_in, perm, (size_t)num_edges, 0u,
                                (unsigned)key_bits(num_nodes), s);
  PG_CHECK_ARG(e == hipSuccess, (int)e, "pangnn_csr_build: radix sort failed: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(other_end_kernel, dim3(grid_for(num_edges)), dim3(kBlock), 0, s, other_row, perm,
                     other, num_edges, num_nodes, bad);
  PG_CHECK_LAUNCH("pangnn_csr_build(other)");
  hipLaunchKernelGGL(rowptr_kernel, dim3(grid_for(num_nodes + 1)), dim3(kBlock), 0, s, keys_out, rowptr,
                     num_edges, num_nodes);
  PG_CHECK_LAUNCH("pangnn_csr_build(rowptr)");
  return 0;
}

// Host-readable validity flag of the last build that used `workspace` (1 = some node id was outside
// [0, N)).  The caller copies 4 bytes from this device address after synchronising.
extern "C" const void* pangnn_csr_build_flag_ptr(void* workspace, int64_t num_edges) {
  if (!workspace || num_edges <= 0) return nullptr;
  return static_cast<char*>(workspace) + 3 * align_up((size_t)num_edges * 4, 256);
}

// ==============================================================================================================
// Small graphs (a mini-batch of sub-graphs, reference regime pangnn.py:152-216): BOTH CSR orders of one edge list and
// the run-sum plans of both orders in ONE launch — workgroup 0 groups by target, workgroup 1 by source.  A fresh batch per
// step otherwise costs ~10 launches per CSR order (radix sort passes) and ~18 small index launches per plan, all
// launch-bound at these sizes.
//
// Per workgroup: (key << ib | edge id) composites of the E edges in LDS, padded to a power of two with 0xFFFFFFFF, bitonic
// sort (unique composites, so the result IS the stable sort by key: inside a row the edges keep ascending original id —
// the same perm / other / rowptr as pangnn_csr_build's stable radix sort), then one thread-contiguous walk over the
// sorted list that writes perm, other, the int32 keys, rowptr, and the chunk plan of that order
// (EdgeStructure._plan_of_sorted_keys: a part starts at every chunk start and at every key change):
//   part_off[c]     part id of the first edge of chunk c (chunk = chunk_edges consecutive sorted positions)
//   part_rowptr[r]  first part of row r  (= part id at rowptr[r]; total parts for rows behind the last edge)
//   last_part[0]    part id of the last edge (parts - 1)
// Limits: 1 <= E <= kSmallMaxEdges, N <= kSmallMaxNodes (composites stay below the pad value).
// ==============================================================================================================
namespace pangnn {

constexpr int kSmallMaxEdges = 16384;
constexpr int kSmallMaxNodes = 65536;
constexpr int kSmallThreads = 1024;

struct SmallOrder {
  int64_t* rowptr;       // [N+1]
  int32_t* other;        // [E]
  int32_t* perm;         // [E]
  int32_t* keys;         // [E]
  int32_t* part_off;     // [ceil(E / chunk_edges)]
  int64_t* part_rowptr;  // [N+1]
  int64_t* last_part;    // [1]
};

__global__ __launch_bounds__(kSmallThreads) void small_structure_kernel(const int64_t* __restrict__ edge_index, int64_t ld,
                                                                        int e, int n, int epad, int ib, int chunk_edges,
                                                                        SmallOrder by_dst, SmallOrder by_src,
                                                                        int* __restrict__ bad) {
  __shared__ uint32_t comp[kSmallMaxEdges];
  __shared__ int wave_tot[kSmallThreads / 64];
  const int tid = threadIdx.x;
  const bool dst_order = blockIdx.x == 0;
  const SmallOrder o = dst_order ? by_dst : by_src;
  const int64_t* key_row = edge_index + (dst_order ? ld : 0);
  const int64_t* other_row = edge_index + (dst_order ? 0 : ld);

  for (int i = tid; i < epad; i += kSmallThreads) {
    uint32_t c = 0xFFFFFFFFu;
    if (i < e) {
      int64_t k = key_row[i];
      if (k < 0 || k >= n) { *bad = 1; k = k < 0 ? 0 : n - 1; }     // clamp: never a wild row
      c = ((uint32_t)k << ib) | (uint32_t)i;
    }
    comp[i] = c;
  }
  __syncthreads();
  // already in order (a source-sorted list grouped by source, the usual case of a collated batch)?  then no sort
  int unsorted = 0;
  for (int i = tid + 1; i < epad; i += kSmallThreads) unsorted |= comp[i - 1] > comp[i];
  if (__syncthreads_or(unsorted)) {
    for (int k = 2; k <= epad; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        // pair index t -> elements (lo, lo | j).  A wave's 64 consecutive pair indices touch one aligned block of 128
        // elements, and for j < 128 both elements of every pair stay inside that block: those stages need no workgroup
        // barrier, only the in-order LDS pipe of the wave itself.
        for (int t = tid; t < (epad >> 1); t += kSmallThreads) {
          const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const int hi = lo | j;
          const uint32_t a = comp[lo], b = comp[hi];
          const bool up = (lo & k) == 0;
          if ((a > b) == up) { comp[lo] = b; comp[hi] = a; }
        }
        if (j >= 128 || j == 1) {
          __syncthreads();           // (j == 1: the next k starts with a wide stride)
        } else {
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
  }

  // Walk over the sorted list: wave w owns the contiguous range [w * seg, (w + 1) * seg), 64 consecutive positions per
  // round (LDS reads without bank conflicts, coalesced stores).  A part starts at every chunk start and every key change;
  // part id = (number of part starts up to and including the position) - 1.
  const int lane = tid & 63, wv = tid >> 6;
  constexpr int kWaves = kSmallThreads / 64;
  const int seg = ((epad + kWaves - 1) / kWaves + 63) & ~63;
  const int w0 = wv * seg < e ? wv * seg : e;
  const int w1 = (wv + 1) * seg < e ? (wv + 1) * seg : e;
  const uint32_t idmask = (1u << ib) - 1u;
  int cnt = 0;
  for (int r0 = w0; r0 < w1; r0 += 64) {
    const int i = r0 + lane;
    bool start = false;
    if (i < w1) {
      const int key = (int)(comp[i] >> ib);
      const int prev = i == 0 ? -1 : (int)(comp[i - 1] >> ib);
      start = (i % chunk_edges == 0) || key != prev;
    }
    cnt += __popcll(__ballot(start));
  }
  if (lane == 0) wave_tot[wv] = cnt;
  __syncthreads();
  int running = 0, total = 0;
  for (int w = 0; w < kWaves; ++w) {
    const int v = wave_tot[w];
    if (w < wv) running += v;
    total += v;
  }
  for (int r0 = w0; r0 < w1; r0 += 64) {
    const int i = r0 + lane;
    bool start = false;
    int key = 0, prev = 0, idx = 0;
    if (i < w1) {
      const uint32_t c = comp[i];
      key = (int)(c >> ib);
      idx = (int)(c & idmask);
      prev = i == 0 ? -1 : (int)(comp[i - 1] >> ib);
      start = (i % chunk_edges == 0) || key != prev;
    }
    const unsigned long long m = __ballot(start);
    if (i < w1) {
      const int id = running + __popcll(m & (~0ull >> (63 - lane))) - 1;
      if (i % chunk_edges == 0) o.part_off[i / chunk_edges] = id;
      for (int r = prev + 1; r <= key; ++r) {       // rows whose first edge is here (empty rows in between included)
        o.rowptr[r] = i;
        o.part_rowptr[r] = id;
      }
      o.perm[i] = idx;
      o.keys[i] = key;
      int64_t v = other_row[idx];
      if (v < 0 || v >= n) { *bad = 1; v = v < 0 ? 0 : n - 1; }
      o.other[i] = (int32_t)v;
    }
    running += __popcll(m);
  }
  // rows behind the last edge (all threads)
  const int last_key = (int)(comp[e - 1] >> ib);
  for (int r = last_key + 1 + tid; r <= n; r += kSmallThreads) {
    o.rowptr[r] = e;
    o.part_rowptr[r] = total;
  }
  if (tid == 0) o.last_part[0] = total - 1;
}

}  // namespace pangnn

extern "C" int pangnn_structure_small_supported(int64_t num_edges, int64_t num_nodes) {
  return num_edges >= 1 && num_edges <= kSmallMaxEdges && num_nodes >= 1 && num_nodes <= kSmallMaxNodes;
}

extern "C" int pangnn_structure_small(const int64_t* edge_index, int64_t ld, int64_t num_edges, int64_t num_nodes,
                                      int32_t chunk_edges, int64_t* rowptr_dst, int32_t* other_dst, int32_t* perm_dst,
                                      int32_t* keys_dst, int32_t* part_off_dst, int64_t* part_rowptr_dst,
                                      int64_t* last_part_dst, int64_t* rowptr_src, int32_t* other_src, int32_t* perm_src,
                                      int32_t* keys_src, int32_t* part_off_src, int64_t* part_rowptr_src,
                                      int64_t* last_part_src, int32_t* bad_flag, pangnn_stream_t stream) {
  PG_CHECK_ARG(pangnn_structure_small_supported(num_edges, num_nodes) && ld >= num_edges, PANGNN_E_BADARG,
               "pangnn_structure_small: needs 1 <= E <= %d, 1 <= N <= %d (got E=%lld N=%lld ld=%lld)", kSmallMaxEdges,
               kSmallMaxNodes, (long long)num_edges, (long long)num_nodes, (long long)ld);
  PG_CHECK_ARG(chunk_edges >= 32 && chunk_edges % 32 == 0, PANGNN_E_BADARG,
               "pangnn_structure_small: chunk_edges must be a positive multiple of 32 (got %d)", (int)chunk_edges);
  PG_CHECK_ARG(edge_index && rowptr_dst && other_dst && perm_dst && keys_dst && part_off_dst && part_rowptr_dst &&
                   last_part_dst && rowptr_src && other_src && perm_src && keys_src && part_off_src && part_rowptr_src &&
                   last_part_src && bad_flag,
               PANGNN_E_BADARG, "pangnn_structure_small: null pointer");
  int epad = 2, ib = 1;
  while (epad < num_edges) { epad <<= 1; ++ib; }
  hipStream_t s = (hipStream_t)stream;
  hipError_t err = hipMemsetAsync(bad_flag, 0, sizeof(int32_t), s);
  PG_CHECK_ARG(err == hipSuccess, (int)err, "pangnn_structure_small: memset failed: %s", hipGetErrorString(err));
  SmallOrder d{rowptr_dst, other_dst, perm_dst, keys_dst, part_off_dst, part_rowptr_dst, last_part_dst};
  SmallOrder r{rowptr_src, other_src, perm_src, keys_src, part_off_src, part_rowptr_src, last_part_src};
  hipLaunchKernelGGL(small_structure_kernel, dim3(2), dim3(kSmallThreads), 0, s, edge_index, ld, (int)num_edges,
                     (int)num_nodes, epad, ib, (int)chunk_edges, d, r, reinterpret_cast<int*>(bad_flag));
  PG_CHECK_LAUNCH("pangnn_structure_small");
  return 0;
}

// ==============================================================================================================
// Collation of a mini-batch out of the flat sub-graph storage (pangnn_amd/subgraphs.py; the reference's DataLoader +
// PyG Batch.from_data_list, pangnn.py:152-216): sub-graphs [i0, i0 + g) occupy the node range [n0, n0 + n), the
// similarity-edge range [e0, e0 + e) and the neighbour-edge range [b0, b0 + b) of the flat lists.  One launch writes the
// batch-local edge lists (ids shifted by -n0), the graph pointer, every node's graph id and the unit node feature.
// ==============================================================================================================
namespace pangnn {

__global__ __launch_bounds__(kBlock) void collate_kernel(const int64_t* __restrict__ ei, int64_t ld_e, int64_t e0, int64_t e,
                                                         const int64_t* __restrict__ nb, int64_t ld_b, int64_t b0, int64_t b,
                                                         const int64_t* __restrict__ node_off, int g, int64_t n0, int64_t n,
                                                         int64_t* __restrict__ out_ei, int64_t* __restrict__ out_nb,
                                                         int64_t* __restrict__ ptr, int64_t* __restrict__ batch,
                                                         float* __restrict__ x) {
  const int64_t total = 2 * e + 2 * b + (g + 1) + n;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    int64_t k = i;
    if (k < 2 * e) {
      const int64_t row = k >= e, col = k - row * e;
      out_ei[k] = ei[row * ld_e + e0 + col] - n0;
      continue;
    }
    k -= 2 * e;
    if (k < 2 * b) {
      const int64_t row = k >= b, col = k - row * b;
      out_nb[k] = nb[row * ld_b + b0 + col] - n0;
      continue;
    }
    k -= 2 * b;
    if (k <= g) {
      ptr[k] = node_off[k] - n0;
      continue;
    }
    k -= g + 1;
    // graph of node k: number of graph ends <= k  (torch.searchsorted(ptr[1:], k, right=True))
    int lo = 0, hi = g;
    const int64_t key = k + n0;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (node_off[mid + 1] <= key) lo = mid + 1; else hi = mid;
    }
    batch[k] = lo;
    x[k] = 1.0f;
  }
}

}  // namespace pangnn

extern "C" int pangnn_collate_subgraphs(const int64_t* edge_index, int64_t ld_e, int64_t e0, int64_t num_edges,
                                        const int64_t* nb_index, int64_t ld_b, int64_t b0, int64_t num_nb,
                                        const int64_t* node_off, int32_t num_graphs, int64_t n0, int64_t num_nodes,
                                        int64_t* out_edge_index, int64_t* out_nb_index, int64_t* out_ptr,
                                        int64_t* out_batch, float* out_x, pangnn_stream_t stream) {
  PG_CHECK_ARG(num_edges >= 0 && num_nb >= 0 && num_nodes >= 0 && num_graphs >= 0 && e0 >= 0 && b0 >= 0 &&
                   ld_e >= e0 + num_edges && ld_b >= b0 + num_nb,
               PANGNN_E_BADARG, "pangnn_collate_subgraphs: bad range");
  PG_CHECK_ARG(node_off && out_ptr && (num_edges == 0 || (edge_index && out_edge_index)) &&
                   (num_nb == 0 || (nb_index && out_nb_index)) && (num_nodes == 0 || (out_batch && out_x)),
               PANGNN_E_BADARG, "pangnn_collate_subgraphs: null pointer");
  const int64_t total = 2 * num_edges + 2 * num_nb + (num_graphs + 1) + num_nodes;
  hipLaunchKernelGGL(collate_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, edge_index, ld_e, e0,
                     num_edges, nb_index, ld_b, b0, num_nb, node_off, (int)num_graphs, n0, num_nodes, out_edge_index,
                     out_nb_index, out_ptr, out_batch, out_x);
  PG_CHECK_LAUNCH("pangnn_collate_subgraphs");
  return 0;
}

// ==============================================================================================================
// Fixed-shape collation (round 4): the same disjoint union, of ANY list of sub-graphs (a shuffled DataLoader batch,
// pangnn.py:152-153), written into buffers sized to the data set's maxima and padded with an inert tail, so that every
// kernel behind it runs on one shape and ONE captured HIP graph serves every mini-batch (train.ReplayedFreshStep).  The
// list of sub-graph ids lives in device memory (rewritten between replays by pangnn_set_i64); per-graph counts are turned
// into offsets by every workgroup itself (<= kMaxPadGraphs entries), then a grid-stride walk fills every output.
//   padding: similarity / neighbour edges are self loops of the padded nodes [n, N_max) — never real nodes (the caller
//   sizes max_nodes > any real node count), so no real row sees them —, spread evenly over them in non-decreasing id
//   order; weight 1, label 0; ptr entries beyond the batch = n, batch id of padded nodes = g; x = 1 everywhere.  live[0..3] = real edges / neighbour edges /
//   nodes / graphs, live[4] = 1 if the batch did not fit the maxima (the outputs are then truncated and must not be used).
// ==============================================================================================================
namespace pangnn {
constexpr int kMaxPadGraphs = 256;
struct BatchOrder { const int32_t* rank; int32_t* other; int32_t* perm; int32_t* keys; };
struct BatchOrders { int enabled; BatchOrder sim_dst, sim_src, nb_dst, nb_src; };

__global__ __launch_bounds__(kBlock) void collate_padded_kernel(
    const int64_t* __restrict__ ei, int64_t ld_e, const int64_t* __restrict__ nb, int64_t ld_b,
    const float* __restrict__ w, const float* __restrict__ y, const int64_t* __restrict__ node_off,
    const int64_t* __restrict__ edge_off, const int64_t* __restrict__ nb_off, const int64_t* __restrict__ ids, int max_g,
    int64_t num_graphs_total, int64_t max_e, int64_t max_b, int64_t max_n, int64_t* __restrict__ out_ei,
    int64_t* __restrict__ out_nb, float* __restrict__ out_w, float* __restrict__ out_y, int64_t* __restrict__ ptr,
    int64_t* __restrict__ batch, float* __restrict__ x, int64_t* __restrict__ live, BatchOrders ord) {
  __shared__ int64_t gid[kMaxPadGraphs], cn[kMaxPadGraphs + 1], ce[kMaxPadGraphs + 1], cb[kMaxPadGraphs + 1];
  __shared__ int64_t raw_id[kMaxPadGraphs], raw_n[kMaxPadGraphs], raw_e[kMaxPadGraphs], raw_b[kMaxPadGraphs];
  __shared__ int g_sh;
  // every slot's counts are fetched in parallel (one thread per slot: the loads of a slot depend on its id, the slots do not
  // depend on each other), then one thread turns them into offsets
  for (int k = threadIdx.x; k < max_g; k += kBlock) {
    const int64_t id = ids[k];
    const bool ok = id >= 0 && id < num_graphs_total;
    raw_id[k] = ok ? id : -1;
    raw_n[k] = ok ? node_off[id + 1] - node_off[id] : 0;
    raw_e[k] = ok ? edge_off[id + 1] - edge_off[id] : 0;
    raw_b[k] = ok ? nb_off[id + 1] - nb_off[id] : 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int g = 0;
    int64_t n = 0, e = 0, b = 0;
    cn[0] = ce[0] = cb[0] = 0;
    for (int k = 0; k < max_g; ++k) {
      if (raw_id[k] < 0) continue;                            // unused slot
      gid[g] = raw_id[k];
      n += raw_n[k]; e += raw_e[k]; b += raw_b[k];
      ++g;
      cn[g] = n; ce[g] = e; cb[g] = b;
    }
    g_sh = g;
  }
  __syncthreads();
  const int g = g_sh;
  const int64_t n = cn[g], e = ce[g], b = cb[g];
  const bool fits = e <= max_e && b <= max_b && n < max_n;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    live[0] = fits ? e : 0; live[1] = fits ? b : 0; live[2] = fits ? n : 0; live[3] = g; live[4] = fits ? 0 : 1;
  }
  const int64_t e_r = fits ? e : 0, b_r = fits ? b : 0, n_r = fits ? n : 0;       // a batch that does not fit is all padding
  // padded edge j of P goes to padded node n_r + floor(j (max_n - n_r) / P): non-decreasing ids >= n_r (the lists stay
  // source-sorted, the pads sort last in both CSR orders) and spread evenly — ONE padded node would be a row of
  // thousands of entries that a single lane group of the thin-row propagate walks serially (0.6 ms per step, measured)
  const int64_t n_padn = max_n - n_r, p_e = max_e - e_r, p_b = max_b - b_r;
  // slot of position k among the cumulative counts c[0..g]: the j with c[j] <= k < c[j + 1]
  auto slot = [&](const int64_t* c, int64_t k) {
    int lo = 0, hi = g - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (c[mid] <= k) lo = mid; else hi = mid - 1;
    }
    return lo;
  };
  // ---- both CSR orders of both lists WITHOUT a sort (round 4): inside a sub-graph the by-target / by-source order of its
  // edges never changes, so the data set holds every edge's position inside its sub-graph in each order (`rank`, computed
  // once); the batch's order is the sub-graphs' orders one after the other (sub-graph j's node ids all precede those of
  // j + 1), i.e. sorted position = first edge of the sub-graph + rank.  The padding is already in order behind them.
  // Emits perm / other / keys of each order; rowptr and the run-sum plans follow from the sorted keys (order_walk_kernel).
  if (ord.enabled) {
    const int64_t tot_o = max_e + max_b;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < tot_o; i += (int64_t)gridDim.x * kBlock) {
      const bool sim = i < max_e;
      const int64_t col = sim ? i : i - max_e;
      const int64_t real = sim ? e_r : b_r, p_l = sim ? p_e : p_b;
      const BatchOrder od = sim ? ord.sim_dst : ord.nb_dst, os = sim ? ord.sim_src : ord.nb_src;
      if (col < real) {
        const int64_t* c = sim ? ce : cb;
        const int j = slot(c, col);
        const int64_t id = gid[j];
        const int64_t t = (sim ? edge_off[id] : nb_off[id]) + (col - c[j]);            // flat edge
        const int64_t shift = cn[j] - node_off[id];
        const int64_t* src_l = sim ? ei : nb;
        const int64_t ld = sim ? ld_e : ld_b;
        const int32_t sv = (int32_t)(src_l[t] + shift), dv = (int32_t)(src_l[ld + t] + shift);
        const int64_t pd = c[j] + od.rank[t], ps = c[j] + os.rank[t];
        od.perm[pd] = (int32_t)col; od.other[pd] = sv; od.keys[pd] = dv;
        os.perm[ps] = (int32_t)col; os.other[ps] = dv; os.keys[ps] = sv;
      } else {
        const int32_t v = (int32_t)(n_r + ((col - real) * n_padn) / (p_l > 0 ? p_l : 1));
        od.perm[col] = (int32_t)col; od.other[col] = v; od.keys[col] = v;
        os.perm[col] = (int32_t)col; os.other[col] = v; os.keys[col] = v;
      }
    }
  }
  const int64_t total = 2 * max_e + 2 * max_b + (max_g + 1) + max_n;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    int64_t k = i;
    if (k < 2 * max_e) {
      const int64_t row = k >= max_e, col = k - row * max_e;
      int64_t v = n_r + ((col - e_r) * n_padn) / (p_e > 0 ? p_e : 1);
      if (col < e_r) {
        const int j = slot(ce, col);
        const int64_t id = gid[j];
        v = ei[row * ld_e + edge_off[id] + (col - ce[j])] - node_off[id] + cn[j];
        if (row == 0) {
          out_w[col] = w[edge_off[id] + (col - ce[j])];
          out_y[col] = y[edge_off[id] + (col - ce[j])];
        }
      } else if (row == 0) {
        out_w[col] = 1.0f;
        out_y[col] = 0.0f;
      }
      out_ei[k] = v;
      continue;
    }
    k -= 2 * max_e;
    if (k < 2 * max_b) {
      const int64_t row = k >= max_b, col = k - row * max_b;
      int64_t v = n_r + ((col - b_r) * n_padn) / (p_b > 0 ? p_b : 1);
      if (col < b_r) {
        const int j = slot(cb, col);
        const int64_t id = gid[j];
        v = nb[row * ld_b + nb_off[id] + (col - cb[j])] - node_off[id] + cn[j];
      }
      out_nb[k] = v;
      continue;
    }
    k -= 2 * max_b;
    if (k <= max_g) {
      ptr[k] = k <= g ? cn[k] : n_r;
      continue;
    }
    k -= max_g + 1;
    batch[k] = k < n_r ? slot(cn, k) : g;
    x[k] = 1.0f;
  }
}

// rowptr and the run-sum plan of up to four orders from their SORTED int32 keys, one workgroup per order: the walk of
// small_structure_kernel (same part ids, same rowptr / part_rowptr fill of empty rows) over keys read from global memory.
struct WalkOrder { const int32_t* keys; int e; int n; int64_t* rowptr; int32_t* part_off; int64_t* part_rowptr; int64_t* last_part; };
struct WalkOrders { WalkOrder o[4]; };
__global__ __launch_bounds__(kSmallThreads) void order_walk_kernel(WalkOrders ws, int chunk_edges) {
  __shared__ int wave_tot[kSmallThreads / 64];
  const WalkOrder o = ws.o[blockIdx.x];
  if (o.keys == nullptr) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int kWaves = kSmallThreads / 64;
  const int e = o.e, n = o.n;
  const int seg = ((e + kWaves - 1) / kWaves + 63) & ~63;
  const int w0 = wv * seg < e ? wv * seg : e;
  const int w1 = (wv + 1) * seg < e ? (wv + 1) * seg : e;
  // the keys are read-only here and alias none of the outputs: restrict-qualified so that the loads of several iterations
  // are in flight at once (each wave walks a few hundred keys: one cache round trip per 64 of them, twice, was the kernel)
  const int32_t* __restrict__ keys = o.keys;
  int64_t* __restrict__ rowptr = o.rowptr;
  int64_t* __restrict__ part_rowptr = o.part_rowptr;
  int32_t* __restrict__ part_off = o.part_off;
  int cnt = 0;
#pragma unroll 4
  for (int r0 = w0; r0 < w1; r0 += 64) {
    const int i = r0 + lane;
    bool start = false;
    if (i < w1) {
      const int key = keys[i];
      const int prev = i == 0 ? -1 : keys[i - 1];
      start = (i % chunk_edges == 0) || key != prev;
    }
    cnt += __popcll(__ballot(start));
  }
  if (lane == 0) wave_tot[wv] = cnt;
  __syncthreads();
  int running = 0, total = 0;
  for (int w = 0; w < kWaves; ++w) {
    const int v = wave_tot[w];
    if (w < wv) running += v;
    total += v;
  }
#pragma unroll 4
  for (int r0 = w0; r0 < w1; r0 += 64) {
    const int i = r0 + lane;
    bool start = false;
    int key = 0, prev = 0;
    if (i < w1) {
      key = keys[i];
      prev = i == 0 ? -1 : keys[i - 1];
      start = (i % chunk_edges == 0) || key != prev;
    }
    const unsigned long long m = __ballot(start);
    if (i < w1) {
      const int id = running + __popcll(m & (~0ull >> (63 - lane))) - 1;
      if (part_off != nullptr && i % chunk_edges == 0) part_off[i / chunk_edges] = id;
      for (int r = prev + 1; r <= key; ++r) {       // rows whose first edge is here (empty rows in between included)
        rowptr[r] = i;
        if (part_rowptr != nullptr) part_rowptr[r] = id;
      }
    }
    running += __popcll(m);
  }
  const int last_key = e > 0 ? keys[e - 1] : -1;
  for (int r = last_key + 1 + tid; r <= n; r += kSmallThreads) {
    rowptr[r] = e;
    if (part_rowptr != nullptr) part_rowptr[r] = total;
  }
  if (tid == 0 && o.last_part != nullptr) o.last_part[0] = total - 1;
}

struct I64x64 { int64_t v[64]; };
__global__ void set_i64_kernel(int64_t* __restrict__ dst, I64x64 vals, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = vals.v[threadIdx.x];
}

}  // namespace pangnn

extern "C" int pangnn_collate_subgraphs_padded(const int64_t* edge_index, int64_t ld_e, const int64_t* nb_index,
                                               int64_t ld_b, const float* edge_attr, const float* y,
                                               const int64_t* node_off, const int64_t* edge_off, const int64_t* nb_off,
                                               int64_t num_graphs_total, const int64_t* graph_ids, int32_t max_graphs,
                                               int64_t max_edges, int64_t max_nb, int64_t max_nodes,
                                               int64_t* out_edge_index, int64_t* out_nb_index, float* out_edge_attr,
                                               float* out_y, int64_t* out_ptr, int64_t* out_batch, float* out_x,
                                               int64_t* out_live, const pangnn_batch_orders* orders,
                                               pangnn_stream_t stream) {
  const char* who = "pangnn_collate_subgraphs_padded";
  PG_CHECK_ARG(max_graphs >= 1 && max_graphs <= kMaxPadGraphs, PANGNN_E_BADARG, "%s: max_graphs must be in [1, %d]", who,
               kMaxPadGraphs);
  PG_CHECK_ARG(max_edges >= 1 && max_nb >= 1 && max_nodes >= 2 && num_graphs_total >= 0 && ld_e >= 0 && ld_b >= 0,
               PANGNN_E_BADARG, "%s: bad size", who);
  PG_CHECK_ARG(edge_index && nb_index && edge_attr && y && node_off && edge_off && nb_off && graph_ids && out_edge_index &&
                   out_nb_index && out_edge_attr && out_y && out_ptr && out_batch && out_x && out_live,
               PANGNN_E_BADARG, "%s: null pointer", who);
  BatchOrders bo{};
  if (orders != nullptr) {
    PG_CHECK_ARG(pangnn_structure_small_supported(max_edges, max_nodes) && pangnn_structure_small_supported(max_nb, max_nodes),
                 PANGNN_E_TOOLARGE, "%s: orders need max_edges, max_nb <= %d and max_nodes <= %d", who, kSmallMaxEdges,
                 kSmallMaxNodes);
    PG_CHECK_ARG(orders->chunk_edges >= 32 && orders->chunk_edges % 32 == 0, PANGNN_E_BADARG,
                 "%s: orders->chunk_edges must be a positive multiple of 32", who);
    const pangnn_batch_order* src[4] = {&orders->sim_dst, &orders->sim_src, &orders->nb_dst, &orders->nb_src};
    BatchOrder* dst[4] = {&bo.sim_dst, &bo.sim_src, &bo.nb_dst, &bo.nb_src};
    for (int k = 0; k < 4; ++k) {
      PG_CHECK_ARG(src[k]->rank && src[k]->rowptr && src[k]->other && src[k]->perm && src[k]->keys, PANGNN_E_BADARG,
                   "%s: orders: null pointer in order %d", who, k);
      *dst[k] = BatchOrder{src[k]->rank, src[k]->other, src[k]->perm, src[k]->keys};
    }
    bo.enabled = 1;
  }
  const int64_t total = 2 * max_edges + 2 * max_nb + (max_graphs + 1) + max_nodes;
  hipLaunchKernelGGL(collate_padded_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, edge_index, ld_e,
                     nb_index, ld_b, edge_attr, y, node_off, edge_off, nb_off, graph_ids, (int)max_graphs, num_graphs_total,
                     max_edges, max_nb, max_nodes, out_edge_index, out_nb_index, out_edge_attr, out_y, out_ptr, out_batch,
                     out_x, out_live, bo);
  PG_CHECK_LAUNCH(who);
  if (orders != nullptr) {
    WalkOrders w{};
    const pangnn_batch_order* src[4] = {&orders->sim_dst, &orders->sim_src, &orders->nb_dst, &orders->nb_src};
    for (int k = 0; k < 4; ++k)
      w.o[k] = WalkOrder{src[k]->keys, (int)(k < 2 ? max_edges : max_nb), (int)max_nodes, src[k]->rowptr, src[k]->part_off,
                         src[k]->part_rowptr, src[k]->last_part};
    hipLaunchKernelGGL(order_walk_kernel, dim3(4), dim3(kSmallThreads), 0, (hipStream_t)stream, w, (int)orders->chunk_edges);
    PG_CHECK_LAUNCH(who);
  }
  return 0;
}

extern "C" int pangnn_set_i64(int64_t* dst, const int64_t* host_values, int32_t n, pangnn_stream_t stream) {
  PG_CHECK_ARG(dst && host_values && n >= 1 && n <= 64, PANGNN_E_BADARG, "pangnn_set_i64: 1 <= n <= 64 values");
  I64x64 vals;
  for (int i = 0; i < 64; ++i) vals.v[i] = i < n ? host_values[i] : 0;
  hipLaunchKernelGGL(set_i64_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dst, vals, (int)n);
  PG_CHECK_LAUNCH("pangnn_set_i64");
  return 0;
}
