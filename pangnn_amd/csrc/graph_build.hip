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
