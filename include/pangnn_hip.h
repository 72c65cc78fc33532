/*
 * pangnn_hip.h — C ABI of libpangnn_hip.so, the MI355X (gfx950) implementation of panGNN's
 * edge-weighted message-passing hot path.
 *
 * Every entry point is plain C: device pointers + sizes + a hipStream_t passed as void*.
 * No torch types cross this boundary.  All pointers are DEVICE pointers unless stated otherwise.
 * Every call only ENQUEUES work on `stream` (no synchronisation, no allocation) and returns
 *   0                      on success,
 *   a positive hipError_t  if a HIP runtime call / launch failed,
 *   a negative PANGNN_E_*  on an argument error (nothing was launched).
 * `pangnn_last_error()` returns a thread-local human-readable message for the last non-zero return.
 *
 * The reference (fischer-hub/panGNN) is pure Python; the operator it calls for this path is the
 * third-party torch_geometric GCNConv.  Each function below cites the reference call site / PyG
 * stage it replaces (paths relative to /root/reference, stage names k1..k7 from SURVEY.md §2.2).
 */
#ifndef PANGNN_HIP_H
#define PANGNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): pangnn_decoder_train_f32 / _mixed / pangnn_decoder_dgrad_f32 took live_edges in round 4 and the 32-byte record
 * layout changed without a bump; pangnn_scale_unless_one_f32 added.  A library of another version is refused
 * by pangnn_amd/_lib.py and by libpangnn_torch.so when it resolves this ABI. */
/* 3 (round 5, later): PANGNN_DTYPE_F16 accepted by the node-level *_dtype arguments, pangnn_spmm_csr_f16 added; the 128 x 128
 * weight gradient is covered (pangnn_linear_supported(128, 128, 1) = 1). */
#define PANGNN_ABI_VERSION 3

#define PANGNN_E_BADARG    (-1)  /* null pointer / negative size / unsupported feature width */
#define PANGNN_E_TOOLARGE  (-2)  /* size exceeds an int32 index range used by the kernels    */
#define PANGNN_E_WORKSPACE (-3)  /* workspace smaller than *_workspace_bytes() asked for     */
#define PANGNN_E_ALIGN     (-4)  /* pointer / leading dimension not 16-byte aligned          */

/* storage types of the *_mixed entry points: the arithmetic is fp32 either way, a matrix may be STORED as bfloat16 */
#define PANGNN_DTYPE_F32  0
#define PANGNN_DTYPE_BF16 1
#define PANGNN_DTYPE_F16  2   /* IEEE half rows (--mixed_precision fp16): accepted wherever a *_dtype argument accepts _BF16
                               * (dense layers, propagate gathers, generated rows, column sums, the decoder's P | Q tables); the
                               * 2-byte operands of one call are all bfloat16 or all float16. */

typedef void* pangnn_stream_t;   /* hipStream_t */

int         pangnn_abi_version(void);
const char* pangnn_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Graph structure (one-time per edge_index; replaces the implicit COO handling of
 * MessagePassing.propagate, src/gnn.py:158,165 -> PyG).
 *
 * Stable counting of a COO edge list edge_index[2][E] (row 0 = source j, row 1 = target i,
 * PyG flow 'source_to_target') into CSR grouped by `group_by`:
 *   group_by = 1: rows are TARGET nodes  (forward propagate,  out[i] += w * x[j])
 *   group_by = 0: rows are SOURCE nodes  (backward propagate, gx[j]  += w * g[i])
 * Outputs: rowptr[N+1] (int64), other[E] (int32: the opposite endpoint of every sorted edge),
 *          perm[E] (int32: original edge id of every sorted edge; stable => ascending in a row).
 * `ld` = element distance between row 0 and row 1 of edge_index (E when contiguous).
 * ---------------------------------------------------------------------------------------- */
size_t pangnn_csr_build_workspace_bytes(int64_t num_edges, int64_t num_nodes);
int    pangnn_csr_build(const int64_t* edge_index, int64_t ld, int64_t num_edges, int64_t num_nodes,
                        int group_by, int64_t* rowptr, int32_t* other, int32_t* perm,
                        void* workspace, size_t workspace_bytes, pangnn_stream_t stream);

/* Device address of the 4-byte validity flag of the last pangnn_csr_build that used `workspace`
 * (non-zero = some node id was outside [0, N) and was clamped).  Read it after synchronising. */
const void* pangnn_csr_build_flag_ptr(void* workspace, int64_t num_edges);

/* Small graphs — one mini-batch of sub-graphs (reference regime: DataLoader batches of 32 sub-graphs, pangnn.py:152-216,
 * collated by PyG's Batch): BOTH CSR orders of pangnn_csr_build (same stable order: identical rowptr / other / perm) and,
 * for each order, the int32 sorted keys and the run-sum plan over chunks of `chunk_edges` consecutive sorted positions
 * (the `part_off` / part row pointers the decoder's S / T kernels take, chunk_edges = 32 * pangnn_decoder_chunk_tiles_for(E)),
 * in ONE launch.  part_off_*[ceil(E / chunk_edges)], part_rowptr_*[N+1], last_part_*[1] = number of parts - 1.
 * bad_flag[0] (device) = 1 if a node id was outside [0, N) (clamped).  Limits: pangnn_structure_small_supported(E, N). */
int pangnn_structure_small_supported(int64_t num_edges, int64_t num_nodes);
int pangnn_structure_small(const int64_t* edge_index, int64_t ld, int64_t num_edges, int64_t num_nodes,
                           int32_t chunk_edges, int64_t* rowptr_dst, int32_t* other_dst, int32_t* perm_dst,
                           int32_t* keys_dst, int32_t* part_off_dst, int64_t* part_rowptr_dst, int64_t* last_part_dst,
                           int64_t* rowptr_src, int32_t* other_src, int32_t* perm_src, int32_t* keys_src,
                           int32_t* part_off_src, int64_t* part_rowptr_src, int64_t* last_part_src, int32_t* bad_flag,
                           pangnn_stream_t stream);

/* Collation of one mini-batch out of a flat sub-graph storage (the reference's DataLoader(batch_size=32) + PyG
 * Batch.from_data_list, pangnn.py:152-216; pangnn_amd/subgraphs.py): the batch is sub-graphs [i0, i0 + num_graphs), i.e.
 * nodes [n0, n0 + num_nodes), similarity edges [e0, e0 + num_edges) of edge_index[2][ld_e] and neighbour edges
 * [b0, b0 + num_nb) of nb_index[2][ld_b]; node_off points at entry i0 of the node offset table.  One launch writes the
 * batch-local lists out_edge_index[2][num_edges], out_nb_index[2][num_nb] (ids - n0), out_ptr[num_graphs + 1],
 * out_batch[num_nodes] (graph id of every node) and out_x[num_nodes] = 1 (the reference's node feature, dataset.py:369). */
int pangnn_collate_subgraphs(const int64_t* edge_index, int64_t ld_e, int64_t e0, int64_t num_edges,
                             const int64_t* nb_index, int64_t ld_b, int64_t b0, int64_t num_nb,
                             const int64_t* node_off, int32_t num_graphs, int64_t n0, int64_t num_nodes,
                             int64_t* out_edge_index, int64_t* out_nb_index, int64_t* out_ptr, int64_t* out_batch,
                             float* out_x, pangnn_stream_t stream);

/* Fixed-shape collation (train.ReplayedFreshStep: ONE captured HIP graph for every mini-batch of the reference's
 * DataLoader(batch_size=32, shuffle=True) loop, pangnn.py:152-216): the disjoint union of the sub-graphs whose ids are in the
 * DEVICE list graph_ids[max_graphs] (entries outside [0, num_graphs_total) are unused slots), in list order, written into
 * buffers of the fixed sizes max_edges / max_nb / max_nodes and padded with an inert tail: padded similarity and neighbour
 * edges are self loops of the padded nodes [n, max_nodes) — never real (max_nodes must exceed every batch's node count) —
 * spread evenly over them with non-decreasing ids (the lists stay source-sorted, the padding sorts last in both CSR orders
 * and no padded row is long), with weight 1 and label 0; out_ptr entries beyond the batch = n, out_batch of padded nodes =
 * g, out_x = 1.
 * node_off / edge_off / nb_off are the data set's [num_graphs_total + 1] offset tables (device); edge_attr / y its flat
 * per-edge arrays.  out_live[5] (device) = real similarity edges, neighbour edges, nodes, graphs, and 1 if the batch did
 * not fit (everything is then padding).  out_live is what the decoder entry points take as `live_edges`. */
/* One CSR order (by target / by source) of one edge list of a padded batch.  `rank` (in, per FLAT edge of the data set):
 * the edge's position among its own sub-graph's edges in this order (stable: equal keys keep list order) — a property of the
 * data set, computed once.  Outputs (device): rowptr [max_nodes + 1], other / perm / keys [max list length] as
 * pangnn_structure_small writes them; part_off / part_rowptr / last_part (nullable together): the run-sum plan of the order. */
typedef struct pangnn_batch_order {
  const int32_t* rank;
  int64_t* rowptr; int32_t* other; int32_t* perm; int32_t* keys;
  int32_t* part_off; int64_t* part_rowptr; int64_t* last_part;
} pangnn_batch_order;
/* `orders` of pangnn_collate_subgraphs_padded (nullable): also emit both CSR orders of both lists of the batch — without a
 * sort: sub-graph j's nodes all precede those of sub-graph j + 1, so the batch's order is the sub-graphs' own (static) orders
 * one after the other, sorted position = first edge of the sub-graph + rank; the padding is in order behind them.  The same
 * tables, entry for entry, as pangnn_structure_small on the collated lists (which costs a 70 us one-workgroup bitonic sort per
 * list).  chunk_edges: 32 * pangnn_decoder_chunk_tiles_for(max_edges).  Needs max_edges, max_nb, max_nodes within
 * pangnn_structure_small_supported. */
typedef struct pangnn_batch_orders {
  int32_t chunk_edges;
  pangnn_batch_order sim_dst, sim_src, nb_dst, nb_src;
} pangnn_batch_orders;
int pangnn_collate_subgraphs_padded(const int64_t* edge_index, int64_t ld_e, const int64_t* nb_index, int64_t ld_b,
                                    const float* edge_attr, const float* y, const int64_t* node_off,
                                    const int64_t* edge_off, const int64_t* nb_off, int64_t num_graphs_total,
                                    const int64_t* graph_ids, int32_t max_graphs, int64_t max_edges, int64_t max_nb,
                                    int64_t max_nodes, int64_t* out_edge_index, int64_t* out_nb_index,
                                    float* out_edge_attr, float* out_y, int64_t* out_ptr, int64_t* out_batch,
                                    float* out_x, int64_t* out_live, const pangnn_batch_orders* orders,
                                    pangnn_stream_t stream);
/* dst[0 .. n) (device) = host_values[0 .. n), n <= 64, carried in the kernel arguments of one tiny launch: ordered with the
 * stream like any kernel and safe to call again before it ran (no pinned staging buffer to overwrite) — how the host hands
 * the next batch's sub-graph ids to a captured graph. */
int pangnn_set_i64(int64_t* dst, const int64_t* host_values, int32_t n, pangnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * gcn_norm (k1-k3; PyG gcn_norm as called by GCNConv.forward, src/gnn.py:158):
 *   deg[i]  = sum over in-edges of w        (w = 1 when edge_weight == NULL, src/gnn.py:165)
 *   dis[i]  = deg[i]^-1/2, 0 where deg == 0
 *   norm[e] = dis[src_e] * w_e * dis[dst_e]
 * Input structure is the target-grouped CSR from pangnn_csr_build(group_by=1).
 * Outputs: deg_inv_sqrt[N]; norm_sorted[E] in CSR order; norm_orig[E] in the caller's edge order
 * (nullable).  Deterministic: per-row sums run in ascending original edge id.
 * ---------------------------------------------------------------------------------------- */
int pangnn_gcn_norm_f32(const int64_t* rowptr_dst, const int32_t* src_sorted, const int32_t* perm_dst,
                        const float* edge_weight, int64_t num_nodes, int64_t num_edges,
                        float* deg_inv_sqrt, float* norm_sorted, float* norm_orig,
                        pangnn_stream_t stream);

/* The two stages of gcn_norm separately, for a destination-partitioned graph where the source
 * nodes' deg^-1/2 lives in another table (all-gathered from the other ranks): rows of the CSR are the
 * n_rows LOCAL targets, src_sorted holds GLOBAL source ids indexing dis_src. */
int pangnn_gcn_degree_f32(const int64_t* rowptr_dst, const int32_t* perm_dst, const float* edge_weight,
                          int64_t n_rows, float* deg_inv_sqrt, pangnn_stream_t stream);
int pangnn_gcn_edge_norm_f32(const int64_t* rowptr_dst, const int32_t* src_sorted, const int32_t* perm_dst,
                             const float* edge_weight, const float* dis_src, const float* dis_dst,
                             int64_t n_rows, int64_t num_edges, float* norm_sorted, float* norm_orig,
                             pangnn_stream_t stream);

/* out[i] = in[perm[i]]  (re-orders per-edge values into another CSR's order) */
int pangnn_permute_f32(const float* in, const int32_t* perm, float* out, int64_t n,
                       pangnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * propagate (k5+k6, and k5^T for backward; MessagePassing.propagate + GCNConv.message + 'add'
 * aggregation + bias, src/gnn.py:158,165):
 *   out[r, 0:F] (+)= bias[0:F] + sum_{e in [rowptr[r], rowptr[r+1])} val[e] * x[idx[e], 0:F]
 * val == NULL means 1.0; idx == NULL means the identity (row r sums x[rowptr[r] .. rowptr[r+1]));
 * bias == NULL means none; accumulate != 0 adds into `out`.
 * nnz = rowptr[n_rows] if known (a scheduling hint only: thin rows use a shallower unroll), else -1.
 * F in {16, 32, 64, 128, 256}; x/out/bias 16-byte aligned, ldx/ldo multiples of 4 floats.
 * No atomics: bitwise reproducible for a fixed structure.
 * ---------------------------------------------------------------------------------------- */
int pangnn_spmm_csr_f32(const int64_t* rowptr, const int32_t* idx, const float* val,
                        const float* x, int64_t ldx, int64_t n_src_rows,
                        const float* bias, float* out, int64_t ldo, int64_t n_rows,
                        int64_t nnz, int32_t F, int accumulate, pangnn_stream_t stream);
/* The same product with the gathered rows stored as bfloat16 (8 bytes per lane and row piece instead of 16): x is
 * [n_src_rows, ldx] bf16, 8-byte aligned; weights, accumulation, bias and the result are fp32.  This is what the
 * reference computes under `accelerate` bf16 mixed precision (PyG: bf16 x_j * fp32 edge weight -> fp32, scatter-add
 * into fp32), and the storage format SURVEY.md §8b lists for config 5.  F in {32, 64, 128, 256}. */
int pangnn_spmm_csr_bf16(const int64_t* rowptr, const int32_t* idx, const float* val, const void* x_bf16,
                         int64_t ldx, int64_t n_src_rows, const float* bias, float* out, int64_t ldo,
                         int64_t n_rows, int64_t nnz, int32_t F, int accumulate, pangnn_stream_t stream);
/* ... and as IEEE half (`--mixed_precision fp16`, /root/reference/src/setup.py:50: PyG's propagate under float16 autocast) */
int pangnn_spmm_csr_f16(const int64_t* rowptr, const int32_t* idx, const float* val, const void* x_f16,
                        int64_t ldx, int64_t n_src_rows, const float* bias, float* out, int64_t ldo,
                        int64_t n_rows, int64_t nnz, int32_t F, int accumulate, pangnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * edge decoder gather (k7; src/gnn.py:171-177):
 *   out[e, 0:D]   = z[src_e], out[e, D:2D] = z[dst_e], out[e, 2D] = extra[e] (if extra != NULL)
 * for e in [e_begin, e_begin + n_edges) of edge_index[2][ld]; out has leading dimension ldo.
 * ---------------------------------------------------------------------------------------- */
int pangnn_edge_gather_concat_f32(const float* z, int64_t ldz, int64_t num_nodes,
                                  const int64_t* edge_index, int64_t ld, int64_t e_begin,
                                  int64_t n_edges, const float* extra, float* out, int64_t ldo,
                                  int32_t D, pangnn_stream_t stream);

/* Re-associated first decoder layer: Linear(cat(z_s, z_d, w)) == P[s] + Q[d] + w*c with
 * P = z W_a^T, Q = z W_b^T + b (node-level GEMMs done by the caller).
 *   out[e, 0:D] = p[src_e] + q[dst_e] (+ extra[e] * cvec[0:D])
 * ---------------------------------------------------------------------------------------- */
int pangnn_edge_pair_add_f32(const float* p, const float* q, int64_t ldpq, int64_t num_nodes,
                             const int64_t* edge_index, int64_t ld, int64_t e_begin, int64_t n_edges,
                             const float* extra, const float* cvec, float* out, int64_t ldo,
                             int32_t D, pangnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Segment reductions over per-edge rows selected through a CSR permutation (backward of the edge
 * gathers above, and 'max' aggregation of src/convolution.py:7):
 *   sum : out[r, 0:F] (+)= sum_{k in row r} m[perm[k], col_off : col_off+F]   (m has n_m_rows rows)
 *   max : out[r, 0:F]   = max_{k in row r} m[perm[k], 0:F] (0 for empty rows), arg[r,f] = perm[k*]
 * ---------------------------------------------------------------------------------------- */
int pangnn_segment_sum_rows_f32(const int64_t* rowptr, const int32_t* perm, const float* m,
                                int64_t ldm, int64_t n_m_rows, int64_t col_off, float* out, int64_t ldo,
                                int64_t n_rows, int32_t F, int accumulate, pangnn_stream_t stream);
int pangnn_segment_max_rows_f32(const int64_t* rowptr, const int32_t* perm, const float* m,
                                int64_t ldm, float* out, int32_t* arg, int64_t ldo,
                                int64_t n_rows, int32_t F, pangnn_stream_t stream);
/* gm[arg[r,f], f] = g[r,f] for non-empty rows; gm must be zero-filled by the caller */
int pangnn_segment_max_bwd_f32(const float* g, const int32_t* arg, const int64_t* rowptr,
                               float* gm, int64_t ldm, int64_t ldo, int64_t n_rows, int32_t F,
                               pangnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused link decoder, node_dim D = 64 (k7 + the per-edge MLP, src/gnn.py:110-116,171-177):
 *   h1[e]   = relu(p[src_e] + q[dst_e] (+ extra[e] * cvec))     p = z W1[:, :D]^T, q = z W1[:, D:2D]^T + b1
 *   h2[e]   = relu(w2 h1[e] + b2)                               w2 [D,D] row-major [out][in]  (mlp.2)
 *   logits[e] = w3 . h2[e] + b3                                 w3 [D], b3 [1]                (mlp.4)
 * p and q are row-major with leading dimensions ldp / ldq (so both may be column windows of one
 * [N, 2D] product).  Edges are taken in the caller's order, e in [0, num_edges) of edge_index[2][ld].  No [E, D]
 * intermediate touches HBM in the forward pass.  fp32 MFMA (exact fp32 products and sums).
 *
 * Backward: given g_logits[E], recomputes the forward and returns
 *   g_h1[E, D] = dL/d(h1 pre-activation)  (the caller segment-sums it into g_p by source and g_q by
 *   target with pangnn_segment_sum_rows_f32), and the parameter gradients g_w2[D,D], g_b2[D],
 *   g_w3[D], g_b3[1], g_cvec[D] (nullable).  Reproducible: per-workgroup partial slabs in
 *   `workspace`, summed in a fixed order; no float atomics.
 *   part_buf / part_off (both NULL, or both set when the edge list is sorted by source): the kernel also
 *   sums the g_h1 rows of every (32-edge tile, source) run into part_buf[part, 0:D]; part_off[tile] is the
 *   index of the tile's first part, i.e. the number of k < 32*tile with k % 32 == 0 or src_k != src_{k-1}.
 *   g_p[s] is then the sum of the consecutive parts of source s (pangnn_spmm_csr_f32 with idx == NULL).
 *   precision: must be 0 here = every product on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains): the strict-fp32
 *              mode.  The bf16 matrix-pipe mode (fp32-level error, the default of the Python layer) is the
 *              pangnn_decoder_train_f32 + pangnn_decoder_dgrad_f32 pair below; it produces no [E, D] tensor.
 * ---------------------------------------------------------------------------------------- */
int pangnn_decoder_mlp_fwd_f32(const float* p, int64_t ldp, const float* q, int64_t ldq, int64_t num_nodes,
                               const int64_t* edge_index, int64_t ld, int64_t num_edges,
                               const float* extra, const float* cvec, const float* w2, const float* b2,
                               const float* w3, const float* b3, int32_t D, float* logits,
                               pangnn_stream_t stream);
/* the same forward with the matrix-pipe mode of the training kernels (`precision`): 0 = f32 MFMA (bit-identical to
 * the logits of pangnn_decoder_mlp_loss_f32), 1 = bf16 matrix pipe with three-way split operands (fp32-level
 * error; bit-identical to the logits of pangnn_decoder_train_f32, about half the time of mode 0) */
int pangnn_decoder_mlp_infer_f32(const float* p, int64_t ldp, const float* q, int64_t ldq, int64_t num_nodes,
                                 const int64_t* edge_index, int64_t ld, int64_t num_edges,
                                 const float* extra, const float* cvec, const float* w2, const float* b2,
                                 const float* w3, const float* b3, int32_t D, float* logits,
                                 int32_t precision, pangnn_stream_t stream);
size_t pangnn_decoder_mlp_bwd_workspace_bytes(int64_t num_edges);
int pangnn_decoder_mlp_bwd_f32(const float* p, int64_t ldp, const float* q, int64_t ldq, int64_t num_nodes,
                               const int64_t* edge_index, int64_t ld, int64_t num_edges,
                               const float* extra, const float* cvec, const float* w2, const float* b2,
                               const float* w3, const float* b3, int32_t D, const float* g_logits,
                               float* g_h1, float* g_w2, float* g_b2, float* g_w3, float* g_b3,
                               float* g_cvec, float* part_buf, const int32_t* part_off, int32_t precision,
                               void* workspace, size_t workspace_bytes, pangnn_stream_t stream);

/* Training form of the decoder: logits, BCEWithLogits(pos_weight) mean loss (denominator `denom`) AND every
 * gradient in ONE pass over the edges — the logits of a tile come out of the backward's recomputed first
 * product, so forward + criterion + backward (pangnn.py:200-207) do not need three kernels and the
 * forward's product is not computed twice.  Outputs as pangnn_decoder_mlp_bwd_f32 plus logits[E], loss[1];
 * all gradients are those of `loss` (upstream gradient 1). */
int pangnn_decoder_mlp_loss_f32(const float* p, int64_t ldp, const float* q, int64_t ldq, int64_t num_nodes,
                                const int64_t* edge_index, int64_t ld, int64_t num_edges, const float* extra,
                                const float* cvec, const float* w2, const float* b2, const float* w3,
                                const float* b3, int32_t D, const float* y, const float* pos_weight,
                                int64_t denom, float* logits, float* loss, float* g_h1, float* g_w2,
                                float* g_b2, float* g_w3, float* g_b3, float* g_cvec, float* part_buf,
                                const int32_t* part_off, int32_t precision, void* workspace,
                                size_t workspace_bytes, pangnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Two-wave-per-SIMD training decoder (csrc/decoder16.hip) — what `precision = 1` training uses.
 * Replaces forward + criterion + backward of src/gnn.py:171-177 / pangnn.py:200-207 with TWO passes and
 * no [E, D] tensor:
 *
 * pangnn_decoder_train_f32  ("S"): one pass over the edges in the caller's order.  Exactly one of
 *   y        (fused loss: logits[E], loss[1] = mean BCEWithLogits(pos_weight) over `denom` edges are outputs and
 *             dL/dlogit is produced in place), or
 *   g_logits (given dL/dlogits[E]; logits nullable)
 *   is set.  Outputs: the parameter gradients g_w2[D,D], g_w3[D], g_b3[1], g_cvec[D] (nullable) — dL/db2 comes
 *   out of the dgrad pass (it needs only the records; keeping its per-lane partial sums out of this kernel is
 *   what lets two waves share a SIMD without spilling),
 *   rec[E][8] uint32 (required): per edge {4 dwords of relu masks — the h2 mask in the low byte of each half, the h1
 *             mask in the high byte —, dL/dlogit_e replicated into dwords 4 .. 7} for the dgrad pass, whose lane group g
 *             reads mask dword g and dL/dlogit through one address (offsets 0 and 16),
 *   part_buf / part_off (both NULL or both set; edge list sorted by source): sums of dL/dh1 over every
 *   (chunk, source) run, where a chunk is pangnn_decoder_chunk_tiles_for(num_edges) (16 for E >= 1e6, down to 1 for short lists) consecutive 32-edge tiles walked by
 *   one wave with the open run carried from tile to tile: part_off[c] = index of chunk c's first part row, a new
 *   part starts at every chunk start and at every change of the key (pangnn_decoder_mlp_bwd_f32 uses the same layout
 *   with one-tile chunks).  Chunks are a property of the edge list, not of the launch: results do not depend on the
 *   grid.
 *   Arithmetic: every product on the bf16 matrix pipe with fp32 accumulation and fp32-exact operand handling —
 *   W2 h1 with both operands split into three bf16 terms (6 partial products); dL/dh1 = m1 g_e (m2^T W2') and
 *   dL/dW2 = diag(w3) m2 (g_e h1) with the relu mask m2 as the EXACT bf16 operand and the other operand split
 *   three ways.  The last tile is padded by clamping edge ids (one tile body, no divergent matrix operands).
 *   workspace: pangnn_decoder_train_workspace_bytes().
 *
 * pangnn_decoder_dgrad_f32  ("T"): dL/dh1 summed over runs of equal keys in a permuted edge order, from `rec`:
 *   position k of the order is edge perm[k] (NULL = identity) with run key keys[k] (non-decreasing, e.g. the
 *   target id in by-target CSR order); part_buf / part_off as above for THIS order.  dL/dQ[t] is then the sum
 *   of the consecutive parts of target t (pangnn_spmm_csr_f32 with idx == NULL).  (dL/dcvec always comes out of
 *   pangnn_decoder_train_*; round 3's instances that recomputed it here were never launched and are gone.)
 *   g_b2[D] (nullable) = dL/db2 = w3[j] sum_e g_e [h2[j][e] > 0] — ask for it in exactly one dgrad call per step.
 *   part_buf / part_off / keys may all be NULL to get the parameter sum alone.  workspace (with g_b2):
 *   pangnn_decoder_dgrad_workspace_bytes().
 * live_edges (both entry points; nullable, DEVICE int64[1]): the list is a fixed-shape batch whose first *live_edges edges
 *   are real and whose tail is padding (a mini-batch collated by pangnn_collate_subgraphs_padded so that one captured
 *   HIP graph serves every batch): in S the padded edges keep their own logit / record slots but get dL/dlogit = 0 and
 *   the fused loss is the mean over *live_edges edges (`denom` is ignored); in T the positions >= *live_edges of the
 *   order are the padding (the pads must sort last: their endpoints are the largest node id).  NULL: every edge is real.
 * Both are reproducible (fixed-order sums, no float atomics).
 * ---------------------------------------------------------------------------------------- */
int    pangnn_decoder_chunk_tiles(void);
/* ... as a function of the list: pangnn_decoder_chunk_tiles() (16) for lists of >= 2048 * 16 tiles, halved until the list
 * has at least 2048 chunks (1 for a mini-batch): the S / T kernels derive the same value from num_edges, and `part_off` /
 * `part_buf` must be laid out for it. */
int    pangnn_decoder_chunk_tiles_for(int64_t num_edges);
size_t pangnn_decoder_train_workspace_bytes(void);
int pangnn_decoder_train_f32(const float* p, int64_t ldp, const float* q, int64_t ldq, int64_t num_nodes,
                             const int64_t* edge_index, int64_t ld, int64_t num_edges, const float* extra,
                             const float* cvec, const float* w2, const float* b2, const float* w3,
                             const float* b3, int32_t D, const float* y, const float* pos_weight,
                             int64_t denom, const float* g_logits, float* logits, float* loss, uint32_t* rec,
                             float* part_buf, const int32_t* part_off, float* g_w2, float* g_w3, float* g_b3,
                             float* g_cvec, const int64_t* live_edges, void* workspace, size_t workspace_bytes,
                             pangnn_stream_t stream);
/* bf16-storage mode of the gather (config 5, bf16 mixed precision: mlp[0] is an autocast Linear, src/gnn.py:173, so
 * its node-level halves P and Q are bf16 tensors): pq_dtype = PANGNN_DTYPE_BF16 reads p and q as bfloat16 rows
 * (ldp / ldq in elements, multiples of 8; half the gather bytes).  bf16 -> f32 is exact, so logits, loss and all
 * gradients equal, bit for bit, those of the f32 entry points on the up-converted tables.  Gradients stay fp32.
 * pangnn_decoder_train_f32 / the precision-1 pangnn_decoder_mlp_infer_f32 are these with PANGNN_DTYPE_F32. */
int pangnn_decoder_train_mixed(const void* p, int64_t ldp, const void* q, int64_t ldq, int32_t pq_dtype,
                               int64_t num_nodes, const int64_t* edge_index, int64_t ld, int64_t num_edges,
                               const float* extra, const float* cvec, const float* w2, const float* b2,
                               const float* w3, const float* b3, int32_t D, const float* y, const float* pos_weight,
                               int64_t denom, const float* g_logits, float* logits, float* loss, uint32_t* rec,
                               float* part_buf, const int32_t* part_off, float* g_w2, float* g_w3, float* g_b3,
                               float* g_cvec, const int64_t* live_edges, void* workspace, size_t workspace_bytes,
                               pangnn_stream_t stream);
int pangnn_decoder_mlp_infer_mixed(const void* p, int64_t ldp, const void* q, int64_t ldq, int32_t pq_dtype,
                                   int64_t num_nodes, const int64_t* edge_index, int64_t ld, int64_t num_edges,
                                   const float* extra, const float* cvec, const float* w2, const float* b2,
                                   const float* w3, const float* b3, int32_t D, float* logits,
                                   pangnn_stream_t stream);
size_t pangnn_decoder_dgrad_workspace_bytes(void);
int pangnn_decoder_dgrad_f32(const uint32_t* rec, const int32_t* perm, const int32_t* keys, const float* w2,
                             const float* w3, int64_t num_edges, float* part_buf, const int32_t* part_off, float* g_b2,
                             const int64_t* live_edges, void* workspace, size_t workspace_bytes, pangnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Node-level dense layers with a short inner dimension, K (in) and M (out) in {64, 128}
 * (GCNConv.lin = k4 of SURVEY.md §2.2, the decoder's node-level P|Q product, and their backward):
 *   fwd  : y[n, 0:M]  = x[n, 0:K] . w[M,K]^T (+ bias[M])          (dL/dx = g . w is fwd with w^T)
 *   wgrad: gw[M,K]    = g[N,M]^T . x[N,K],  gb[M] = sum_n g[n,:]  (gb nullable)
 * fp32 storage and accumulation; the products run on the bf16 matrix pipe with both operands split into three exact
 * bf16 terms (six partial products: the error against an fp64 evaluation is that of an fp32 FMA chain; the 128 x 128 shape
 * forward and builds with -DPANGNN_LIN_F32_MFMA use v_mfma_f32_32x32x2_f32; the 128 x 128 weight gradient runs as two
 * 64-column halves of g).  HBM-bound streaming kernels; wgrad is slab-reduced (reproducible).
 * pangnn_linear_supported(K, M, wgrad) tells the caller whether a shape is covered.
 * ---------------------------------------------------------------------------------------- */
int    pangnn_linear_supported(int32_t K, int32_t M, int wgrad);
int    pangnn_linear_fwd_f32(const float* x, int64_t ldx, const float* w, const float* bias, float* y,
                             int64_t ldy, int64_t n, int32_t K, int32_t M, pangnn_stream_t stream);
size_t pangnn_linear_wgrad_workspace_bytes(int32_t K, int32_t M);
int    pangnn_linear_wgrad_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t n,
                               int32_t K, int32_t M, float* gw, float* gb, void* workspace,
                               size_t workspace_bytes, pangnn_stream_t stream);
/* The same layers with the encoder's activation folded in (src/gnn.py:108,129-166: h = ELU(conv(...)) is always
 * followed by exactly one dense layer — GCNConv.lin of the next conv, linear_out, or mlp[0] — so ELU(x) never has
 * to exist in HBM, and neither does its backward pass as a kernel of its own):
 *   act_fwd   in_act = 1 : y = ELU(x) . w^T (+ bias)          (x is the PRE-activation; alpha = 1)
 *             gate != NULL: y = (x . w^T) * ELU'(gate[n, 0:M]) — dL/d(pre-activation) straight from the dL/dx
 *                           product of the following layer (gate = that pre-activation); excludes in_act
 *   act_wgrad in_act = 1 : gw = g^T . ELU(x)
 * in_act = 0, gate = NULL are the plain entry points above. */
int    pangnn_linear_act_fwd_f32(const float* x, int64_t ldx, const float* w, const float* bias, float* y,
                                 int64_t ldy, int64_t n, int32_t K, int32_t M, int32_t in_act,
                                 const float* gate, int64_t ldgate, pangnn_stream_t stream);
int    pangnn_linear_act_wgrad_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t n,
                                   int32_t K, int32_t M, int32_t in_act, float* gw, float* gb,
                                   void* workspace, size_t workspace_bytes, pangnn_stream_t stream);
/* bf16-storage modes (config 5: `accelerate` bf16 mixed precision, pangnn.py:25 — the reference's autocast Linear
 * layers, src/gnn.py:93,111,173, hand bf16 tensors to the next layer).  x / y / gate / g may each be stored as
 * bfloat16 (PANGNN_DTYPE_*; ld in ELEMENTS; rows start on 8 bytes); products and sums are the same exact-fp32 MFMA
 * chains as above, a bf16 result is rounded to nearest even once, on store.  w, bias, gw, gb stay fp32 (the
 * parameters are fp32 under autocast).  A gated product is the gradient of its gate tensor and is stored like it:
 * gate_dtype must equal y_dtype.  The *_f32 entry points are these with every dtype = PANGNN_DTYPE_F32. */
int    pangnn_linear_act_fwd_mixed(const void* x, int32_t x_dtype, int64_t ldx, const float* w, const float* bias,
                                   void* y, int32_t y_dtype, int64_t ldy, int64_t n, int32_t K, int32_t M,
                                   int32_t in_act, const void* gate, int32_t gate_dtype, int64_t ldgate,
                                   pangnn_stream_t stream);
int    pangnn_linear_act_wgrad_mixed(const void* g, int32_t g_dtype, int64_t ldg, const void* x, int32_t x_dtype,
                                     int64_t ldx, int64_t n, int32_t K, int32_t M, int32_t in_act, float* gw,
                                     float* gb, void* workspace, size_t workspace_bytes, pangnn_stream_t stream);
/* dL/dx of the dense layer y = act(x) . w^T straight from the layer's own weight w [M][K] row-major (the kernel stages w
 * through the strides of its transpose — no transposed copy of w per step, one launch less in a launch-bound mini-batch step):
 *   gx [n][K] = g [n][M] . w   (* ELU'(gate[n, 0:K]) when gate != NULL: dL/d(pre-activation), as act_fwd's gate)
 * Storage types / alignment as pangnn_linear_act_fwd_mixed with x = g, y = gx.  (autograd of nn.Linear / GCNConv.lin:
 * src/gnn.py:104,110-116; pangnn.py:207.) */
int    pangnn_linear_dgrad_mixed(const void* g, int32_t g_dtype, int64_t ldg, const float* w, void* gx, int32_t gx_dtype,
                                 int64_t ldgx, int64_t n, int32_t K, int32_t M, const void* gate, int32_t gate_dtype,
                                 int64_t ldgate, pangnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Confusion counts of thresholded link predictions, accumulated on the device:
 *   counts[2*label + prediction] += 1,  prediction = (score >= threshold),
 *   score = sigmoid(scores[i]) if apply_sigmoid else scores[i];  label = labels[i] > 0.5.
 * Replaces  probabilities = torch.sigmoid(output); (probabilities >= binary_th).int();
 *           BinaryConfusionMatrix.update(prediction, labels)      (pangnn.py:218-222, 257-262)
 * counts = {tn, fp, fn, tp} (torchmetrics layout [[tn, fp], [fn, tp]]), int64, caller-zeroed, never reset here.
 * ---------------------------------------------------------------------------------------- */
int    pangnn_confusion_update_f32(const float* scores, const float* labels, int64_t n, float threshold,
                                   int apply_sigmoid, int64_t* counts, pangnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * out[0, c] = sum_n r[n] g[n, c], out[1, c] = sum_n s[n] g[n, c]  (out [2, F], F divides 256).
 * Backward of the scalar-feature embedding feeding the first GCN layer (src/gnn.py:97,125,158): the layer's
 * input is h0 = x w^T + 1 b^T, so dL/dw = (A_hat x)^T g and dL/db = (A_hat 1)^T g with g = dL/d(A_hat h0);
 * r = A_hat x and s = A_hat 1 are node vectors computed once per graph.  Reproducible two-stage sum.
 * ---------------------------------------------------------------------------------------- */
/* (r, s) = (A_hat x, A_hat 1) over one CSR order in ONE launch: r[i] = sum_{k in row i} val[k] x[other[k]],
 * s[i] = sum_{k in row i} val[k] (val = NULL: unit weights).  What the scalar-feature first layer needs of a graph
 * (src/gnn.py:97,125,158 by linearity: functional._EmbedConvIn); a fresh mini-batch pays it every step. */
int    pangnn_node_actions_f32(const int64_t* rowptr, const int32_t* other, const float* val, const float* x,
                               int64_t n_rows, float* r, float* s, pangnn_stream_t stream);
/* Operands of the decoder's re-associated first layer from mlp[0] = Linear(2D (+1), D) (src/gnn.py:110,173-175) in one
 * launch: w_pq [2D][D] = [W[:, :D] ; W[:, D:2D]], b_pq [2D] = [0 ; b], cvec [D] = W[:, 2D] (skip != 0).  w: [D] rows of
 * ldw >= 2D (+1) floats.  Plain data movement. */
int    pangnn_pq_operands_f32(const float* w, int64_t ldw, const float* b, int32_t d, int skip, float* w_pq,
                              float* b_pq, float* cvec, pangnn_stream_t stream);
/* out[c] = sum_n g[n, c] of a SHORT matrix (the node rows of a mini-batch; g f32 or bfloat16, ld in elements) in one
 * launch, fixed order of additions — the bias gradient of GCNConv (PyG: out + bias; src/gnn.py:165).  Long matrices: the
 * two-stage sums above. */
int    pangnn_colsum_small(const void* g, int32_t g_dtype, int64_t ldg, int64_t n, int32_t F, float* out,
                           pangnn_stream_t stream);
size_t pangnn_weighted_colsum_workspace_bytes(int32_t F);
int    pangnn_weighted_colsum_f32(const float* g, int64_t ldg, const float* r, const float* s, int64_t n,
                                  int32_t F, float* out, void* workspace, size_t workspace_bytes,
                                  pangnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Propagate over the positional-neighbour graph of a whole genome set (src/dataset.py:356-361: edges (i, j) for
 * j in [i - k, i + k] within [0, N), self loops included, unit weights — a band matrix; GCNConv at src/gnn.py:165):
 *   out[t, :] = bias + sum_{s = t-k .. t+k} dis[s] dis[t] x[s, :]       dis = deg^-1/2 (pangnn_gcn_norm_f32)
 * The same sums in the same order as pangnn_spmm_csr_f32 over that edge list (bit-identical), without index / weight
 * arrays.  The band is symmetric: the transposed propagate is the same call.  colsum (optional, [F]): column sums of
 * x — the bias gradient when x is the upstream gradient (backward of the layer in one pass).  x f32 or bf16 rows.
 * ---------------------------------------------------------------------------------------- */
size_t pangnn_band_propagate_workspace_bytes(int32_t F);
int    pangnn_band_propagate(const void* x, int32_t x_dtype, int64_t ldx, const float* dis, const float* bias, float* out,
                             int64_t ldo, int64_t n, int32_t F, int32_t k, float* colsum, void* workspace,
                             size_t workspace_bytes, pangnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The first GCN layer of the scalar-feature model as a rank-2 matrix (replaces, for `conv_in(embedding(x))`
 * of src/gnn.py:97,125,156-158 / :143-146 / :128-131, the embedding Linear(1, D), the propagate and GCNConv.lin):
 *   h0 = x w^T + 1 b^T (one scalar x per node)  =>  A_hat (h0) W^T + b_in = r a^T + s c^T + b_in,
 *   r = A_hat x, s = A_hat 1 (node vectors, once per graph: pangnn_spmm_csr_f32 on a 2-column table),
 *   a = W w, c = W b (H-vectors, per step; propagate and dense layer commute, so the same holds for D >= H).
 * pangnn_rank2_rows:       out[n, :] = r[n] a + s[n] c + bias   (out [N, F] stored as f32 or bf16, F % 4 == 0)
 * pangnn_weighted_colsum3: out[0] = sum_n r[n] g[n,:], out[1] = sum_n s[n] g[n,:], out[2] = sum_n g[n,:]  (out [3, F]):
 *   everything the layer's backward needs — dL/da, dL/dc, dL/db_in — from which dL/dW = dL/da w^T + dL/dc b^T,
 *   dL/dw = W^T dL/da, dL/db = W^T dL/dc.  g stored as f32 or bf16; reproducible two-stage sum.
 * ---------------------------------------------------------------------------------------- */
/* The same layer with the parameter algebra inside the kernels (one launch forward, three backward — what the
 * model uses): a = W w and c = W b are formed per workgroup in LDS; the backward runs pangnn_weighted_colsum3 and then
 * forms dL/dW_in [H, D], dL/dw [D], dL/db [D], dL/db_in [H] (nullable) in one small kernel.
 * workspace: pangnn_embed_conv_in_grads_workspace_bytes(H). */
int    pangnn_embed_conv_in_rows(const float* r, const float* s, const float* w_emb, const float* b_emb, const float* w_in,
                                 const float* b_in, int32_t D, void* out, int32_t out_dtype, int64_t ldo, int64_t n,
                                 int32_t H, pangnn_stream_t stream);
size_t pangnn_embed_conv_in_grads_workspace_bytes(int32_t H);
int    pangnn_embed_conv_in_grads(const void* g, int32_t g_dtype, int64_t ldg, const float* r, const float* s, int64_t n,
                                  const float* w_emb, const float* b_emb, const float* w_in, int32_t D, int32_t H,
                                  float* g_w_emb, float* g_b_emb, float* g_w_in, float* g_b_in, void* workspace,
                                  size_t workspace_bytes, pangnn_stream_t stream);

/* The same layer FUSED into the dense layer that follows it (GCNConv.lin of conv_out, gnn.py:164-166 / linear_out,
 * gnn.py:147): y = ELU(h) W_out^T (+ bias_out) with the [n, H] rows h = r a^T + s c^T + b_in generated inside the kernels
 * (8 bytes per row read instead of 4 H; the values are pangnn_embed_conv_in_rows', bit for bit).  Backward: dL/dW_out [M, H],
 * dL/dbias_out [M] (nullable) and sums [3, H] = [r s 1]^T dL/dh — dL/dh itself is never written — which
 * pangnn_embed_conv_in_grads_from_sums turns into the first layer's parameter gradients (as pangnn_embed_conv_in_grads
 * does after its own column sums).  (H, M) as pangnn_linear_supported(H, M, 1); fp32 storage. */
int    pangnn_embed_linear_supported(int32_t H, int32_t M);
int    pangnn_embed_linear_fwd(const float* r, const float* s, int64_t n, const float* w_emb, const float* b_emb,
                               const float* w_in, const float* b_in, int32_t D, int32_t H, const float* w_out,
                               const float* bias_out, int32_t M, float* y, int64_t ldy, pangnn_stream_t stream);
size_t pangnn_embed_linear_bwd_workspace_bytes(int32_t H, int32_t M);
int    pangnn_embed_linear_bwd(const float* g, int64_t ldg, const float* r, const float* s, int64_t n, const float* w_emb,
                               const float* b_emb, const float* w_in, const float* b_in, int32_t D, int32_t H,
                               const float* w_out, int32_t M, float* g_w_out, float* g_b_out, float* sums, void* workspace,
                               size_t workspace_bytes, pangnn_stream_t stream);
int    pangnn_embed_conv_in_grads_from_sums(const float* sums, const float* w_emb, const float* b_emb, const float* w_in,
                                            int32_t D, int32_t H, float* g_w_emb, float* g_b_emb, float* g_w_in,
                                            float* g_b_in, pangnn_stream_t stream);
int    pangnn_rank2_rows(const float* r, const float* s, const float* a, const float* c, const float* bias, void* out,
                         int32_t out_dtype, int64_t ldo, int64_t n, int32_t F, pangnn_stream_t stream);
size_t pangnn_weighted_colsum3_workspace_bytes(int32_t F);
int    pangnn_weighted_colsum3(const void* g, int32_t g_dtype, int64_t ldg, const float* r, const float* s, int64_t n,
                               int32_t F, float* out, void* workspace, size_t workspace_bytes, pangnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * BCEWithLogitsLoss(pos_weight), mean reduction (pangnn.py:98,203), loss and dL/dlogits in one pass:
 *   loss[0]     = 1/denom * sum_i (1-y_i) x_i + (1 + (pw-1) y_i) softplus(-x_i)
 *   g_logits[i] = 1/denom * ((1-y_i) - (1 + (pw-1) y_i) sigmoid(-x_i))
 * pos_weight is a DEVICE scalar (nullable = 1).  denom = number of edges of the whole job (differs from
 * n on a partitioned shard).  Reproducible two-stage sum.
 * ---------------------------------------------------------------------------------------- */
size_t pangnn_bce_logits_workspace_bytes(void);
int    pangnn_bce_logits_f32(const float* logits, const float* y, const float* pos_weight, int64_t n,
                             int64_t denom, float* loss, float* g_logits, void* workspace,
                             size_t workspace_bytes, pangnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * normalize_sim_scores (src/preprocessing.py:454-548) over a relation sorted by (source gene, candidate genome):
 * segment k = items [rowptr[k], rowptr[k+1]).  Per segment p = softmax(score / t) (one candidate: p = 1),
 * q = -10 log10(clip(1 - p, epsilon, 1 - epsilon)) + pseudo_count, float64 as in the reference.  One wavefront
 * per segment, fixed reduction order.  
 * ---------------------------------------------------------------------------------------- */
int    pangnn_softmax_qscore_f64(const int64_t* rowptr, const double* score, int64_t num_segments,
                                 int64_t num_items, double t, double epsilon, double pseudo_count, double* q,
                                 pangnn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * `accelerator.backward(loss)` (pangnn.py:207) behind the one-pass training decoder: that pass has every gradient of the
 * step's loss in memory before autograd's backward starts, so backward only has to multiply them by the upstream
 * gradient of the loss — which `loss.backward()` / accelerate's `loss / gradient_accumulation_steps` make EXACTLY 1,
 * a fact the host cannot read without a synchronisation.  This entry scales up to PANGNN_SCALE_MAX_TENSORS fp32 buffers in
 * place by the DEVICE scalar scale[0] and does nothing at all (no loads beyond the scalar, no stores) when it is 1.0f.
 * ptrs / counts are HOST arrays of n_tensors device pointers / element counts (copied into the launch arguments).
 * ---------------------------------------------------------------------------------------- */
#define PANGNN_SCALE_MAX_TENSORS 8
int    pangnn_scale_unless_one_f32(float* const* ptrs, const int64_t* counts, int32_t n_tensors, const float* scale,
                                   pangnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PANGNN_HIP_H */
