/* Plain-C restatement of the propagate stage — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * PARITY UNPINNED (arithmetic): follows PyG's documented gcn_norm + MessagePassing.propagate
 * semantics for GCNConv(add_self_loops=False) as called at /root/reference/src/gnn.py:158,165
 * (torch_geometric is not in the reference tree; see oracle/gcn_oracle.py header).
 * Sequential loops in edge order, fp32 accumulators: the summation order a single-threaded
 * scatter_add_/index_add_ would use.  Checked against oracle/gcn_oracle.py in tests.
 *
 * Build: gcc -O2 -shared -fPIC oracle/propagate_oracle.c -o oracle/_build/libpropagate_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* norm[e] = deg^-1/2[src] * w * deg^-1/2[dst], deg = weighted in-degree (w == NULL -> ones) */
void oracle_gcn_norm_f32(const int64_t* edge_index, int64_t num_edges, int64_t num_nodes,
                         const float* w, float* norm) {
  const int64_t* row = edge_index;
  const int64_t* col = edge_index + num_edges;
  float* deg = (float*)calloc((size_t)(num_nodes > 0 ? num_nodes : 1), sizeof(float));
  for (int64_t e = 0; e < num_edges; ++e) deg[col[e]] += w ? w[e] : 1.0f;
  for (int64_t i = 0; i < num_nodes; ++i) {
    float d = powf(deg[i], -0.5f);
    deg[i] = isinf(d) ? 0.0f : d;
  }
  for (int64_t e = 0; e < num_edges; ++e) norm[e] = deg[row[e]] * (w ? w[e] : 1.0f) * deg[col[e]];
  free(deg);
}

/* out[col[e], :] += norm[e] * x[row[e], :]  (+ bias) */
void oracle_propagate_f32(const int64_t* edge_index, int64_t num_edges, int64_t num_nodes,
                          const float* norm, const float* x, int64_t f, const float* bias,
                          float* out) {
  const int64_t* row = edge_index;
  const int64_t* col = edge_index + num_edges;
  memset(out, 0, (size_t)(num_nodes * f) * sizeof(float));
  for (int64_t e = 0; e < num_edges; ++e) {
    const float* xr = x + row[e] * f;
    float* o = out + col[e] * f;
    const float v = norm[e];
    for (int64_t k = 0; k < f; ++k) o[k] += v * xr[k];
  }
  if (bias)
    for (int64_t i = 0; i < num_nodes; ++i)
      for (int64_t k = 0; k < f; ++k) out[i * f + k] += bias[k];
}
