// The per-step operators that READ A GRAPH, in C++ end to end (round 5):
//   gcn_propagate, embed_conv_in, embed_conv_in_linear, embed_propagate, decoder_loss, decoder_mlp and their backward ops
// — schema, HIP ("CUDA" key) implementation and autograd formula (torch::autograd::Function under the Autograd key), like
// pangnn::linear.  Rounds 3-4 registered them from Python over ctypes because the structure cache lived in Python.
//
// A graph enters an op as its `edge_index` (and `edge_weight`) TENSOR — no opaque handles in the schemas, so FakeTensor
// tracing and torch.compile see ordinary ops — and the implementation finds what was built for that tensor in the
// STRUCTURE REGISTRY below: both CSR orders, the decoder kernels' run-sum plans, the band width, the degree
// normalisations per weight tensor and the first layer's node vectors per feature tensor.  The registry is keyed on
// (data pointer, version counter, edge count, target-row count, device) and guarded by a weak reference to the tensor's
// storage (a freed tensor's address can be handed to a new one).  What it holds is BUILT by pangnn_amd/graph.py with the C
// entry points of include/pangnn_hip.h (pangnn_csr_build, pangnn_structure_small, pangnn_gcn_norm_f32, ...) and pushed here
// once (pangnn::_register_*); on a miss — an op called on tensors nobody prepared, e.g. a compiled graph replayed on a fresh
// batch — the op calls pangnn::_prepare_structure, the one hook implemented in Python, which builds and pushes what is
// missing, then carries on.  In the steady state a step runs no Python between the dispatcher and the kernels.
#include <map>
#include <mutex>
#include <vector>

#include "torch_common.h"

using namespace pangnn_torch;

namespace {

// ---------------------------------------------------------------------------------------------------------------
// structure registry
// ---------------------------------------------------------------------------------------------------------------
enum Need : int64_t {          // bits of pangnn::_prepare_structure(..., need); mirrored in pangnn_amd/graph.py (NEED_*)
  kByDst = 1, kBySrc = 2, kBand = 4, kRunsum = 8, kPlanDst = 16, kPlanSrc = 32, kNorm = 64, kNormSrc = 128, kActions = 256,
  kEntry = 512                 // the structure itself (edge count, contiguous edge list): what an op that needs no table asks for
};

struct Csr { at::Tensor rowptr, other, perm, seg_ptr, parts_rowptr; };     // seg_ptr / parts_rowptr: long rows (graph.CSR.long_rows)
struct Plan { at::Tensor part_off, part_rowptr, keys; int64_t n_parts = 0; };

struct Ident {                 // identity of a tensor: storage (weak), address, version
  c10::optional<c10::weak_intrusive_ptr<c10::StorageImpl>> storage;
  const void* ptr = nullptr;
  int64_t version = -1, numel = -1;
  bool set = false;
  static Ident of(const at::Tensor& t) {
    Ident i;
    i.storage.emplace(t.storage().getWeakStorageImpl());
    i.ptr = t.data_ptr();
    i.version = (int64_t)t._version();
    i.numel = t.numel();
    i.set = true;
    return i;
  }
  bool is(const at::Tensor& t) const {
    if (!set || t.data_ptr() != ptr || (int64_t)t._version() != version || t.numel() != numel) return false;
    if (!storage.has_value() || storage->expired()) return false;
    return storage->lock().get() == t.storage().unsafeGetStorageImpl();
  }
  bool alive() const { return set && storage.has_value() && !storage->expired(); }
};

struct Actions { Ident x; at::Tensor r, s; };
struct Norm {
  bool unit = true;            // edge_weight = None
  Ident weight;
  at::Tensor dis, by_dst, orig, by_src;
  Actions actions;             // one slot, like graph.py's cache (a static graph has one feature tensor)
};

struct Entry {
  Ident key;
  int64_t n_dst = 0, n_src = 0, num_edges = 0;
  int device = -1;
  at::Tensor ei;               // contiguous int64 [2, E] the kernels read (the caller's tensor when it is contiguous)
  Csr by_dst, by_src;
  int band = -1, sorted_by_src = -1;           // -1: unknown
  std::map<std::pair<int, int>, Plan> plans;   // (kind: 0 run sums in edge order, 1 by target, 2 by source; chunk tiles)
  std::vector<Norm> norms;
  uint64_t stamp = 0;
};

constexpr size_t kMaxEntries = 32;
std::mutex g_mu;
std::vector<Entry> g_entries;
uint64_t g_clock = 0;

Entry* find_locked(const at::Tensor& edge_index, int64_t n_dst) {
  for (auto& e : g_entries)
    if (e.n_dst == n_dst && e.device == edge_index.device().index() && e.key.is(edge_index)) {
      e.stamp = ++g_clock;
      return &e;
    }
  return nullptr;
}

Entry& find_or_add_locked(const at::Tensor& edge_index, int64_t n_dst, int64_t n_src) {
  if (Entry* e = find_locked(edge_index, n_dst)) return *e;
  // drop entries whose key tensor died (their components would otherwise pin device memory), then the least recently used
  for (size_t i = 0; i < g_entries.size();)
    if (!g_entries[i].key.alive()) g_entries.erase(g_entries.begin() + i); else ++i;
  if (g_entries.size() >= kMaxEntries) {
    size_t lru = 0;
    for (size_t i = 1; i < g_entries.size(); ++i)
      if (g_entries[i].stamp < g_entries[lru].stamp) lru = i;
    g_entries.erase(g_entries.begin() + lru);
  }
  Entry e;
  e.key = Ident::of(edge_index);
  e.n_dst = n_dst;
  e.n_src = n_src;
  e.num_edges = edge_index.size(1);
  e.device = edge_index.device().index();
  e.stamp = ++g_clock;
  g_entries.push_back(std::move(e));
  return g_entries.back();
}

Norm* find_norm(Entry& e, const c10::optional<at::Tensor>& w) {
  const bool unit = !(w.has_value() && w->defined());
  for (auto& n : e.norms)
    if (unit ? n.unit : (!n.unit && n.weight.is(*w))) return &n;
  return nullptr;
}

void check_edge_index(const char* op, const at::Tensor& edge_index) {
  on_gpu(edge_index, "edge_index");
  TORCH_CHECK(edge_index.scalar_type() == at::kLong && edge_index.dim() == 2 && edge_index.size(0) == 2, "pangnn::", op,
              ": edge_index must be int64 [2, E]");
}

// -- registration ops (called by pangnn_amd/graph.py once per built component) -----------------------------------
void register_structure(const at::Tensor& edge_index, int64_t n_dst, int64_t n_src, const at::Tensor& ei_contig,
                        at::TensorList by_dst, at::TensorList by_src, int64_t band, int64_t sorted_by_src) {
  check_edge_index("_register_structure", edge_index);
  TORCH_CHECK(by_dst.size() == 0 || by_dst.size() == 3 || by_dst.size() == 5,
              "pangnn::_register_structure: by_dst is [] or [rowptr, other, perm (, seg_ptr, parts_rowptr)]");
  TORCH_CHECK(by_src.size() == 0 || by_src.size() == 3 || by_src.size() == 5,
              "pangnn::_register_structure: by_src is [] or [rowptr, other, perm (, seg_ptr, parts_rowptr)]");
  const int64_t e = edge_index.size(1);
  auto check = [&](at::TensorList c, int64_t rows, const char* which) {
    if (c.size() == 0) return;
    csr_operands("_register_structure", c[0], c[1], which, edge_index, rows);
    operand("_register_structure", "perm", c[2], edge_index, at::kInt);
    TORCH_CHECK(c[1].size(0) == e && c[2].size(0) == e, "pangnn::_register_structure: ", which, " holds ", c[1].size(0),
                " entries for ", e, " edges");
    if (c.size() == 5) {
      operand("_register_structure", "seg_ptr", c[3], edge_index, at::kLong);
      operand("_register_structure", "parts_rowptr", c[4], edge_index, at::kLong);
      TORCH_CHECK(c[3].dim() == 1 && c[3].size(0) >= 2 && c[4].dim() == 1 && c[4].size(0) == rows + 1,
                  "pangnn::_register_structure: ", which, ": seg_ptr [V + 1], parts_rowptr [rows + 1]");
    }
  };
  auto make = [](at::TensorList c) { return c.size() == 5 ? Csr{c[0], c[1], c[2], c[3], c[4]} : Csr{c[0], c[1], c[2], {}, {}}; };
  check(by_dst, n_dst, "by_dst");
  check(by_src, n_src, "by_src");
  operand("_register_structure", "ei_contig", ei_contig, edge_index, at::kLong);
  TORCH_CHECK(ei_contig.dim() == 2 && ei_contig.size(0) == 2 && ei_contig.size(1) == e,
              "pangnn::_register_structure: ei_contig must be the contiguous [2, E] copy of edge_index");
  std::lock_guard<std::mutex> lock(g_mu);
  Entry& en = find_or_add_locked(edge_index, n_dst, n_src);
  en.ei = ei_contig;
  if (by_dst.size()) en.by_dst = make(by_dst);
  if (by_src.size()) en.by_src = make(by_src);
  if (band >= 0) en.band = (int)band;
  if (sorted_by_src >= 0) en.sorted_by_src = (int)sorted_by_src;
}

void register_plan(const at::Tensor& edge_index, int64_t n_dst, int64_t kind, int64_t chunk_tiles, const at::Tensor& part_off,
                   const at::Tensor& part_rowptr, const at::Tensor& keys, int64_t n_parts) {
  check_edge_index("_register_plan", edge_index);
  TORCH_CHECK(kind >= 0 && kind <= 2 && chunk_tiles >= 1 && n_parts >= 0, "pangnn::_register_plan: bad kind / chunk size");
  operand("_register_plan", "part_off", part_off, edge_index, at::kInt);
  operand("_register_plan", "part_rowptr", part_rowptr, edge_index, at::kLong);
  operand("_register_plan", "keys", keys, edge_index, at::kInt);
  const int64_t e = edge_index.size(1), span = 32 * chunk_tiles;
  TORCH_CHECK(keys.size(0) == e && part_off.size(0) >= (e + span - 1) / span, "pangnn::_register_plan: keys [E] and one part "
              "offset per chunk of ", span, " edges");
  std::lock_guard<std::mutex> lock(g_mu);
  Entry* en = find_locked(edge_index, n_dst);
  TORCH_CHECK(en != nullptr, "pangnn::_register_plan: register the structure first");
  en->plans[{(int)kind, (int)chunk_tiles}] = Plan{part_off, part_rowptr, keys, n_parts};
}

void register_norm(const at::Tensor& edge_index, int64_t n_dst, const c10::optional<at::Tensor>& weight, const at::Tensor& dis,
                   const at::Tensor& by_dst, const at::Tensor& orig, const c10::optional<at::Tensor>& by_src) {
  check_edge_index("_register_norm", edge_index);
  const int64_t e = edge_index.size(1);
  operand("_register_norm", "dis", dis, edge_index, at::kFloat);
  operand("_register_norm", "by_dst", by_dst, edge_index, at::kFloat);
  operand("_register_norm", "orig", orig, edge_index, at::kFloat);
  TORCH_CHECK(dis.size(0) >= n_dst && by_dst.size(0) == e && orig.size(0) == e, "pangnn::_register_norm: dis [N], norms [E]");
  if (by_src.has_value() && by_src->defined()) {
    operand("_register_norm", "by_src", *by_src, edge_index, at::kFloat);
    TORCH_CHECK(by_src->size(0) == e, "pangnn::_register_norm: by_src must be [E]");
  }
  std::lock_guard<std::mutex> lock(g_mu);
  Entry* en = find_locked(edge_index, n_dst);
  TORCH_CHECK(en != nullptr, "pangnn::_register_norm: register the structure first");
  Norm* n = find_norm(*en, weight);
  if (n == nullptr) {
    if (en->norms.size() >= 4) en->norms.erase(en->norms.begin());
    en->norms.emplace_back();
    n = &en->norms.back();
    n->unit = !(weight.has_value() && weight->defined());
    if (!n->unit) n->weight = Ident::of(*weight);
  }
  n->dis = dis;
  n->by_dst = by_dst;
  n->orig = orig;
  if (by_src.has_value() && by_src->defined()) n->by_src = *by_src;
}

void register_actions(const at::Tensor& edge_index, int64_t n_dst, const c10::optional<at::Tensor>& weight, const at::Tensor& x,
                      const at::Tensor& r, const at::Tensor& s) {
  check_edge_index("_register_actions", edge_index);
  operand("_register_actions", "r", r, edge_index, at::kFloat);
  operand("_register_actions", "s", s, edge_index, at::kFloat);
  TORCH_CHECK(r.size(0) == n_dst && s.size(0) == n_dst, "pangnn::_register_actions: r, s must be [N]");
  std::lock_guard<std::mutex> lock(g_mu);
  Entry* en = find_locked(edge_index, n_dst);
  TORCH_CHECK(en != nullptr, "pangnn::_register_actions: register the structure first");
  Norm* n = find_norm(*en, weight);
  TORCH_CHECK(n != nullptr, "pangnn::_register_actions: register the normalisation first");
  n->actions = Actions{Ident::of(x), r, s};
}

void registry_clear(const at::Tensor&) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_entries.clear();
}
int64_t registry_size(const at::Tensor&) {
  std::lock_guard<std::mutex> lock(g_mu);
  return (int64_t)g_entries.size();
}
void registry_forget(const at::Tensor& edge_index, int64_t n_dst) {
  std::lock_guard<std::mutex> lock(g_mu);
  for (size_t i = 0; i < g_entries.size(); ++i)
    if (g_entries[i].n_dst == n_dst && g_entries[i].key.ptr == edge_index.data_ptr()) {
      g_entries.erase(g_entries.begin() + i);
      return;
    }
}

// -- what an op gets out of the registry (copies of the tensor handles: the lock is not held while kernels are enqueued) --
struct View {
  int64_t n_dst = 0, n_src = 0, num_edges = 0;
  at::Tensor ei;
  Csr by_dst, by_src;
  int band = -1, sorted_by_src = -1;
  Plan runsum, plan_dst, plan_src;
  bool has_runsum = false, has_plan_dst = false, has_plan_src = false;
  at::Tensor dis, norm_dst, norm_orig, norm_src, r, s;
};

int chunk_tiles_of(int64_t e) { return pangnn_decoder_chunk_tiles_for(e); }

int64_t missing(const at::Tensor& edge_index, int64_t n_dst, const c10::optional<at::Tensor>& w,
                const c10::optional<at::Tensor>& x, int64_t need, View* v) {
  std::lock_guard<std::mutex> lock(g_mu);
  Entry* en = find_locked(edge_index, n_dst);
  if (en == nullptr || !en->ei.defined()) return need | kEntry;
  int64_t miss = 0;
  const int ct = chunk_tiles_of(en->num_edges);
  v->n_dst = en->n_dst; v->n_src = en->n_src; v->num_edges = en->num_edges; v->ei = en->ei;
  v->by_dst = en->by_dst; v->by_src = en->by_src; v->band = en->band; v->sorted_by_src = en->sorted_by_src;
  if ((need & kByDst) && !en->by_dst.rowptr.defined()) miss |= kByDst;
  if ((need & kBySrc) && !en->by_src.rowptr.defined()) miss |= kBySrc;
  if ((need & kBand) && en->band < 0) miss |= kBand;
  auto plan = [&](int kind, Plan* out, bool* has) {
    auto it = en->plans.find({kind, ct});
    if (it == en->plans.end()) return false;
    *out = it->second; *has = true;
    return true;
  };
  if (need & kRunsum) {
    if (en->sorted_by_src < 0) miss |= kRunsum;
    else if (en->sorted_by_src == 1 && en->num_edges > 0 && !plan(0, &v->runsum, &v->has_runsum)) miss |= kRunsum;
  }
  if ((need & kPlanDst) && en->num_edges > 0 && !plan(1, &v->plan_dst, &v->has_plan_dst)) miss |= kPlanDst;
  if ((need & kPlanSrc) && en->num_edges > 0 && !plan(2, &v->plan_src, &v->has_plan_src)) miss |= kPlanSrc;
  if (need & (kNorm | kNormSrc | kActions)) {
    Norm* n = find_norm(*en, w);
    if (n == nullptr || !n->dis.defined()) miss |= need & (kNorm | kNormSrc | kActions);
    else {
      v->dis = n->dis; v->norm_dst = n->by_dst; v->norm_orig = n->orig; v->norm_src = n->by_src;
      if ((need & kNormSrc) && !n->by_src.defined()) miss |= kNormSrc;
      if (need & kActions) {
        if (x.has_value() && x->defined() && n->actions.x.is(*x)) { v->r = n->actions.r; v->s = n->actions.s; }
        else miss |= kActions;
      }
    }
  }
  return miss;
}

// the registry's view of (edge_index, edge_weight, x) with everything `need` names; builds through the Python hook on a miss
View lookup(const char* op, const at::Tensor& edge_index, int64_t n_dst, const c10::optional<at::Tensor>& w,
            const c10::optional<at::Tensor>& x, int64_t need) {
  check_edge_index(op, edge_index);
  View v;
  int64_t miss = missing(edge_index, n_dst, w, x, need, &v);
  if (miss) {
    static auto prepare = typed_op<void(const at::Tensor&, int64_t, const c10::optional<at::Tensor>&,
                                        const c10::optional<at::Tensor>&, int64_t)>("pangnn::_prepare_structure");
    prepare.call(edge_index, n_dst, w, x, miss);
    v = View();
    miss = missing(edge_index, n_dst, w, x, need, &v);
    TORCH_CHECK(miss == 0, "pangnn::", op, ": the structure registry still lacks component mask ", miss,
                " after pangnn::_prepare_structure (edge_index ", edge_index.sizes(), ", ", n_dst, " target rows)");
  }
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// small shared pieces
// ---------------------------------------------------------------------------------------------------------------
at::Tensor f32c(const at::Tensor& t) { return t.to(at::kFloat).contiguous(); }
c10::optional<at::Tensor> f32c(const c10::optional<at::Tensor>& t) {
  if (t.has_value() && t->defined()) return f32c(*t);
  return c10::nullopt;
}
bool defined(const c10::optional<at::Tensor>& t) { return t.has_value() && t->defined(); }
at::Tensor bytes(size_t n, const at::Tensor& ref) { return at::empty({(int64_t)n}, ref.options().dtype(at::kByte)); }

// out[r] = bias + sum_{k in row r} val[k] x[other[k]] (f32 rows, or bfloat16 rows gathered as stored); fp32 result
at::Tensor spmm_rows(const Csr& csr, const at::Tensor& val, const at::Tensor& x, int64_t n_rows,
                     const c10::optional<at::Tensor>& bias) {
  const int64_t f = x.size(1);
  if (csr.seg_ptr.defined() && (f == 16 || f == 32 || f == 64 || f == 128 || f == 256)) {
    // a hub row: the same kernel over segments (one partial row each), then the contiguous part sum adds a row's partials in
    // order together with the bias (graph.CSR.long_rows)
    const Csr segs{csr.seg_ptr, csr.other, csr.perm, {}, {}};
    const at::Tensor parts = spmm_rows(segs, val, x, csr.seg_ptr.size(0) - 1, c10::nullopt);
    const auto bc = f32c(bias);
    auto out = at::empty({n_rows, f}, parts.options());
    check_rc(pangnn_spmm_csr_f32(csr.parts_rowptr.data_ptr<int64_t>(), nullptr, nullptr, parts.data_ptr<float>(), parts.stride(0),
                                 parts.size(0), opt_ptr<float>(bc), out.data_ptr<float>(), out.stride(0), n_rows, parts.size(0),
                                 (int32_t)f, 0, stream_of(x)),
             "pangnn_spmm_csr_f32(long-row parts)");
    return out;
  }
  const bool rows16 = is_rows16(x) && (f == 32 || f == 64 || f == 128 || f == 256);
  at::Tensor xc;
  if (rows16) {
    const bool ok = x.stride(1) == 1 && x.stride(0) % 4 == 0 && reinterpret_cast<uintptr_t>(x.data_ptr()) % 8 == 0;
    xc = ok ? x : x.contiguous();
  } else {
    xc = f32c(x);
  }
  const auto bc = f32c(bias);
  auto out = at::empty({n_rows, f}, x.options().dtype(at::kFloat));
  const auto fn16 = xc.scalar_type() == at::kHalf ? pangnn_spmm_csr_f16 : pangnn_spmm_csr_bf16;
  const int rc = rows16 ? fn16(csr.rowptr.data_ptr<int64_t>(), csr.other.data_ptr<int32_t>(), val.data_ptr<float>(), xc.data_ptr(),
                               xc.stride(0), xc.size(0), opt_ptr<float>(bc), out.data_ptr<float>(), out.stride(0), n_rows,
                               csr.other.size(0), (int32_t)f, 0, stream_of(x))
                        : pangnn_spmm_csr_f32(csr.rowptr.data_ptr<int64_t>(), csr.other.data_ptr<int32_t>(), val.data_ptr<float>(),
                                              xc.data_ptr<float>(), xc.stride(0), xc.size(0), opt_ptr<float>(bc),
                                              out.data_ptr<float>(), out.stride(0), n_rows, csr.other.size(0), (int32_t)f, 0,
                                              stream_of(x));
  check_rc(rc, "pangnn_spmm_csr");
  return out;
}

// column sums of dL/dout in fp32 (GCNConv's bias gradient): one launch for the short matrices of a mini-batch
at::Tensor colsum(const at::Tensor& g) {
  if (g.dim() == 2 && g.size(0) <= 4096 && g.stride(1) == 1 && g.size(1) > 0 && g.size(1) <= 1024 &&
      (g.scalar_type() == at::kFloat || is_rows16(g))) {
    auto out = at::empty({g.size(1)}, g.options().dtype(at::kFloat));
    check_rc(pangnn_colsum_small(g.data_ptr(), dtype_code(g), g.stride(0), g.size(0), (int32_t)g.size(1), out.data_ptr<float>(),
                                 stream_of(g)),
             "pangnn_colsum_small");
    return out;
  }
  return at::sum(g, {0}, false, at::kFloat);
}

bool band_ok(const at::Tensor& x, const View& v, bool unit_weights) {
  return unit_weights && x.dim() == 2 && (x.size(1) == 64 || x.size(1) == 128) && v.n_src == v.n_dst && v.band > 0;
}

// pangnn_band_propagate: (out, column sums of out | undefined)
std::pair<at::Tensor, at::Tensor> band_call(const at::Tensor& x, const c10::optional<at::Tensor>& bias, const at::Tensor& dis,
                                            int k, bool want_colsum) {
  const at::Tensor xr = rows_any(x);
  const auto bc = f32c(bias);
  const int64_t n = xr.size(0), f = xr.size(1);
  auto out = at::empty({n, f}, xr.options().dtype(at::kFloat));
  at::Tensor cs, ws;
  size_t wsb = 0;
  if (want_colsum) {
    cs = at::empty({f}, out.options());
    wsb = pangnn_band_propagate_workspace_bytes((int32_t)f);
    ws = bytes(wsb, out);
  }
  check_rc(pangnn_band_propagate(xr.data_ptr(), dtype_code(xr), xr.stride(0), dis.data_ptr<float>(), opt_ptr<float>(bc),
                                 out.data_ptr<float>(), out.stride(0), n, (int32_t)f, k, want_colsum ? cs.data_ptr<float>() : nullptr,
                                 want_colsum ? ws.data_ptr() : nullptr, wsb, stream_of(x)),
           "pangnn_band_propagate");
  return {out, cs};
}

// ---------------------------------------------------------------------------------------------------------------
// gcn_propagate: A_hat x + bias  (PyG MessagePassing.propagate + GCNConv.message + bias, gnn.py:158,165)
// ---------------------------------------------------------------------------------------------------------------
at::Tensor gcn_propagate(const at::Tensor& x, const c10::optional<at::Tensor>& bias, const at::Tensor& edge_index,
                         const c10::optional<at::Tensor>& edge_weight, bool allow_band, int64_t out_dtype) {
  on_gpu(x, "x");
  TORCH_CHECK(x.dim() == 2 && x.is_floating_point(), "pangnn::gcn_propagate: x must be a floating-point [N, F]");
  operand_any_float("gcn_propagate", "bias", bias, x);
  operand_any_float("gcn_propagate", "edge_weight", edge_weight, x);
  const DeviceGuard guard(x.device());
  const bool unit = !defined(edge_weight);
  const bool maybe_band = allow_band && unit && (x.size(1) == 64 || x.size(1) == 128);
  const View v = lookup("gcn_propagate", edge_index, x.size(0), edge_weight, c10::nullopt,
                        kByDst | kNorm | (maybe_band ? kBand : 0));
  TORCH_CHECK(v.n_src == x.size(0), "pangnn::gcn_propagate: x has ", x.size(0), " rows, the structure ", v.n_src, " source rows");
  at::Tensor y;
  if (maybe_band && band_ok(x, v, unit)) y = band_call(x, bias, v.dis, v.band, false).first;
  else y = spmm_rows(v.by_dst, v.norm_dst, x, v.n_dst, bias);
  return out_dtype ? y.to(scalar_of(out_dtype)) : y;
}

std::tuple<at::Tensor, at::Tensor> gcn_propagate_backward(const at::Tensor& g, const at::Tensor& edge_index,
                                                         const c10::optional<at::Tensor>& edge_weight, bool allow_band,
                                                         bool has_bias, int64_t x_dtype) {
  on_gpu(g, "g");
  TORCH_CHECK(g.dim() == 2 && g.is_floating_point(), "pangnn::gcn_propagate_backward: g must be a floating-point [N, F]");
  const DeviceGuard guard(g.device());
  const bool unit = !defined(edge_weight);
  const bool maybe_band = allow_band && unit && (g.size(1) == 64 || g.size(1) == 128);
  const View v = lookup("gcn_propagate_backward", edge_index, g.size(0), edge_weight, c10::nullopt,
                        kByDst | kNorm | (maybe_band ? kBand : 0));
  at::Tensor gx, gb;
  if (maybe_band && band_ok(g, v, unit)) {
    auto r = band_call(f32c(g), c10::nullopt, v.dis, v.band, has_bias);      // the band is symmetric: the same kernel
    gx = r.first;
    gb = r.second;
  } else {
    const View vs = lookup("gcn_propagate_backward", edge_index, g.size(0), edge_weight, c10::nullopt, kBySrc | kNorm | kNormSrc);
    const int64_t f = g.size(1);
    const bool keep16 = is_rows16(g) && (f == 32 || f == 64 || f == 128 || f == 256);
    const at::Tensor gg = keep16 ? g : f32c(g);
    gx = spmm_rows(vs.by_src, vs.norm_src, gg, vs.n_src, c10::nullopt);
    if (has_bias) gb = colsum(gg);
  }
  if (x_dtype) gx = gx.to(scalar_of(x_dtype));
  if (!gb.defined()) gb = at::empty({0}, g.options().dtype(at::kFloat));
  return {gx, gb};
}

// ---------------------------------------------------------------------------------------------------------------
// first layer of the scalar-feature model by linearity (gnn.py:97,125 + :131 / :146 / :158)
// ---------------------------------------------------------------------------------------------------------------
struct FirstLayer { at::Tensor wv, bv, win; c10::optional<at::Tensor> bin; int64_t n, h, d; };
FirstLayer first_layer(const char* op, const at::Tensor& x, const at::Tensor& w, const at::Tensor& b, const at::Tensor& w_in,
                       const c10::optional<at::Tensor>& b_in) {
  on_gpu(x, "x");
  TORCH_CHECK(w_in.dim() == 2 && w.numel() == w_in.size(1) && b.numel() == w_in.size(1), "pangnn::", op,
              ": embedding weight / bias must have node_dim = ", w_in.size(1), " entries");
  for (const at::Tensor* t : {&w, &b, &w_in})
    TORCH_CHECK(t->is_cuda() && t->device() == x.device() && t->is_floating_point(), "pangnn::", op,
                ": parameters must be floating-point tensors on ", x.device());
  operand_any_float(op, "b_in", b_in, x);
  FirstLayer p;
  p.wv = f32c(w.detach().reshape({-1}));
  p.bv = f32c(b.detach().reshape({-1}));
  p.win = f32c(w_in.detach());
  p.bin = f32c(b_in);
  p.n = x.size(0);
  p.h = p.win.size(0);
  p.d = p.win.size(1);
  return p;
}

at::Tensor embed_conv_in(const at::Tensor& x, const at::Tensor& w, const at::Tensor& b, const at::Tensor& w_in,
                         const c10::optional<at::Tensor>& b_in, const at::Tensor& edge_index,
                         const c10::optional<at::Tensor>& edge_weight, int64_t out_dtype) {
  const FirstLayer p = first_layer("embed_conv_in", x, w, b, w_in, b_in);
  const DeviceGuard guard(x.device());
  const View v = lookup("embed_conv_in", edge_index, p.n, edge_weight, x, kByDst | kNorm | kActions);
  auto out = at::empty({p.n, p.h}, p.win.options().dtype(scalar_of(out_dtype)));
  check_rc(pangnn_embed_conv_in_rows(v.r.data_ptr<float>(), v.s.data_ptr<float>(), p.wv.data_ptr<float>(), p.bv.data_ptr<float>(),
                                     p.win.data_ptr<float>(), opt_ptr<float>(p.bin), (int32_t)p.d, out.data_ptr(), dtype_code(out),
                                     out.stride(0), p.n, (int32_t)p.h, stream_of(x)),
           "pangnn_embed_conv_in_rows");
  return out;
}

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> embed_conv_in_backward(
    const at::Tensor& g, const at::Tensor& x, const at::Tensor& w, const at::Tensor& b, const at::Tensor& w_in,
    const at::Tensor& edge_index, const c10::optional<at::Tensor>& edge_weight, bool has_bias) {
  const FirstLayer p = first_layer("embed_conv_in_backward", x, w, b, w_in, c10::nullopt);
  const DeviceGuard guard(x.device());
  const View v = lookup("embed_conv_in_backward", edge_index, p.n, edge_weight, x, kByDst | kNorm | kActions);
  const at::Tensor gr = rows_any(g);
  TORCH_CHECK(gr.size(0) == p.n && gr.size(1) == p.h, "pangnn::embed_conv_in_backward: g must be [N, H]");
  auto opt = p.win.options();
  auto g_w = at::empty({p.d, 1}, opt), g_b = at::empty({p.d}, opt), g_win = at::empty({p.h, p.d}, opt);
  auto g_bin = at::empty({has_bias ? p.h : 0}, opt);
  const size_t wsb = pangnn_embed_conv_in_grads_workspace_bytes((int32_t)p.h);
  auto ws = bytes(wsb, p.win);
  check_rc(pangnn_embed_conv_in_grads(gr.data_ptr(), dtype_code(gr), gr.stride(0), v.r.data_ptr<float>(), v.s.data_ptr<float>(), p.n,
                                      p.wv.data_ptr<float>(), p.bv.data_ptr<float>(), p.win.data_ptr<float>(), (int32_t)p.d,
                                      (int32_t)p.h, g_w.data_ptr<float>(), g_b.data_ptr<float>(), g_win.data_ptr<float>(),
                                      has_bias ? g_bin.data_ptr<float>() : nullptr, ws.data_ptr(), wsb, stream_of(x)),
           "pangnn_embed_conv_in_grads");
  return {g_w, g_b, g_win, g_bin};
}

at::Tensor embed_conv_in_linear(const at::Tensor& x, const at::Tensor& w, const at::Tensor& b, const at::Tensor& w_in,
                                const c10::optional<at::Tensor>& b_in, const at::Tensor& w_out,
                                const c10::optional<at::Tensor>& bias_out, const at::Tensor& edge_index,
                                const c10::optional<at::Tensor>& edge_weight) {
  const FirstLayer p = first_layer("embed_conv_in_linear", x, w, b, w_in, b_in);
  operand_any_float("embed_conv_in_linear", "bias_out", bias_out, x);
  TORCH_CHECK(w_out.dim() == 2 && w_out.is_cuda() && w_out.device() == x.device() && w_out.is_floating_point(),
              "pangnn::embed_conv_in_linear: w_out must be a floating-point [M, H] on ", x.device());
  const at::Tensor wout = f32c(w_out.detach());
  const auto bout = f32c(bias_out);
  const int64_t m = wout.size(0);
  TORCH_CHECK(wout.size(1) == p.h && pangnn_embed_linear_supported((int32_t)p.h, (int32_t)m),
              "pangnn::embed_conv_in_linear: unsupported widths H=", p.h, ", M=", m);
  const DeviceGuard guard(x.device());
  const View v = lookup("embed_conv_in_linear", edge_index, p.n, edge_weight, x, kByDst | kNorm | kActions);
  auto y = at::empty({p.n, m}, wout.options());
  check_rc(pangnn_embed_linear_fwd(v.r.data_ptr<float>(), v.s.data_ptr<float>(), p.n, p.wv.data_ptr<float>(), p.bv.data_ptr<float>(),
                                   p.win.data_ptr<float>(), opt_ptr<float>(p.bin), (int32_t)p.d, (int32_t)p.h, wout.data_ptr<float>(),
                                   opt_ptr<float>(bout), (int32_t)m, y.data_ptr<float>(), y.stride(0), stream_of(x)),
           "pangnn_embed_linear_fwd");
  return y;
}

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> embed_conv_in_linear_backward(
    const at::Tensor& g, const at::Tensor& x, const at::Tensor& w, const at::Tensor& b, const at::Tensor& w_in,
    const c10::optional<at::Tensor>& b_in, const at::Tensor& w_out, const at::Tensor& edge_index,
    const c10::optional<at::Tensor>& edge_weight, bool has_bias_out) {
  const FirstLayer p = first_layer("embed_conv_in_linear_backward", x, w, b, w_in, b_in);
  const at::Tensor wout = f32c(w_out.detach());
  const int64_t m = wout.size(0), h = p.h, d = p.d;
  const DeviceGuard guard(x.device());
  const View v = lookup("embed_conv_in_linear_backward", edge_index, p.n, edge_weight, x, kByDst | kNorm | kActions);
  const at::Tensor gg = g.scalar_type() == at::kFloat ? g : g.to(at::kFloat);
  const at::Tensor gr = rows_any(gg);
  TORCH_CHECK(gr.size(0) == p.n && gr.size(1) == m, "pangnn::embed_conv_in_linear_backward: g must be [N, M]");
  const bool has_bin = defined(p.bin);
  auto out = at::empty({m * h + m + 3 * h + d + d + h * d + h}, wout.options());     // one allocation
  int64_t o = 0;
  auto take = [&](int64_t k) { auto t = out.narrow(0, o, k); o += k; return t; };
  at::Tensor g_wout = take(m * h).view({m, h}), g_bout = take(m), sums = take(3 * h), g_w = take(d).view({d, 1}), g_b = take(d),
             g_win = take(h * d).view({h, d}), g_bin = take(h);
  const size_t wsb = pangnn_embed_linear_bwd_workspace_bytes((int32_t)h, (int32_t)m);
  auto ws = bytes(wsb, wout);
  check_rc(pangnn_embed_linear_bwd(gr.data_ptr<float>(), gr.stride(0), v.r.data_ptr<float>(), v.s.data_ptr<float>(), p.n,
                                   p.wv.data_ptr<float>(), p.bv.data_ptr<float>(), p.win.data_ptr<float>(), opt_ptr<float>(p.bin),
                                   (int32_t)d, (int32_t)h, wout.data_ptr<float>(), (int32_t)m, g_wout.data_ptr<float>(),
                                   has_bias_out ? g_bout.data_ptr<float>() : nullptr, sums.data_ptr<float>(), ws.data_ptr(), wsb,
                                   stream_of(x)),
           "pangnn_embed_linear_bwd");
  check_rc(pangnn_embed_conv_in_grads_from_sums(sums.data_ptr<float>(), p.wv.data_ptr<float>(), p.bv.data_ptr<float>(),
                                                p.win.data_ptr<float>(), (int32_t)d, (int32_t)h, g_w.data_ptr<float>(),
                                                g_b.data_ptr<float>(), g_win.data_ptr<float>(),
                                                has_bin ? g_bin.data_ptr<float>() : nullptr, stream_of(x)),
           "pangnn_embed_conv_in_grads_from_sums");
  auto none = [&]() { return at::empty({0}, wout.options()); };
  return {g_w, g_b, g_win, has_bin ? g_bin : none(), g_wout, has_bias_out ? g_bout : none()};
}

// round 2's form of the first layer: agg = A_hat (x w^T + 1 b^T) with the real propagate kernel; backward by linearity
at::Tensor embed_propagate(const at::Tensor& x, const at::Tensor& w, const at::Tensor& b, const at::Tensor& edge_index,
                           const c10::optional<at::Tensor>& edge_weight) {
  on_gpu(x, "x");
  TORCH_CHECK(w.is_cuda() && b.is_cuda() && w.device() == x.device() && b.device() == x.device() && w.numel() == b.numel(),
              "pangnn::embed_propagate: w [D, 1] and b [D] on ", x.device());
  const DeviceGuard guard(x.device());
  const View v = lookup("embed_propagate", edge_index, x.size(0), edge_weight, x, kByDst | kNorm | kActions);
  const at::Tensor xv = x.detach().to(at::kFloat).reshape({-1});
  const at::Tensor h0 = at::addcmul(b.detach().reshape({1, -1}).to(at::kFloat), xv.unsqueeze(1), w.detach().reshape({1, -1}).to(at::kFloat));
  return spmm_rows(v.by_dst, v.norm_dst, h0, v.n_dst, c10::nullopt);
}

std::tuple<at::Tensor, at::Tensor> embed_propagate_backward(const at::Tensor& g, const at::Tensor& x, const at::Tensor& edge_index,
                                                           const c10::optional<at::Tensor>& edge_weight) {
  on_gpu(g, "g");
  const DeviceGuard guard(g.device());
  const View v = lookup("embed_propagate_backward", edge_index, x.size(0), edge_weight, x, kByDst | kNorm | kActions);
  const at::Tensor gg = g.scalar_type() == at::kFloat ? g : g.to(at::kFloat);
  const at::Tensor gr = rows_any(gg);
  const int64_t n = gr.size(0), f = gr.size(1);
  auto out = at::empty({2, f}, gr.options());
  const size_t wsb = pangnn_weighted_colsum_workspace_bytes((int32_t)f);
  auto ws = bytes(wsb, gr);
  check_rc(pangnn_weighted_colsum_f32(gr.data_ptr<float>(), gr.stride(0), v.r.data_ptr<float>(), v.s.data_ptr<float>(), n, (int32_t)f,
                                      out.data_ptr<float>(), ws.data_ptr(), wsb, stream_of(g)),
           "pangnn_weighted_colsum_f32");
  return {out[0].reshape({f, 1}), out[1]};
}

// ---------------------------------------------------------------------------------------------------------------
// decoder: two-wave-per-SIMD training pass (S) + by-target pass (T) of csrc/decoder16.hip, and the inference kernel
// ---------------------------------------------------------------------------------------------------------------
struct DecIn {
  at::Tensor pq;                       // [N, 2D] rows as stored (f32 or bf16)
  c10::optional<at::Tensor> ex, cv;
  at::Tensor w2, b2, w3, b3;
  int64_t d = 0;
  const void* p() const { return pq.data_ptr(); }
  const void* q() const { return static_cast<const char*>(pq.data_ptr()) + d * pq.element_size(); }
};

DecIn decoder_inputs(const char* op, const at::Tensor& pq, const c10::optional<at::Tensor>& extra,
                     const c10::optional<at::Tensor>& cvec, const at::Tensor& w2, const at::Tensor& b2, const at::Tensor& w3,
                     const at::Tensor& b3) {
  on_gpu(pq, "pq");
  TORCH_CHECK(pq.dim() == 2 && pq.is_floating_point() && pq.size(1) == 128, "pangnn::", op,
              ": pq must be [N, 128] (node_dim 64: P | Q), got ", pq.sizes());
  operand_any_float(op, "extra", extra, pq);
  operand_any_float(op, "cvec", cvec, pq);
  for (const at::Tensor* t : {&w2, &b2, &w3, &b3})
    TORCH_CHECK(t->is_cuda() && t->device() == pq.device() && t->is_floating_point(), "pangnn::", op,
                ": decoder parameters must be floating-point tensors on ", pq.device());
  TORCH_CHECK(w2.numel() == 64 * 64 && b2.numel() == 64 && w3.numel() == 64 && b3.numel() == 1, "pangnn::", op,
              ": mlp[2] is [64, 64] + [64], mlp[4] is [1, 64] + [1]");
  TORCH_CHECK(defined(extra) == defined(cvec), "pangnn::", op, ": the skip feature and its weight column go together");
  DecIn in;
  in.pq = rows_any(pq);
  in.d = in.pq.size(1) / 2;
  in.ex = f32c(extra);
  in.cv = f32c(cvec);
  in.w2 = f32c(w2); in.b2 = f32c(b2); in.w3 = f32c(w3.reshape({-1})); in.b3 = f32c(b3.reshape({-1}));
  return in;
}

// out[s] = sum of the consecutive part rows of row s (pangnn_spmm_csr_f32 with idx = NULL)
void sum_parts(const Plan& plan, const at::Tensor& parts, int64_t n_rows, at::Tensor& out) {
  check_rc(pangnn_spmm_csr_f32(plan.part_rowptr.data_ptr<int64_t>(), nullptr, nullptr, parts.data_ptr<float>(), parts.stride(0),
                               parts.size(0), nullptr, out.data_ptr<float>(), out.stride(0), n_rows, parts.size(0),
                               (int32_t)parts.size(1), 0, stream_of(parts)),
           "pangnn_spmm_csr_f32(parts)");
}

// dL/dh1 summed over the rows of one CSR order from the per-edge records (T kernel + part sum); `g_b2`: also dL/db2
void dgrad_sum(const at::Tensor& rec, const View& v, int by /* 1 dst, 2 src, 0 none */, const DecIn& in, int64_t n_rows,
               at::Tensor* out, at::Tensor* g_b2, const c10::optional<at::Tensor>& live) {
  const Plan* plan = by == 1 ? &v.plan_dst : by == 2 ? &v.plan_src : nullptr;
  const Csr* csr = by == 1 ? &v.by_dst : by == 2 ? &v.by_src : nullptr;
  at::Tensor parts, ws;
  if (plan) parts = at::empty({plan->n_parts, 64}, in.w2.options());
  size_t wsb = 0;
  if (g_b2) {
    wsb = pangnn_decoder_dgrad_workspace_bytes();
    ws = bytes(wsb, in.w2);
  }
  check_rc(pangnn_decoder_dgrad_f32(reinterpret_cast<const uint32_t*>(rec.data_ptr<int32_t>()),
                                    csr ? csr->perm.data_ptr<int32_t>() : nullptr, plan ? plan->keys.data_ptr<int32_t>() : nullptr,
                                    in.w2.data_ptr<float>(), in.w3.data_ptr<float>(), v.num_edges,
                                    plan ? parts.data_ptr<float>() : nullptr, plan ? plan->part_off.data_ptr<int32_t>() : nullptr,
                                    g_b2 ? g_b2->data_ptr<float>() : nullptr, opt_ptr<int64_t>(live), g_b2 ? ws.data_ptr() : nullptr,
                                    wsb, stream_of(rec)),
           "pangnn_decoder_dgrad_f32");
  if (plan) sum_parts(*plan, parts, n_rows, *out);
}

struct DecGrads { at::Tensor loss, logits, g_pq, g_cv, g_w2, g_b2, g_w3, g_b3; };

// the one-pass training decoder on the P | Q table: S (logits, loss or the given dL/dlogits, parameter gradients, by-source
// run sums, records) then T (by-target sums, dL/db2); no [E, 64] tensor exists
DecGrads decoder_train(const char* op, const DecIn& in, const at::Tensor& edge_index, const c10::optional<at::Tensor>& y,
                       const c10::optional<at::Tensor>& pos_weight, int64_t denom, const c10::optional<at::Tensor>& g_logits,
                       const c10::optional<at::Tensor>& live) {
  const int64_t n = in.pq.size(0), d = in.d;
  View v = lookup(op, edge_index, n, c10::nullopt, c10::nullopt, kByDst | kRunsum | kPlanDst);
  const int64_t e = v.num_edges;
  if (e > 0 && v.sorted_by_src != 1) v = lookup(op, edge_index, n, c10::nullopt, c10::nullopt, kByDst | kBySrc | kRunsum | kPlanDst | kPlanSrc);
  const bool fused = defined(y);
  auto fo = in.w2.options();
  DecGrads r;
  if (fused) {
    r.logits = at::empty({e}, fo);
    r.loss = at::empty({1}, fo);
  }
  r.g_w2 = at::empty({64, 64}, fo);
  r.g_b2 = at::empty({64}, fo);
  r.g_w3 = at::empty({64}, fo);
  r.g_b3 = at::empty({1}, fo);
  r.g_cv = defined(in.cv) ? at::empty({64}, fo) : at::empty({0}, fo);
  auto rec = at::empty({std::max<int64_t>(e, 1), 8}, fo.dtype(at::kInt));
  const bool runs = v.has_runsum && e > 0;
  at::Tensor parts;
  if (runs) parts = at::empty({v.runsum.n_parts, d}, fo);
  const size_t wsb = pangnn_decoder_train_workspace_bytes();
  auto ws = bytes(wsb, in.w2);
  const auto yy = f32c(y), gl = f32c(g_logits);
  c10::optional<at::Tensor> pw;
  if (defined(pos_weight)) pw = f32c(*pos_weight).reshape({-1});
  check_rc(pangnn_decoder_train_mixed(in.p(), in.pq.stride(0), in.q(), in.pq.stride(0), dtype_code(in.pq), n,
                                      v.ei.data_ptr<int64_t>(), e, e, opt_ptr<float>(in.ex), opt_ptr<float>(in.cv),
                                      in.w2.data_ptr<float>(), in.b2.data_ptr<float>(), in.w3.data_ptr<float>(), in.b3.data_ptr<float>(),
                                      (int32_t)d, opt_ptr<float>(yy), opt_ptr<float>(pw), denom, opt_ptr<float>(gl),
                                      fused ? r.logits.data_ptr<float>() : nullptr, fused ? r.loss.data_ptr<float>() : nullptr,
                                      reinterpret_cast<uint32_t*>(rec.data_ptr<int32_t>()), runs ? parts.data_ptr<float>() : nullptr,
                                      runs ? v.runsum.part_off.data_ptr<int32_t>() : nullptr, r.g_w2.data_ptr<float>(),
                                      r.g_w3.data_ptr<float>(), r.g_b3.data_ptr<float>(),
                                      defined(in.cv) ? r.g_cv.data_ptr<float>() : nullptr, opt_ptr<int64_t>(live), ws.data_ptr(), wsb,
                                      stream_of(in.pq)),
           "pangnn_decoder_train_mixed");
  r.g_pq = at::empty({n, 2 * d}, fo);
  at::Tensor gp = r.g_pq.narrow(1, 0, d), gq = r.g_pq.narrow(1, d, d);
  if (e == 0) {
    r.g_pq.zero_();
    r.g_b2.zero_();
    return r;
  }
  at::Tensor* b2_pending = &r.g_b2;               // dL/db2 comes out of exactly one T call
  if (runs) sum_parts(v.runsum, parts, n, gp);
  else { dgrad_sum(rec, v, 2, in, n, &gp, b2_pending, live); b2_pending = nullptr; }
  dgrad_sum(rec, v, 1, in, n, &gq, b2_pending, live);
  return r;
}

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> decoder_loss(
    const at::Tensor& pq, const at::Tensor& edge_index, const c10::optional<at::Tensor>& extra, const c10::optional<at::Tensor>& cvec,
    const at::Tensor& w2, const at::Tensor& b2, const at::Tensor& w3, const at::Tensor& b3, const at::Tensor& y,
    const c10::optional<at::Tensor>& pos_weight, int64_t denom, const c10::optional<at::Tensor>& live) {
  const DecIn in = decoder_inputs("decoder_loss", pq, extra, cvec, w2, b2, w3, b3);
  operand_any_float("decoder_loss", "y", y, pq);
  operand_any_float("decoder_loss", "pos_weight", pos_weight, pq);
  TORCH_CHECK(!defined(live) || (live->is_cuda() && live->device() == pq.device() && live->scalar_type() == at::kLong),
              "pangnn::decoder_loss: live must be a device int64 tensor");
  TORCH_CHECK(y.dim() == 1 && y.size(0) == edge_index.size(1) && denom > 0, "pangnn::decoder_loss: y must be [E], denom > 0");
  const DeviceGuard guard(pq.device());
  DecGrads r = decoder_train("decoder_loss", in, edge_index, y, pos_weight, denom, c10::nullopt, live);
  return {r.loss.view(at::IntArrayRef{}), r.logits, r.g_pq, r.g_cv, r.g_w2.view_as(w2), r.g_b2, r.g_w3.view_as(w3), r.g_b3.view_as(b3)};
}

at::Tensor decoder_mlp(const at::Tensor& pq, const at::Tensor& edge_index, const c10::optional<at::Tensor>& extra,
                       const c10::optional<at::Tensor>& cvec, const at::Tensor& w2, const at::Tensor& b2, const at::Tensor& w3,
                       const at::Tensor& b3) {
  const DecIn in = decoder_inputs("decoder_mlp", pq, extra, cvec, w2, b2, w3, b3);
  const DeviceGuard guard(pq.device());
  const View v = lookup("decoder_mlp", edge_index, in.pq.size(0), c10::nullopt, c10::nullopt, kEntry);
  const int64_t e = v.num_edges;
  auto logits = at::empty({e}, in.w2.options());
  check_rc(pangnn_decoder_mlp_infer_mixed(in.p(), in.pq.stride(0), in.q(), in.pq.stride(0), dtype_code(in.pq), in.pq.size(0),
                                          v.ei.data_ptr<int64_t>(), e, e, opt_ptr<float>(in.ex), opt_ptr<float>(in.cv),
                                          in.w2.data_ptr<float>(), in.b2.data_ptr<float>(), in.w3.data_ptr<float>(),
                                          in.b3.data_ptr<float>(), (int32_t)in.d, logits.data_ptr<float>(), stream_of(pq)),
           "pangnn_decoder_mlp_infer_mixed");
  return logits;
}

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> decoder_mlp_backward(
    const at::Tensor& g, const at::Tensor& pq, const at::Tensor& edge_index, const c10::optional<at::Tensor>& extra,
    const c10::optional<at::Tensor>& cvec, const at::Tensor& w2, const at::Tensor& b2, const at::Tensor& w3, const at::Tensor& b3) {
  const DecIn in = decoder_inputs("decoder_mlp_backward", pq, extra, cvec, w2, b2, w3, b3);
  TORCH_CHECK(g.is_cuda() && g.device() == pq.device() && g.dim() == 1 && g.size(0) == edge_index.size(1),
              "pangnn::decoder_mlp_backward: g must be [E] on ", pq.device());
  const DeviceGuard guard(pq.device());
  DecGrads r = decoder_train("decoder_mlp_backward", in, edge_index, c10::nullopt, c10::nullopt, 0, g, c10::nullopt);
  return {r.g_pq, r.g_cv, r.g_w2.view_as(w2), r.g_b2, r.g_w3.view_as(w3), r.g_b3.view_as(b3)};
}

// ---------------------------------------------------------------------------------------------------------------
// autograd formulas (Autograd key): forward redispatches below autograd, backward calls the registered backward op — a
// tracer sees both
// ---------------------------------------------------------------------------------------------------------------
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;
using OptT = c10::optional<at::Tensor>;

at::Tensor or_undef(const OptT& t) { return defined(t) ? *t : at::Tensor(); }
OptT opt_of(const at::Tensor& t) { return t.defined() ? OptT(t) : OptT(); }

class GcnPropagateFunction : public torch::autograd::Function<GcnPropagateFunction> {
 public:
  static at::Tensor forward(AutogradContext* ctx, const at::Tensor& x, const OptT& bias, const at::Tensor& edge_index,
                            const OptT& edge_weight, bool allow_band, int64_t out_dtype) {
    at::AutoDispatchBelowADInplaceOrView below;
    ctx->save_for_backward({edge_index, or_undef(edge_weight)});
    ctx->saved_data["allow_band"] = allow_band;
    ctx->saved_data["has_bias"] = defined(bias);
    ctx->saved_data["x_dtype"] = (int64_t)dtype_code(x);
    static auto op = typed_op<at::Tensor(const at::Tensor&, const OptT&, const at::Tensor&, const OptT&, bool, int64_t)>(
        "pangnn::gcn_propagate");
    return op.call(x, bias, edge_index, edge_weight, allow_band, out_dtype);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const bool has_bias = ctx->saved_data["has_bias"].toBool();
    static auto op = typed_op<std::tuple<at::Tensor, at::Tensor>(const at::Tensor&, const at::Tensor&, const OptT&, bool, bool, int64_t)>(
        "pangnn::gcn_propagate_backward");
    auto [gx, gb] = op.call(grads[0], saved[0], opt_of(saved[1]), ctx->saved_data["allow_band"].toBool(), has_bias,
                            ctx->saved_data["x_dtype"].toInt());
    return {ctx->needs_input_grad(0) ? gx : at::Tensor(), has_bias ? gb : at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(),
            at::Tensor()};
  }
};
at::Tensor gcn_propagate_autograd(const at::Tensor& x, const OptT& bias, const at::Tensor& edge_index, const OptT& edge_weight,
                                  bool allow_band, int64_t out_dtype) {
  return GcnPropagateFunction::apply(x, bias, edge_index, edge_weight, allow_band, out_dtype);
}

class EmbedConvInFunction : public torch::autograd::Function<EmbedConvInFunction> {
 public:
  static at::Tensor forward(AutogradContext* ctx, const at::Tensor& x, const at::Tensor& w, const at::Tensor& b,
                            const at::Tensor& w_in, const OptT& b_in, const at::Tensor& edge_index, const OptT& edge_weight,
                            int64_t out_dtype) {
    at::AutoDispatchBelowADInplaceOrView below;
    ctx->save_for_backward({x, w, b, w_in, edge_index, or_undef(edge_weight)});
    ctx->saved_data["has_bias"] = defined(b_in);
    static auto op = typed_op<at::Tensor(const at::Tensor&, const at::Tensor&, const at::Tensor&, const at::Tensor&, const OptT&,
                                         const at::Tensor&, const OptT&, int64_t)>("pangnn::embed_conv_in");
    return op.call(x, w, b, w_in, b_in, edge_index, edge_weight, out_dtype);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto s = ctx->get_saved_variables();
    const bool has_bias = ctx->saved_data["has_bias"].toBool();
    static auto op = typed_op<std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor>(
        const at::Tensor&, const at::Tensor&, const at::Tensor&, const at::Tensor&, const at::Tensor&, const at::Tensor&, const OptT&,
        bool)>("pangnn::embed_conv_in_backward");
    auto [g_w, g_b, g_win, g_bin] = op.call(grads[0], s[0], s[1], s[2], s[3], s[4], opt_of(s[5]), has_bias);
    return {at::Tensor(), g_w.reshape(s[1].sizes()), g_b, g_win, has_bias ? g_bin : at::Tensor(), at::Tensor(), at::Tensor(),
            at::Tensor()};
  }
};
at::Tensor embed_conv_in_autograd(const at::Tensor& x, const at::Tensor& w, const at::Tensor& b, const at::Tensor& w_in,
                                  const OptT& b_in, const at::Tensor& edge_index, const OptT& edge_weight, int64_t out_dtype) {
  return EmbedConvInFunction::apply(x, w, b, w_in, b_in, edge_index, edge_weight, out_dtype);
}

class EmbedConvInLinearFunction : public torch::autograd::Function<EmbedConvInLinearFunction> {
 public:
  static at::Tensor forward(AutogradContext* ctx, const at::Tensor& x, const at::Tensor& w, const at::Tensor& b,
                            const at::Tensor& w_in, const OptT& b_in, const at::Tensor& w_out, const OptT& bias_out,
                            const at::Tensor& edge_index, const OptT& edge_weight) {
    at::AutoDispatchBelowADInplaceOrView below;
    ctx->save_for_backward({x, w, b, w_in, or_undef(b_in), w_out, edge_index, or_undef(edge_weight)});
    ctx->saved_data["has_bout"] = defined(bias_out);
    static auto op = typed_op<at::Tensor(const at::Tensor&, const at::Tensor&, const at::Tensor&, const at::Tensor&, const OptT&,
                                         const at::Tensor&, const OptT&, const at::Tensor&, const OptT&)>(
        "pangnn::embed_conv_in_linear");
    return op.call(x, w, b, w_in, b_in, w_out, bias_out, edge_index, edge_weight);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto s = ctx->get_saved_variables();
    const bool has_bout = ctx->saved_data["has_bout"].toBool(), has_bin = s[4].defined();
    static auto op = typed_op<std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor>(
        const at::Tensor&, const at::Tensor&, const at::Tensor&, const at::Tensor&, const at::Tensor&, const OptT&, const at::Tensor&,
        const at::Tensor&, const OptT&, bool)>("pangnn::embed_conv_in_linear_backward");
    auto [g_w, g_b, g_win, g_bin, g_wout, g_bout] = op.call(grads[0], s[0], s[1], s[2], s[3], opt_of(s[4]), s[5], s[6], opt_of(s[7]),
                                                            has_bout);
    return {at::Tensor(), g_w.reshape(s[1].sizes()), g_b, g_win, has_bin ? g_bin : at::Tensor(), g_wout,
            has_bout ? g_bout : at::Tensor(), at::Tensor(), at::Tensor()};
  }
};
at::Tensor embed_conv_in_linear_autograd(const at::Tensor& x, const at::Tensor& w, const at::Tensor& b, const at::Tensor& w_in,
                                         const OptT& b_in, const at::Tensor& w_out, const OptT& bias_out,
                                         const at::Tensor& edge_index, const OptT& edge_weight) {
  return EmbedConvInLinearFunction::apply(x, w, b, w_in, b_in, w_out, bias_out, edge_index, edge_weight);
}

class EmbedPropagateFunction : public torch::autograd::Function<EmbedPropagateFunction> {
 public:
  static at::Tensor forward(AutogradContext* ctx, const at::Tensor& x, const at::Tensor& w, const at::Tensor& b,
                            const at::Tensor& edge_index, const OptT& edge_weight) {
    at::AutoDispatchBelowADInplaceOrView below;
    ctx->save_for_backward({x, edge_index, or_undef(edge_weight)});
    static auto op = typed_op<at::Tensor(const at::Tensor&, const at::Tensor&, const at::Tensor&, const at::Tensor&, const OptT&)>(
        "pangnn::embed_propagate");
    return op.call(x, w, b, edge_index, edge_weight);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto s = ctx->get_saved_variables();
    static auto op = typed_op<std::tuple<at::Tensor, at::Tensor>(const at::Tensor&, const at::Tensor&, const at::Tensor&, const OptT&)>(
        "pangnn::embed_propagate_backward");
    auto [g_w, g_b] = op.call(grads[0], s[0], s[1], opt_of(s[2]));
    return {at::Tensor(), g_w, g_b, at::Tensor(), at::Tensor()};
  }
};
at::Tensor embed_propagate_autograd(const at::Tensor& x, const at::Tensor& w, const at::Tensor& b, const at::Tensor& edge_index,
                                    const OptT& edge_weight) {
  return EmbedPropagateFunction::apply(x, w, b, edge_index, edge_weight);
}

// The upstream gradient of the fused loss: `loss.backward(unit_grad)` of this package's own train step hands over a
// registered tensor (recognised by address: nothing to do); `loss.backward()` / accelerate's `(loss / 1).backward()` hand over
// a device scalar that IS 1 — pangnn_scale_unless_one_f32 finds that out on the device and leaves; anything else is applied in
// place (the stored gradients are consumed: a second backward through the node raises).  While a tracer runs the formula
// (fake tensors carry no memory) the products are taken out of place with ATen.
std::mutex g_unit_mu;
std::vector<const void*> g_unit_grads;
void set_unit_grad(const at::Tensor& t) {
  std::lock_guard<std::mutex> lock(g_unit_mu);
  g_unit_grads.push_back(t.data_ptr());
}
bool traced(const at::Tensor& t) {
  return !t.is_cuda() || t.key_set().has(c10::DispatchKey::Python) || t.key_set().has(c10::DispatchKey::Meta) ||
         t.key_set().has(c10::DispatchKey::Functionalize);
}
bool is_unit_grad(const at::Tensor& go) {
  if (go.dim() != 0 || traced(go)) return false;
  std::lock_guard<std::mutex> lock(g_unit_mu);
  for (const void* p : g_unit_grads)
    if (p == go.data_ptr()) return true;
  return false;
}
std::vector<at::Tensor> scale_by_loss_grad(AutogradContext* ctx, std::vector<at::Tensor> g, const at::Tensor& go) {
  if (is_unit_grad(go)) return g;
  bool plain = !traced(go);
  for (auto& t : g)
    if (t.defined() && t.numel() > 0 && (traced(t) || t.scalar_type() != at::kFloat || !t.is_contiguous())) plain = false;
  if (!plain) {
    for (auto& t : g)
      if (t.defined()) t = t * go;
    return g;
  }
  TORCH_CHECK(!ctx->saved_data.count("scaled"), "pangnn_amd: the fused decoder loss computes its gradients in its forward pass "
              "and hands them over once; backward through it a second time is not supported (call the model again)");
  ctx->saved_data["scaled"] = true;
  float* ptrs[PANGNN_SCALE_MAX_TENSORS];
  int64_t counts[PANGNN_SCALE_MAX_TENSORS];
  int n = 0;
  for (auto& t : g)
    if (t.defined() && t.numel() > 0) {
      TORCH_CHECK(n < PANGNN_SCALE_MAX_TENSORS, "pangnn::decoder_loss backward: too many gradient tensors");
      ptrs[n] = t.data_ptr<float>();
      counts[n++] = t.numel();
    }
  if (n) {
    const DeviceGuard guard(go.device());
    const at::Tensor g32 = go.scalar_type() == at::kFloat ? go : go.to(at::kFloat);
    check_rc(pangnn_scale_unless_one_f32(ptrs, counts, n, g32.data_ptr<float>(), stream_of(go)), "pangnn_scale_unless_one_f32");
  }
  return g;
}

class DecoderLossFunction : public torch::autograd::Function<DecoderLossFunction> {
 public:
  static variable_list forward(AutogradContext* ctx, const at::Tensor& pq, const at::Tensor& edge_index, const OptT& extra,
                               const OptT& cvec, const at::Tensor& w2, const at::Tensor& b2, const at::Tensor& w3,
                               const at::Tensor& b3, const at::Tensor& y, const OptT& pos_weight, int64_t denom, const OptT& live) {
    at::AutoDispatchBelowADInplaceOrView below;
    static auto op = typed_op<std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor>(
        const at::Tensor&, const at::Tensor&, const OptT&, const OptT&, const at::Tensor&, const at::Tensor&, const at::Tensor&,
        const at::Tensor&, const at::Tensor&, const OptT&, int64_t, const OptT&)>("pangnn::decoder_loss");
    auto [loss, logits, g_pq, g_cv, g_w2, g_b2, g_w3, g_b3] = op.call(pq, edge_index, extra, cvec, w2, b2, w3, b3, y, pos_weight, denom, live);
    ctx->save_for_backward({g_pq, g_cv, g_w2, g_b2, g_w3, g_b3});
    ctx->saved_data["has_cv"] = defined(cvec);
    ctx->mark_non_differentiable({logits, g_pq, g_cv, g_w2, g_b2, g_w3, g_b3});
    return {loss, logits, g_pq, g_cv, g_w2, g_b2, g_w3, g_b3};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    variable_list out(12);
    if (!grads[0].defined()) return out;
    const auto s = ctx->get_saved_variables();
    const bool has_cv = ctx->saved_data["has_cv"].toBool();
    auto g = scale_by_loss_grad(ctx, {s[0], has_cv ? s[1] : at::Tensor(), s[2], s[3], s[4], s[5]}, grads[0]);
    out[0] = g[0];
    if (has_cv) out[3] = g[1];
    out[4] = g[2]; out[5] = g[3]; out[6] = g[4]; out[7] = g[5];
    return out;
  }
};
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> decoder_loss_autograd(
    const at::Tensor& pq, const at::Tensor& edge_index, const OptT& extra, const OptT& cvec, const at::Tensor& w2, const at::Tensor& b2,
    const at::Tensor& w3, const at::Tensor& b3, const at::Tensor& y, const OptT& pos_weight, int64_t denom, const OptT& live) {
  auto o = DecoderLossFunction::apply(pq, edge_index, extra, cvec, w2, b2, w3, b3, y, pos_weight, denom, live);
  return {o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]};
}

class DecoderMlpFunction : public torch::autograd::Function<DecoderMlpFunction> {
 public:
  static at::Tensor forward(AutogradContext* ctx, const at::Tensor& pq, const at::Tensor& edge_index, const OptT& extra,
                            const OptT& cvec, const at::Tensor& w2, const at::Tensor& b2, const at::Tensor& w3, const at::Tensor& b3) {
    at::AutoDispatchBelowADInplaceOrView below;
    ctx->save_for_backward({pq, w2, b2, w3, b3, edge_index, or_undef(extra), or_undef(cvec)});
    static auto op = typed_op<at::Tensor(const at::Tensor&, const at::Tensor&, const OptT&, const OptT&, const at::Tensor&,
                                         const at::Tensor&, const at::Tensor&, const at::Tensor&)>("pangnn::decoder_mlp");
    return op.call(pq, edge_index, extra, cvec, w2, b2, w3, b3);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto s = ctx->get_saved_variables();
    static auto op = typed_op<std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor>(
        const at::Tensor&, const at::Tensor&, const at::Tensor&, const OptT&, const OptT&, const at::Tensor&, const at::Tensor&,
        const at::Tensor&, const at::Tensor&)>("pangnn::decoder_mlp_backward");
    auto [g_pq, g_cv, g_w2, g_b2, g_w3, g_b3] = op.call(grads[0], s[0], s[5], opt_of(s[6]), opt_of(s[7]), s[1], s[2], s[3], s[4]);
    return {g_pq, at::Tensor(), at::Tensor(), s[7].defined() ? g_cv : at::Tensor(), g_w2, g_b2, g_w3, g_b3};
  }
};
at::Tensor decoder_mlp_autograd(const at::Tensor& pq, const at::Tensor& edge_index, const OptT& extra, const OptT& cvec,
                                const at::Tensor& w2, const at::Tensor& b2, const at::Tensor& w3, const at::Tensor& b3) {
  return DecoderMlpFunction::apply(pq, edge_index, extra, cvec, w2, b2, w3, b3);
}

}  // namespace

TORCH_LIBRARY_FRAGMENT(pangnn, m) {
  // structure registry (pangnn_amd/graph.py pushes what it built; _prepare_structure is implemented there)
  m.def("_register_structure(Tensor edge_index, int n_dst, int n_src, Tensor ei_contig, Tensor[] by_dst, Tensor[] by_src, int band, "
        "int sorted_by_src) -> ()");
  m.def("_register_plan(Tensor edge_index, int n_dst, int kind, int chunk_tiles, Tensor part_off, Tensor part_rowptr, Tensor keys, "
        "int n_parts) -> ()");
  m.def("_register_norm(Tensor edge_index, int n_dst, Tensor? weight, Tensor dis, Tensor by_dst, Tensor orig, Tensor? by_src) -> ()");
  m.def("_register_actions(Tensor edge_index, int n_dst, Tensor? weight, Tensor x, Tensor r, Tensor s) -> ()");
  m.def("_prepare_structure(Tensor edge_index, int n_dst, Tensor? edge_weight, Tensor? x, int need) -> ()");
  m.def("_registry_clear(Tensor any) -> ()");
  m.def("_registry_size(Tensor any) -> int");
  m.def("_registry_forget(Tensor edge_index, int n_dst) -> ()");
  m.def("_set_unit_grad(Tensor t) -> ()");
  // the per-step operators that read a graph
  m.def("gcn_propagate(Tensor x, Tensor? bias, Tensor edge_index, Tensor? edge_weight, bool allow_band, int out_dtype) -> Tensor");
  m.def("gcn_propagate_backward(Tensor g, Tensor edge_index, Tensor? edge_weight, bool allow_band, bool has_bias, int x_dtype) -> "
        "(Tensor, Tensor)");
  m.def("embed_conv_in(Tensor x, Tensor w, Tensor b, Tensor w_in, Tensor? b_in, Tensor edge_index, Tensor? edge_weight, "
        "int out_dtype) -> Tensor");
  m.def("embed_conv_in_backward(Tensor g, Tensor x, Tensor w, Tensor b, Tensor w_in, Tensor edge_index, Tensor? edge_weight, "
        "bool has_bias) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("embed_conv_in_linear(Tensor x, Tensor w, Tensor b, Tensor w_in, Tensor? b_in, Tensor w_out, Tensor? bias_out, "
        "Tensor edge_index, Tensor? edge_weight) -> Tensor");
  m.def("embed_conv_in_linear_backward(Tensor g, Tensor x, Tensor w, Tensor b, Tensor w_in, Tensor? b_in, Tensor w_out, "
        "Tensor edge_index, Tensor? edge_weight, bool has_bias_out) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("embed_propagate(Tensor x, Tensor w, Tensor b, Tensor edge_index, Tensor? edge_weight) -> Tensor");
  m.def("embed_propagate_backward(Tensor g, Tensor x, Tensor edge_index, Tensor? edge_weight) -> (Tensor, Tensor)");
  m.def("decoder_loss(Tensor pq, Tensor edge_index, Tensor? extra, Tensor? cvec, Tensor w2, Tensor b2, Tensor w3, Tensor b3, "
        "Tensor y, Tensor? pos_weight, int denom, Tensor? live) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
  m.def("decoder_mlp(Tensor pq, Tensor edge_index, Tensor? extra, Tensor? cvec, Tensor w2, Tensor b2, Tensor w3, Tensor b3) -> Tensor");
  m.def("decoder_mlp_backward(Tensor g, Tensor pq, Tensor edge_index, Tensor? extra, Tensor? cvec, Tensor w2, Tensor b2, Tensor w3, "
        "Tensor b3) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(pangnn, CompositeExplicitAutograd, m) {
  m.impl("_register_structure", &register_structure);
  m.impl("_register_plan", &register_plan);
  m.impl("_register_norm", &register_norm);
  m.impl("_register_actions", &register_actions);
  m.impl("_registry_clear", &registry_clear);
  m.impl("_registry_size", &registry_size);
  m.impl("_registry_forget", &registry_forget);
  m.impl("_set_unit_grad", &set_unit_grad);
}

TORCH_LIBRARY_IMPL(pangnn, CUDA, m) {
  m.impl("gcn_propagate", &gcn_propagate);
  m.impl("gcn_propagate_backward", &gcn_propagate_backward);
  m.impl("embed_conv_in", &embed_conv_in);
  m.impl("embed_conv_in_backward", &embed_conv_in_backward);
  m.impl("embed_conv_in_linear", &embed_conv_in_linear);
  m.impl("embed_conv_in_linear_backward", &embed_conv_in_linear_backward);
  m.impl("embed_propagate", &embed_propagate);
  m.impl("embed_propagate_backward", &embed_propagate_backward);
  m.impl("decoder_loss", &decoder_loss);
  m.impl("decoder_mlp", &decoder_mlp);
  m.impl("decoder_mlp_backward", &decoder_mlp_backward);
}

TORCH_LIBRARY_IMPL(pangnn, Autograd, m) {
  m.impl("gcn_propagate", &gcn_propagate_autograd);
  m.impl("embed_conv_in", &embed_conv_in_autograd);
  m.impl("embed_conv_in_linear", &embed_conv_in_linear_autograd);
  m.impl("embed_propagate", &embed_propagate_autograd);
  m.impl("decoder_loss", &decoder_loss_autograd);
  m.impl("decoder_mlp", &decoder_mlp_autograd);
}
