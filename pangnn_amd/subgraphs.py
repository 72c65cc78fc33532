"""Per-ortholog-group sub-graphs — the reference's actual training regime — built for all groups at
once as sort / unique / searchsorted tensor programs (runs on the device the relation lives on).

Restates /root/reference/src/dataset.py:222-322 with src/helper.py:327-433 and
src/preprocessing.py:73-156,264-325:

  for every ortholog group (>= 2 genes):
    1. BFS over the normalised similarity relation, `neighbours` hops from the group's genes
       (get_connected_nodes)                                                       -> "core" nodes
    2. positional neighbours v +- k (k <= neighbours, inside [0, N), ignoring genome ends) of every
       core node are added as extra nodes; neighbour edges (core v, v +- k) in both directions,
       de-duplicated (get_neighbour_graph + remove_duplicate_edges_tuple)
    3. similarity edges = every normalised edge whose two endpoints are in the sub-graph
       (build_edge_index over the sub dict); weights and labels as for the whole graph
    4. the group is dropped if it has no similarity rows, or (real data on a genome subset) fewer
       similarity edges than group members (dataset.py:248,252)

The reference numbers a sub-graph's nodes in CPython string-set iteration order (helper.py:344), which
no other implementation can reproduce; here local ids are: core nodes by ascending gene position, then
the extra neighbour nodes by ascending position.  Parity with the reference-built fixtures is therefore
checked on GLOBAL ids: same node set, same similarity edge set with the same weights and labels, same
neighbour edge set (tests/test_construct.py).

The result is one flat "dataset" whose index tensors already carry dataset-wide node numbering, so a
mini-batch of consecutive sub-graphs (PyG `Batch.from_data_list`, pangnn.py:152-153) is a pair of
slices and one subtraction — no per-step collation on the host.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch


def _unique_sorted(keys: torch.Tensor) -> torch.Tensor:
    return torch.unique(keys)          # sorted ascending, duplicates dropped


def _isin_sorted(sorted_keys: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    if sorted_keys.numel() == 0:
        return torch.zeros_like(q, dtype=torch.bool)
    pos = torch.searchsorted(sorted_keys, q).clamp_(max=sorted_keys.numel() - 1)
    return sorted_keys[pos] == q


def _expand_by_src(rowptr: torch.Tensor, dst_sorted: torch.Tensor, gid: torch.Tensor, node: torch.Tensor):
    """all (gid, t, edge position) for edges node -> t of the src-grouped relation"""
    beg, end = rowptr[node], rowptr[node + 1]
    cnt = end - beg
    total = int(cnt.sum())
    rep = torch.repeat_interleave(torch.arange(node.numel(), device=node.device), cnt)
    off = torch.arange(total, device=node.device) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt)
    epos = beg[rep] + off
    return gid[rep], dst_sorted[epos], epos, rep


def build_subgraphs(num_nodes: int, src, dst, weight, group_id: torch.Tensor, group_member: torch.Tensor,
                    neighbours: int = 1, labels_group_of: Optional[torch.Tensor] = None, pair_src=None,
                    pair_dst=None, require_edges_ge_members: bool = False):
    """(src, dst, weight): normalised similarity relation (self hits already removed).
    (group_id, group_member): the ortholog groups as parallel arrays (group ids 0..G-1, gene node ids).
    Labels as in construct.whole_graph: `labels_group_of[node]` or explicit (pair_src, pair_dst)."""
    dev = src.device
    n = int(num_nodes)
    ok = (src != dst) & (src >= 0) & (dst >= 0) & (src < n) & (dst < n)
    src, dst, weight = src[ok], dst[ok], weight[ok]
    order = torch.argsort(src * n + dst, stable=True)                 # canonical (src, dst) order
    src, dst, weight = src[order], dst[order], weight[order]
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(src, minlength=n), 0)
    has_rows = (rowptr[1:] - rowptr[:-1]) > 0

    ngroups = int(group_id.max().item()) + 1 if group_id.numel() else 0
    gsize = torch.bincount(group_id, minlength=ngroups)
    keep_m = gsize[group_id] > 1                                       # dataset.py:230
    gid, mem = group_id[keep_m], group_member[keep_m]

    # 1. BFS (get_connected_nodes): keys = gid * n + node
    core = _unique_sorted(gid * n + mem)
    frontier = core
    for _ in range(int(neighbours)):
        fg, fn = torch.div(frontier, n, rounding_mode="floor"), frontier % n
        eg, et, _, _ = _expand_by_src(rowptr, dst, fg, fn)
        cand = _unique_sorted(eg * n + et)
        new = cand[~_isin_sorted(core, cand)]
        if new.numel() == 0:
            break
        core = _unique_sorted(torch.cat([core, new]))
        frontier = new
    cg, cn = torch.div(core, n, rounding_mode="floor"), core % n

    # 2. positional neighbours of core nodes (get_neighbour_graph)
    ks = torch.tensor([k for k in range(-neighbours, neighbours + 1) if k != 0], dtype=torch.int64, device=dev)
    nb_g = cg.repeat_interleave(ks.numel())
    nb_a = cn.repeat_interleave(ks.numel())
    nb_b = nb_a + ks.repeat(cn.numel())
    inr = (nb_b >= 0) & (nb_b < n)
    nb_g, nb_a, nb_b = nb_g[inr], nb_a[inr], nb_b[inr]
    extra = _unique_sorted(nb_g * n + nb_b)
    extra = extra[~_isin_sorted(core, extra)]
    xg, xn = torch.div(extra, n, rounding_mode="floor"), extra % n

    # node list per group: core (ascending) then extra (ascending)
    all_g = torch.cat([cg, xg])
    all_n = torch.cat([cn, xn])
    flag = torch.cat([torch.zeros_like(cg), torch.ones_like(xg)])
    o = torch.argsort((all_g * 2 + flag) * n + all_n)
    all_g, all_n = all_g[o], all_n[o]
    nodes_per_group = torch.bincount(all_g, minlength=ngroups)
    node_off_g = torch.zeros(ngroups + 1, dtype=torch.int64, device=dev)
    node_off_g[1:] = torch.cumsum(nodes_per_group, 0)
    local = torch.arange(all_g.numel(), device=dev) - node_off_g[all_g]
    # lookup (gid, node) -> local id
    lk = all_g * n + all_n
    lo = torch.argsort(lk)
    lk_sorted, lk_local = lk[lo], local[lo]

    def lookup(g, v):
        q = g * n + v
        if lk_sorted.numel() == 0:
            return torch.zeros_like(q, dtype=torch.bool), torch.zeros_like(q)
        pos = torch.searchsorted(lk_sorted, q).clamp_(max=lk_sorted.numel() - 1)
        return lk_sorted[pos] == q, lk_local[pos]

    # 3. similarity edges with both endpoints inside the sub-graph (build_edge_index on the sub dict)
    eg, et, epos, rep = _expand_by_src(rowptr, dst, all_g, all_n)
    inside, t_local = lookup(eg, et)
    eg, epos, s_local, t_local = eg[inside], epos[inside], local[rep][inside], t_local[inside]
    e_w = weight[epos].to(torch.float32)
    s_glob, t_glob = src[epos], dst[epos]
    if labels_group_of is not None:
        a, b = labels_group_of[s_glob], labels_group_of[t_glob]
        y = ((a == b) & (a >= 0)).to(torch.float32)
    else:
        k = torch.cat([pair_src * n + pair_dst, pair_dst * n + pair_src]).unique()
        y = _isin_sorted(k, s_glob * n + t_glob).to(torch.float32)
    edges_per_group = torch.bincount(eg, minlength=ngroups)

    # neighbour edges (core v, v +- k), both directions, de-duplicated
    _, a_local = lookup(nb_g, nb_a)
    _, b_local = lookup(nb_g, nb_b)
    big = int(nodes_per_group.max().item()) + 1 if ngroups else 1
    und = _unique_sorted(torch.cat([(nb_g * big + a_local) * big + b_local, (nb_g * big + b_local) * big + a_local]))
    ng = torch.div(und, big * big, rounding_mode="floor")
    na = torch.div(und, big, rounding_mode="floor") % big
    nbb = und % big
    nb_per_group = torch.bincount(ng, minlength=ngroups)

    # 4. drop groups (dataset.py:230,235,248,252)
    sim_rows = torch.zeros(ngroups, dtype=torch.int64, device=dev).index_add_(0, all_g, has_rows[all_n].long())
    keep_g = (gsize > 1) & (nodes_per_group > 0) & (sim_rows > 0)
    if require_edges_ge_members:
        keep_g &= edges_per_group >= gsize
    new_gid = torch.cumsum(keep_g.long(), 0) - 1
    n_sub = int(keep_g.sum())

    def compact(g, *cols):
        m = keep_g[g]
        return (new_gid[g[m]],) + tuple(c[m] for c in cols)

    all_g2, all_n2, local2 = compact(all_g, all_n, local)
    eg2, s2, t2, w2, y2 = compact(eg, s_local, t_local, e_w, y)
    ng2, na2, nb2 = compact(ng, na, nbb)
    node_off = torch.zeros(n_sub + 1, dtype=torch.int64, device=dev)
    node_off[1:] = torch.cumsum(torch.bincount(all_g2, minlength=n_sub), 0)
    edge_off = torch.zeros(n_sub + 1, dtype=torch.int64, device=dev)
    edge_off[1:] = torch.cumsum(torch.bincount(eg2, minlength=n_sub), 0)
    nb_off = torch.zeros(n_sub + 1, dtype=torch.int64, device=dev)
    nb_off[1:] = torch.cumsum(torch.bincount(ng2, minlength=n_sub), 0)
    # dataset-wide node numbering: local id + first node of the sub-graph
    return SubGraphDataset(
        num_graphs=n_sub, node_off=node_off, edge_off=edge_off, nb_off=nb_off, node_global=all_n2,
        edge_index=torch.stack([s2 + node_off[eg2], t2 + node_off[eg2]]), edge_attr=w2, y=y2,
        neighbour_edge_index=torch.stack([na2 + node_off[ng2], nb2 + node_off[ng2]]),
        group_of_graph=torch.nonzero(keep_g).view(-1))


# batches of a GPU-resident data set are collated by one HIP launch (pangnn_collate_subgraphs); False: the index ops
# a CPU-resident data set always uses (what tests/test_construct.py compares the kernel with)
COLLATE_KERNEL = True


class SubGraphDataset:
    """Flat storage of all sub-graphs; `batch(i0, i1)` is the disjoint union of sub-graphs [i0, i1)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __len__(self):
        return self.num_graphs

    @classmethod
    def from_data_list(cls, graphs, device=None) -> "SubGraphDataset":
        """The flat data set of a LIST of per-group sub-graphs as the reference builds them (`UnionGraphDataset`'s
        `generate_sub_graphs`, src/dataset.py:280-310: PyG `Data` objects with `x [n, 1]`, `edge_index [2, e]` and
        `neighbour_edge_index [2, b]` in LOCAL ids, `edge_attr [e]`, `y [e]`; any object with those attributes) — so that
        the reference's own sub-graphs train through `batch()`, `train.GraphedTrainStep` and `train.ReplayedFreshStep`
        (one captured HIP graph for every shuffled mini-batch) without going through this package's graph construction.
        Edge order inside a sub-graph is kept (PyG's `Batch.from_data_list` keeps it too: logits come back in that order);
        one host -> device copy per tensor kind, once per data set."""
        graphs = list(graphs)
        if not graphs:
            raise ValueError("from_data_list needs at least one sub-graph")
        if device is None:
            device = graphs[0].edge_index.device
        for i, gr in enumerate(graphs):
            n = int(gr.x.shape[0])
            ei, nb = gr.edge_index, gr.neighbour_edge_index
            if ei.dim() != 2 or ei.shape[0] != 2 or nb.dim() != 2 or nb.shape[0] != 2:
                raise ValueError(f"sub-graph {i}: edge_index / neighbour_edge_index must be [2, E]")
            if hasattr(gr, "union_edge_index") and not hasattr(gr, "neighbour_edge_index"):
                raise ValueError(f"sub-graph {i} is a --union_edge_weights sub-graph (src/dataset.py:286-299: union_edge_index, "
                                 f"edge_attr = [1 per neighbour edge | similarity weights]); the flat data set holds the default "
                                 f"layout (edge_index + neighbour_edge_index, one weight per similarity edge)")
            if gr.edge_attr.shape[0] != ei.shape[1] or gr.y.shape[0] != ei.shape[1]:
                # (a longer edge_attr is the union layout, whose similarity weights are the LAST e entries: never slice it)
                raise ValueError(f"sub-graph {i}: edge_attr has {gr.edge_attr.shape[0]} and y {gr.y.shape[0]} entries for "
                                 f"{ei.shape[1]} similarity edges")
            for name, t in (("edge_index", ei), ("neighbour_edge_index", nb)):
                if t.numel() and (int(t.min()) < 0 or int(t.max()) >= n):
                    raise ValueError(f"sub-graph {i}: {name} holds ids outside [0, {n})")

        def offsets(counts):
            off = torch.zeros(len(counts) + 1, dtype=torch.int64)
            off[1:] = torch.cumsum(torch.tensor(counts, dtype=torch.int64), 0)
            return off

        node_off = offsets([int(gr.x.shape[0]) for gr in graphs])
        edge_off = offsets([int(gr.edge_index.shape[1]) for gr in graphs])
        nb_off = offsets([int(gr.neighbour_edge_index.shape[1]) for gr in graphs])
        ei = torch.cat([gr.edge_index.to("cpu", torch.int64) + int(node_off[i]) for i, gr in enumerate(graphs)], dim=1)
        nb = torch.cat([gr.neighbour_edge_index.to("cpu", torch.int64) + int(node_off[i]) for i, gr in enumerate(graphs)], dim=1)
        w = torch.cat([gr.edge_attr.to("cpu", torch.float32) for gr in graphs])
        y = torch.cat([gr.y.to("cpu", torch.float32) for gr in graphs])
        dev = torch.device(device)
        return cls(num_graphs=len(graphs), node_off=node_off.to(dev), edge_off=edge_off.to(dev), nb_off=nb_off.to(dev),
                   node_global=None, edge_index=ei.contiguous().to(dev), edge_attr=w.to(dev), y=y.to(dev),
                   neighbour_edge_index=nb.contiguous().to(dev), group_of_graph=None)

    def _host(self):
        """host copies of the three offset tables and the order property of the flat edge list, read back ONCE: collating a
        batch then needs no device -> host synchronisation"""
        h = self.__dict__.get("_host_cache")
        if h is None:
            src = self.edge_index[0]
            h = self._host_cache = SimpleNamespace(
                node=self.node_off.tolist(), edge=self.edge_off.tolist(), nb=self.nb_off.tolist(),
                sorted_by_src=bool((src[1:] >= src[:-1]).all()) if src.numel() > 1 else True)
        return h

    def batch(self, i0: int, i1: int):
        i1 = min(i1, self.num_graphs)
        h = self._host()
        n0, n1 = h.node[i0], h.node[i1]
        e0, e1 = h.edge[i0], h.edge[i1]
        b0, b1 = h.nb[i0], h.nb[i1]
        dev = self.edge_index.device
        ptr = self.node_off[i0:i1 + 1] - n0
        # what this producer knows about the batch (graph.EdgeStructure hints): ids are local and in range by
        # construction; a slice of a source-sorted flat list shifted by a constant is source-sorted
        hints = {"sim": {"valid_ids": True, "sorted_by_src": h.sorted_by_src}, "nb": {"valid_ids": True}}
        if dev.type == "cuda" and COLLATE_KERNEL:
            # one launch (pangnn_collate_subgraphs) instead of seven index ops: the step around it is launch-bound
            from . import _lib
            lib = _lib.load()
            n, e, b, g = n1 - n0, e1 - e0, b1 - b0, i1 - i0
            buf = torch.empty(2 * e + 2 * b + (g + 1) + n, dtype=torch.int64, device=dev)
            x = torch.empty(n, 1, dtype=torch.float32, device=dev)
            ei, nb = buf[:2 * e].view(2, e), buf[2 * e:2 * e + 2 * b].view(2, b)
            ptr, bid = buf[2 * e + 2 * b:2 * e + 2 * b + g + 1], buf[2 * e + 2 * b + g + 1:]
            src_ei, src_nb = self.edge_index, self.neighbour_edge_index
            if not (src_ei.is_contiguous() and src_nb.is_contiguous() and self.node_off.is_contiguous()):
                raise ValueError("SubGraphDataset tensors must be contiguous")
            with _lib.device_guard(dev):
                _lib.check(lib.pangnn_collate_subgraphs(
                    src_ei.data_ptr(), src_ei.shape[1], e0, e, src_nb.data_ptr(), src_nb.shape[1], b0, b,
                    self.node_off.data_ptr() + 8 * i0, g, n0, n, _lib.ptr(ei), _lib.ptr(nb), ptr.data_ptr(),
                    _lib.ptr(bid), _lib.ptr(x), _lib.stream_ptr()), "pangnn_collate_subgraphs")
            return SimpleNamespace(_pangnn_hints=hints, x=x, edge_index=ei, edge_attr=self.edge_attr[e0:e1],
                                   y=self.y[e0:e1], neighbour_edge_index=nb, ptr=ptr, num_graphs=g, batch=bid)
        return SimpleNamespace(
            _pangnn_hints=hints,
            x=torch.ones(n1 - n0, 1, dtype=torch.float32, device=dev),
            edge_index=(self.edge_index[:, e0:e1] - n0).contiguous(),
            edge_attr=self.edge_attr[e0:e1].contiguous(), y=self.y[e0:e1].contiguous(),
            neighbour_edge_index=(self.neighbour_edge_index[:, b0:b1] - n0).contiguous(),
            ptr=ptr, num_graphs=i1 - i0,
            # graph id of every node: position of the node among the graphs' end offsets (no data-dependent output size)
            batch=torch.searchsorted(ptr[1:].contiguous(), torch.arange(n1 - n0, device=dev), right=True))

    def graph(self, i: int):
        return self.batch(i, i + 1)

    # ---- fixed-shape batches (train.ReplayedFreshStep): one set of buffers, one captured HIP graph, every mini-batch
    def padded_spec(self, batch_size: int = 32, graphs=None, slack: Optional[float] = None):
        """(max_graphs, max_nodes, max_edges, max_nb): shapes that hold the disjoint union of ANY `batch_size` sub-graphs out
        of `graphs` (default: all) — the sums of the `batch_size` largest counts, plus the one node that is never real
        (the padding's endpoint).  `slack`: instead `slack` times the MEAN batch (capped by the worst case) — what a
        shuffled batch needs in practice (the sum of 32 draws concentrates around its mean); a batch that does not fit is
        refused by `set_graph_ids`, and the caller keeps a worst-case set of buffers for it (train.ReplayedFreshStep).
        Host arithmetic on the offset tables, which are read back once per data set."""
        h = self._host()
        idx = list(range(self.num_graphs)) if graphs is None else [int(i) for i in graphs]
        bs = int(batch_size)

        def cap(off):
            sizes = sorted((off[i + 1] - off[i] for i in idx), reverse=True)
            worst = sum(sizes[:bs])
            if slack is None or not sizes:
                return worst
            return min(worst, int(float(slack) * bs * sum(sizes) / len(sizes)) + 1)
        return bs, cap(h.node) + 1, max(cap(h.edge), 1), max(cap(h.nb), 1)

    def fits(self, spec, ids) -> bool:
        """whether the disjoint union of the sub-graphs `ids` fits the padded shapes `spec` (host arithmetic)"""
        h = self._host()
        ids = [int(i) for i in ids]
        return len(ids) <= spec[0] and sum(h.node[i + 1] - h.node[i] for i in ids) < spec[1] \
            and sum(h.edge[i + 1] - h.edge[i] for i in ids) <= spec[2] and sum(h.nb[i + 1] - h.nb[i] for i in ids) <= spec[3]

    def structure_index(self):
        """per flat edge of both lists: its position inside its own sub-graph in the by-target and in the by-source order
        (int32; stable — equal keys keep list order).  A property of the data set, computed once with two stable sorts per
        list; with it a batch's CSR orders are written without any sort (pangnn_collate_subgraphs_padded, `orders`)."""
        idx = self.__dict__.get("_structure_index")
        if idx is None:
            dev = self.edge_index.device
            n_total = int(self._host().node[-1])

            def ranks(ei, off):
                cnt = off[1:] - off[:-1]
                gid = torch.repeat_interleave(torch.arange(self.num_graphs, device=dev), cnt)
                first = off[:-1][gid]
                pos = torch.arange(ei.shape[1], device=dev)
                out = []
                for row in (1, 0):            # by target, by source: node ids are data-set wide, so the id alone orders
                    order = torch.argsort(ei[row], stable=True)      # sub-graph after sub-graph, row after row
                    r = torch.empty_like(pos)
                    r[order] = pos
                    out.append((r - first).to(torch.int32).contiguous())
                return out
            assert n_total < 2 ** 31
            sd, ss = ranks(self.edge_index, self.edge_off)
            nd, ns = ranks(self.neighbour_edge_index, self.nb_off)
            idx = self._structure_index = SimpleNamespace(sim_dst=sd, sim_src=ss, nb_dst=nd, nb_src=ns)
        return idx

    def padded_buffers(self, spec, orders: bool = True):
        """the fixed buffers of a padded batch + the device list of sub-graph ids it is collated from.  `orders`: also the
        buffers of both CSR orders (and the decoder's run-sum plans) of both edge lists, written by the collation itself —
        no per-batch sort (`structure_index`); False, or shapes beyond the small-structure limits: built per batch by
        graph.EdgeStructure as for any other batch.
        Layout facts a consumer of the PyG-like fields needs: `ptr` has spec[0] + 1 entries (entries past the real graphs
        repeat the real node count); `batch` gives a PADDED node the id of a dedicated padding segment — the number of real
        graphs of the current batch, which for a full batch equals spec[0] — so per-graph pooling over `batch` must be sized
        `num_graphs` = spec[0] + 1 (what this buffer reports), and its last used row is the padding's."""
        g, n, e, b = spec
        dev = self.edge_index.device
        i64 = torch.empty(2 * e + 2 * b + (g + 1) + n + 8 + g + 16, dtype=torch.int64, device=dev)
        f32 = torch.empty(2 * e + n + 16, dtype=torch.float32, device=dev)
        o = [0]

        def take(buf, k):                       # segments start on 16-byte boundaries
            v = buf[o[0]:o[0] + k]
            o[0] += -(-k * buf.element_size() // 16) * 16 // buf.element_size()
            return v
        ei, nb = take(i64, 2 * e).view(2, e), take(i64, 2 * b).view(2, b)
        ptr, bid, live, ids = take(i64, g + 1), take(i64, n), take(i64, 8), take(i64, g)
        ids.fill_(-1)
        live.zero_()
        o[0] = 0
        w, y, x = take(f32, e), take(f32, e), take(f32, n).view(n, 1)
        hints = {"sim": {"valid_ids": True, "sorted_by_src": self._host().sorted_by_src, "band_width": 0},
                 "nb": {"valid_ids": True, "band_width": 0}}
        buf = SimpleNamespace(_pangnn_hints=hints, x=x, edge_index=ei, edge_attr=w, y=y, neighbour_edge_index=nb, ptr=ptr,
                              batch=bid, live=live, live_edges=live[:1], graph_ids=ids, spec=tuple(spec), num_graphs=g + 1,
                              orders=None)
        from . import _lib
        lib = _lib.load()
        if orders and dev.type == "cuda" and lib.pangnn_structure_small_supported(e, n) \
                and lib.pangnn_structure_small_supported(b, n):
            buf.orders = self._order_buffers(lib, n, e, b, dev)
        return buf

    def _order_buffers(self, lib, n, e, b, dev):
        """device tables of the four CSR orders of a padded batch + the ctypes descriptor the collation takes"""
        import ctypes

        class Order(ctypes.Structure):
            _fields_ = [(k, ctypes.c_void_p) for k in ("rank", "rowptr", "other", "perm", "keys", "part_off", "part_rowptr",
                                                       "last_part")]

        class Orders(ctypes.Structure):
            _fields_ = [("chunk_edges", ctypes.c_int32), ("sim_dst", Order), ("sim_src", Order), ("nb_dst", Order),
                        ("nb_src", Order)]
        ct = int(lib.pangnn_decoder_chunk_tiles_for(e))
        span = 32 * ct
        nc = (e + span - 1) // span
        rk = self.structure_index()
        al = lambda k, m: -(-k // m) * m                                        # noqa: E731
        ea, ba, na, ca = al(e, 4), al(b, 4), al(n + 1, 2), al(nc, 4)
        i32 = torch.empty(6 * ea + 6 * ba + 2 * ca, dtype=torch.int32, device=dev)
        i64 = torch.empty(6 * na + 4, dtype=torch.int64, device=dev)
        seg = {}
        o32 = o64 = 0
        for name, m, ma in (("sim_dst", e, ea), ("sim_src", e, ea), ("nb_dst", b, ba), ("nb_src", b, ba)):
            t = {}
            for f in ("other", "perm", "keys"):
                t[f] = i32[o32:o32 + m]
                o32 += ma
            t["rowptr"] = i64[o64:o64 + n + 1]
            o64 += na
            if name.startswith("sim"):
                t["part_off"] = i32[o32:o32 + nc]
                o32 += ca
                t["part_rowptr"] = i64[o64:o64 + n + 1]
                o64 += na
                t["last_part"] = i64[6 * na + (0 if name == "sim_dst" else 2):][:1]
            seg[name] = t
        desc = Orders()
        desc.chunk_edges = span
        for name in ("sim_dst", "sim_src", "nb_dst", "nb_src"):
            o = getattr(desc, name)
            o.rank = getattr(rk, name).data_ptr()
            for f in ("rowptr", "other", "perm", "keys", "part_off", "part_rowptr", "last_part"):
                setattr(o, f, seg[name][f].data_ptr() if f in seg[name] else None)
        return SimpleNamespace(desc=desc, tables=seg, chunk_tiles=ct, n_chunks=nc, _keep=(i32, i64, rk))

    def _structures_of(self, buf):
        """graph.EdgeStructure objects of the batch's two edge lists over the order tables the collation wrote"""
        from .graph import CSR, EdgeStructure
        g, n, e, b = buf.spec
        out = {}
        for name, ei, m in (("sim", buf.edge_index, e), ("nb", buf.neighbour_edge_index, b)):
            td, ts = buf.orders.tables[name + "_dst"], buf.orders.tables[name + "_src"]
            st = EdgeStructure(ei, n, hints=buf._pangnn_hints[name])
            st._by_dst, st._by_src = CSR(td["rowptr"], td["other"], td["perm"]), CSR(ts["rowptr"], ts["other"], ts["perm"])
            st._by_dst.__dict__["_long"] = st._by_src.__dict__["_long"] = False    # a collated batch of small sub-graphs
            st._small_built = True
            if name == "sim":
                ct, nc = buf.orders.chunk_tiles, buf.orders.n_chunks
                plans = st.__dict__.setdefault("_csr_plans", {})
                for by, t in (("dst", td), ("src", ts)):
                    plan = SimpleNamespace(n_parts=nc + min(n, m), part_off=t["part_off"], part_rowptr=t["part_rowptr"],
                                           keys=t["keys"], chunk_tiles=ct, _last=t["last_part"])
                    plan.n_parts_exact = (lambda pl: (lambda: int(pl._last) + 1))(plan)
                    plans[(by, ct)] = plan
            out[name] = st
        return out

    def collate_padded(self, buf):
        """(re)fill the padded batch `buf` from the sub-graph ids in buf.graph_ids (device): ONE launch, no host read-back;
        capturable — a replay collates whatever ids the list holds at that moment"""
        import ctypes
        from . import _lib
        lib = _lib.load()
        g, n, e, b = buf.spec
        src_ei, src_nb = self.edge_index, self.neighbour_edge_index
        if not (src_ei.is_contiguous() and src_nb.is_contiguous() and self.node_off.is_contiguous()
                and self.edge_off.is_contiguous() and self.nb_off.is_contiguous() and self.edge_attr.is_contiguous()
                and self.y.is_contiguous()):
            raise ValueError("SubGraphDataset tensors must be contiguous")
        if self.edge_attr.dtype != torch.float32 or self.y.dtype != torch.float32:
            raise ValueError("edge_attr / y must be float32")
        with _lib.device_guard(src_ei.device):
            _lib.check(lib.pangnn_collate_subgraphs_padded(
                src_ei.data_ptr(), src_ei.shape[1], src_nb.data_ptr(), src_nb.shape[1], self.edge_attr.data_ptr(),
                self.y.data_ptr(), self.node_off.data_ptr(), self.edge_off.data_ptr(), self.nb_off.data_ptr(),
                self.num_graphs, buf.graph_ids.data_ptr(), g, e, b, n, buf.edge_index.data_ptr(),
                buf.neighbour_edge_index.data_ptr(), buf.edge_attr.data_ptr(), buf.y.data_ptr(), buf.ptr.data_ptr(),
                buf.batch.data_ptr(), buf.x.data_ptr(), buf.live.data_ptr(),
                None if buf.orders is None else ctypes.byref(buf.orders.desc), _lib.stream_ptr()),
                "pangnn_collate_subgraphs_padded")
        # what is cached on the batch object belongs to the previous content of the buffers (same addresses): replace it
        # by the structures over the tables this collation wrote, or drop it (built on first use as for any batch)
        from .graph import forget, structure_key
        for ei in (buf.edge_index, buf.neighbour_edge_index):        # only THIS batch's entries of the identity-keyed cache
            forget(structure_key(ei, n), ei)
        if buf.orders is None:
            buf.__dict__.pop("_pangnn_structs", None)
        else:
            from .graph import register
            sts = self._structures_of(buf)
            buf._pangnn_structs = {name: (structure_key(st.edge_index, st.num_nodes), st) for name, st in sts.items()}
            for st in sts.values():
                register(st)
        return buf

    def set_graph_ids(self, buf, ids):
        """hand the next batch's sub-graph ids (host ints, at most buf.spec[0] and 64) to the device list: one tiny launch whose
        arguments carry the values (no staging buffer that a later call could overwrite before it is read); returns the
        batch's real (nodes, edges, neighbour edges) — host arithmetic"""
        import ctypes
        from . import _lib
        g = buf.spec[0]
        ids = [int(i) for i in ids]
        if not 0 < len(ids) <= min(g, 64) or g > 64:
            raise ValueError(f"a padded batch takes 1..{min(g, 64)} sub-graph ids")
        if min(ids) < 0 or max(ids) >= self.num_graphs:
            raise ValueError("sub-graph id out of range")
        h = self._host()
        n = sum(h.node[i + 1] - h.node[i] for i in ids)
        e = sum(h.edge[i + 1] - h.edge[i] for i in ids)
        b = sum(h.nb[i + 1] - h.nb[i] for i in ids)
        if n >= buf.spec[1] or e > buf.spec[2] or b > buf.spec[3]:
            raise ValueError(f"batch ({n} nodes, {e} edges, {b} neighbour edges) does not fit the padded shapes {buf.spec}")
        vals = (ctypes.c_int64 * g)(*(ids + [-1] * (g - len(ids))))
        with _lib.device_guard(buf.graph_ids.device):
            _lib.check(_lib.load().pangnn_set_i64(buf.graph_ids.data_ptr(), ctypes.cast(vals, ctypes.c_void_p), g,
                                                  _lib.stream_ptr()), "pangnn_set_i64")
        return n, e, b

    def class_balance(self):
        pos = self.y.sum().clamp_min(1)
        return ((self.y == 0).sum() / pos).to(torch.float32)
