"""CPU oracle for panGNN's graph construction — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.

PINNED: every function here is checked against fixtures produced by the reference's own
construction code (tests/golden/*.npz, generator tests/golden/make_fixtures.py).

The reference works on dict-of-dict {gene_str: {gene_str: score}}; this restatement works on the
same relation as integer COO triples (src, dst, score) in dict insertion order, with
`genome_of[node]` standing in for the `id.split('_')[0]` genome prefix.  Where the reference's edge
order is an accident of CPython set iteration (helper.py:428-431) the canonical order is the
lexicographic (src, dst) sort; `canonical_order` gives the permutation.
"""
from __future__ import annotations

import numpy as np
from scipy.special import logsumexp


def canonical_order(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    return np.lexsort((dst, src))


def remove_trivial_cases(src, dst, score, genome_of):
    """preprocessing.py:370-385: per source gene, keep only candidates whose genome holds MORE THAN
    ONE candidate of that source (self hits count as candidates of the own genome)."""
    g = int(genome_of.max()) + 1
    key = src.astype(np.int64) * g + genome_of[dst]
    _, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    keep = cnt[inv] > 1
    return src[keep], dst[keep], score[keep]


def normalize_sim_scores(src, dst, score, genome_of, t=0.8, epsilon=1e-8, pseudo_count=1.0):
    """preprocessing.py:454-548 (q_score_norm path; the flag only affects asserts there):
    per (source gene, candidate genome) group, self hits excluded:
        p = softmax(score / t)  (a single candidate gets p = 1)
        w = -10*log10(clip(1 - p, eps, 1 - eps)) + pseudo_count
    Returns (src, dst, w) with self hits dropped; order within the output follows the input."""
    not_self = src != dst
    src, dst, score = src[not_self], dst[not_self], score[not_self]
    g = int(genome_of.max()) + 1
    key = src.astype(np.int64) * g + genome_of[dst]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    bounds = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1], True])
    w = np.empty(len(src), dtype=np.float64)
    for a, b in zip(bounds[:-1], bounds[1:]):
        idx = order[a:b]
        if b - a > 1:
            x = np.asarray(score[idx], dtype=np.float64) / t          # softmax_with_temperature
            p = np.exp(x - logsumexp(x, axis=-1, keepdims=True))
        else:
            p = np.array([1.0])
        q = np.where(np.isnan(p), -10 * np.log10(1 - epsilon),
                     -10 * np.log10(np.clip(1 - p, epsilon, 1 - epsilon)))
        w[idx] = q + pseudo_count
    return src, dst, w


def whole_graph(num_nodes, nrm_src, nrm_dst, nrm_w, grp_src, grp_dst, neighbours=1):
    """dataset.py:325-384 + preprocessing.py:73-156,264-325: similarity edge_index (self loops and
    unknown genes (-1) skipped), weights, labels, and the positional neighbour edges
    (i, j) for j in [i-n, i+n] ∩ [0, N), INCLUDING j == i, in nested-loop order.
    Similarity edges are returned in canonical (src, dst) order."""
    ok = (nrm_src != nrm_dst) & (nrm_src >= 0) & (nrm_dst >= 0)
    s, d, w = nrm_src[ok], nrm_dst[ok], nrm_w[ok]
    o = canonical_order(s, d)
    s, d, w = s[o], d[o], w[o]
    pair = set(zip(grp_src.tolist(), grp_dst.tolist()))
    y = np.fromiter(((a, b) in pair or (b, a) in pair for a, b in zip(s.tolist(), d.tolist())),
                    dtype=np.float32, count=len(s))
    i = np.repeat(np.arange(num_nodes, dtype=np.int64), 2 * neighbours + 1)
    j = i + np.tile(np.arange(-neighbours, neighbours + 1, dtype=np.int64), num_nodes)
    keep = (j >= 0) & (j < num_nodes)
    nb = np.stack([i[keep], j[keep]])
    return np.stack([s, d]), w.astype(np.float32), y, nb
