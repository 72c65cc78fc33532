"""Seeded, vectorised synthetic pan-genome generator (`--simulate_dataset n G frac frags shuffled`).

Restates /root/reference/src/simulate.py:83-230 + src/dataset.py:58-73 as tensor operations so that
BASELINE.json configs 2, 4 and 5 can be produced at all (the reference's generator is nested Python
loops over dicts, and its graph construction is quadratic: SURVEY.md §3.2).  The reference never
seeds its RNGs, so parity is DISTRIBUTIONAL: same edge model, same marginal laws, checked against the
reference-generated fixtures by moments / degree statistics (tests/test_construct.py).

Model (node = gene; genome-major ids; ortholog group = position before synteny shuffling):
  * positives: every pair of genes at the same position, both directions, one raw score per pair
    int(Gamma(shape = mu^2/1e4, scale = 1e4/mu)), mu = 500                      simulate.py:11-17,156-168
  * negatives: every gene of genomes 0..G-2 draws k = clip(NegBin(0.2, 0.2/(m+0.2)), 1, n) distinct
    positions of the NEXT genome, both directions, mu = 200; m = floor(#neg/#genes) from `frac`;
    a negative that hits the gene's own ortholog overwrites that pair's score   simulate.py:120-133,170-190
  * synteny: each genome is cut into blocks of floor(n/frags) genes and `shuffled` randomly chosen
    blocks are permuted among themselves; node ids follow the shuffled order     simulate.py:202-230, dataset.py:66-73
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch

from . import construct


def _gamma_int(mean: float, count: int, gen, device):
    shape, scale = (mean ** 2) / 10000.0, 10000.0 / mean       # dispersion 10000 (simulate.py:156,182)
    g = torch._standard_gamma(torch.full((count,), shape, dtype=torch.float64, device=device), generator=gen)
    return torch.floor(g * scale)                                # int(x) of a positive float


def _neg_binomial(n: float, mean: float, count: int, gen, device):
    """numpy negative_binomial(n, p) with p = n/(mean+n): Poisson(Gamma(n, (1-p)/p))."""
    lam = torch._standard_gamma(torch.full((count,), n, dtype=torch.float64, device=device), generator=gen)
    lam = lam * (mean / n)
    return torch.poisson(lam, generator=gen).to(torch.int64)


def _distinct_positions(k: torch.Tensor, n: int, gen, device):
    """for every source i, k[i] DISTINCT uniform positions in [0, n) (random.sample semantics).
    Returns (owner, pos), grouped by owner.  Sources asking for more than n/2 positions (the clipped
    tail of the negative binomial) get an explicit random permutation; the rest use rejection."""
    num = k.numel()
    big = torch.nonzero(k > n // 2).view(-1)
    big_owner, big_pos = [], []
    for i in big.tolist():
        ki = int(k[i].item())
        big_owner.append(torch.full((ki,), i, dtype=torch.int64, device=device))
        big_pos.append(torch.randperm(n, generator=gen, device=device)[:ki])
    k_small = k.clone()
    k_small[big] = 0
    ids = torch.arange(num, device=device)
    owner = torch.repeat_interleave(ids, k_small)
    pos = torch.randint(0, n, (owner.numel(),), generator=gen, device=device)
    for _ in range(200):
        key = torch.unique(owner * n + pos)                      # sorted, duplicates dropped
        owner = key // n
        pos = key - owner * n
        miss = k_small - torch.bincount(owner, minlength=num)
        if int(miss.max().item()) <= 0:
            break
        extra_owner = torch.repeat_interleave(ids, miss.clamp_min(0))
        owner = torch.cat([owner, extra_owner])
        pos = torch.cat([pos, torch.randint(0, n, (extra_owner.numel(),), generator=gen, device=device)])
    else:
        raise RuntimeError("rejection sampling of distinct positions did not converge")
    if big_owner:
        owner = torch.cat([owner] + big_owner)
        pos = torch.cat([pos] + big_pos)
        o = torch.argsort(owner * n + pos)
        owner, pos = owner[o], pos[o]
    return owner, pos


def _mix(seed: int, *parts: int) -> int:
    """counter-based seed of one piece of the data set (a genome, a genome pair): any rank can draw any piece"""
    h = (int(seed) * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF
    for p_ in parts:
        h ^= (int(p_) + 0x9E3779B97F4A7C15 + ((h << 6) & 0xFFFFFFFFFFFFFFFF) + (h >> 2)) & 0xFFFFFFFFFFFFFFFF
        h = (h * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    return h & 0x7FFFFFFFFFFFFFFF


def _synteny_order(n: int, genomes: int, frag_size: int, n_shuffle: int, seed: int):
    """new position of every gene after shuffle_synteny_blocks; returns pos_of[g, p] -> slot.  One generator per
    genome (seeded by (seed, genome)): the order of a genome does not depend on which other genomes are drawn."""
    out = torch.empty(genomes, n, dtype=torch.int64)
    base = torch.arange(n)
    for g in range(genomes):
        if n_shuffle <= 1 or frag_size <= 0:
            out[g] = base
            continue
        gen = torch.Generator().manual_seed(_mix(seed, 1, g))
        blocks = list(torch.split(base, frag_size))
        chosen = torch.randperm(len(blocks), generator=gen)[:n_shuffle].tolist()
        perm = torch.randperm(len(chosen), generator=gen).tolist()
        moved = [blocks[chosen[j]] for j in perm]
        for slot, blk in zip(chosen, moved):
            blocks[slot] = blk
        order = torch.cat(blocks)                                 # order[slot] = original position
        inv = torch.empty(n, dtype=torch.int64)
        inv[order] = base
        out[g] = inv
    return out


def _mean_negatives(n: int, G: int, frac_pos: float) -> int:
    num_pos = G * (G - 1) // 2 * n
    return math.floor((math.floor(num_pos / frac_pos) - num_pos) / (n * G))      # simulate.py:128


def _adjacent_block(b: int, n: int, node, m: int, means, seed: int, device):
    """everything the relation holds between genome b and genome b + 1: the ortholog pair of every position and the
    negatives genome b draws in genome b + 1 (both directions; simulate.py:156-190).  Drawn from its own generator
    (seed, b), so a rank that owns genomes around b produces exactly what a whole-graph run produces for them."""
    neg_mean, pos_mean = means
    gen = torch.Generator(device=device).manual_seed(_mix(seed, 2, b))
    p_all = torch.arange(n, device=device)
    a, bb = node[b], node[b + 1]                                   # node ids of position p in the two genomes
    ps = _gamma_int(pos_mean, n, gen, device)
    k = _neg_binomial(0.2, float(m), n, gen, device).clamp_(1, n) if m > 0 else \
        torch.ones(n, dtype=torch.int64, device=device)
    sp, q = _distinct_positions(k, n, gen, device)                 # source position, target position
    ns = _gamma_int(neg_mean, sp.numel(), gen, device)
    hit = q == sp                                                  # negative lands on the own ortholog
    if bool(hit.any()):
        ps[sp[hit]] = ns[hit]                                      # overwrite that pair's score
    keep = ~hit
    na, nb, nsc = a[sp[keep]], bb[q[keep]], ns[keep]
    src = torch.cat([a[p_all], bb[p_all], na, nb])
    dst = torch.cat([bb[p_all], a[p_all], nb, na])
    return src, dst, torch.cat([ps, ps, nsc, nsc])


def simulate_raw(n: int, genomes: int, frac_pos: float, num_fragments: float = 10, n_shuffle: float = 2,
                 means=(200, 500), seed: int = 0, device="cpu", blocks=None, mean_neg=None):
    """raw similarity relation (before remove_trivial_cases) + node metadata.
    `blocks` (list of b): only the adjacent genome pairs (b, b + 1) listed — what survives remove_trivial_cases can
    only come from those (a gene has exactly one candidate, its ortholog, in every non-adjacent genome), which is
    what rank-local generation relies on; None = the whole relation, non-adjacent ortholog pairs included."""
    device = torch.device(device)
    G, N = int(genomes), int(n) * int(genomes)
    m = _mean_negatives(n, G, frac_pos) if mean_neg is None else int(mean_neg)   # override: a slice of a bigger data set
    pos_of = _synteny_order(n, G, math.floor(n / num_fragments), int(n_shuffle), seed).to(device)
    node = (torch.arange(G, device=device).view(-1, 1) * n + pos_of)        # node id of gene (g, p)
    p_all = torch.arange(n, device=device)

    parts = [_adjacent_block(b, n, node, m, means, seed, device) for b in (range(G - 1) if blocks is None else blocks)]
    if blocks is None and G > 2:
        # ortholog pairs of non-adjacent genomes (one generator for all of them)
        ga, gb = torch.triu_indices(G, G, offset=1, device=device)
        far = gb - ga > 1
        ga, gb = ga[far], gb[far]
        gen = torch.Generator(device=device).manual_seed(_mix(seed, 3))
        a = node[ga][:, p_all].reshape(-1)
        b_ = node[gb][:, p_all].reshape(-1)
        ps = _gamma_int(means[1], a.numel(), gen, device)
        parts.append((torch.cat([a, b_]), torch.cat([b_, a]), torch.cat([ps, ps])))
    if parts:
        src, dst, score = (torch.cat([p_[i] for p_ in parts]) for i in range(3))
    else:
        src = dst = torch.zeros(0, dtype=torch.int64, device=device)
        score = torch.zeros(0, dtype=torch.float64, device=device)
    genome_of = torch.arange(N, device=device) // n
    group_of = torch.empty(N, dtype=torch.int64, device=device)
    group_of[node.reshape(-1)] = p_all.repeat(G)
    return SimpleNamespace(num_nodes=N, src=src, dst=dst, score=score, genome_of=genome_of, group_of=group_of,
                           mean_neg_per_gene=m)


def simulate_shard(n: int, genomes: int, frac_pos: float, num_fragments: float = 10, n_shuffle: float = 2,
                   neighbours: int = 1, seed: int = 0, device="cpu", temperature: float = 0.8, rank: int = 0,
                   world: int = 1, mean_neg=None, bounds=None):
    """Rank `rank`'s destination-partitioned shard of the simulated graph WITHOUT building the whole graph: the rank
    draws only the genome pairs that touch its node range (per-pair generators), normalises them (every
    (source, candidate genome) group lies inside one pair) and keeps the edges whose target it owns.  Bit-identical to
    `dist.partition_graph(simulate_graph(...), rank, world[, bounds])` (tests/test_construct.py).  Global quantities a shard
    cannot know are left as LOCAL counts for the caller to all-reduce: `n_pos_local`, `e_sim_local`
    (-> class_balance = (E - P) / P and e_sim_total)."""
    device = torch.device(device)
    G, N = int(genomes), int(n) * int(genomes)
    if bounds is None:
        n_local = (N + world - 1) // world
        lo, hi = rank * n_local, min((rank + 1) * n_local, N)
        n_pad = n_local * world
    else:                                   # unequal node ranges (dist.balanced_bounds)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        n_local, n_pad = hi - lo, None
    g_lo, g_hi = lo // n, max(lo, hi - 1) // n
    blocks = list(range(max(g_lo - 1, 0), min(g_hi, G - 2) + 1)) if hi > lo else []
    raw = simulate_raw(n, genomes, frac_pos, num_fragments, n_shuffle, seed=seed, device=device, blocks=blocks,
                       mean_neg=mean_neg)
    s, d, sc = construct.remove_trivial_cases(raw.src, raw.dst, raw.score, raw.genome_of)
    s, d, w = construct.normalize_sim_scores(s, d, sc, raw.genome_of, t=temperature)
    own = (d >= lo) & (d < hi) & (s != d)
    s, d, w = s[own], d[own], w[own]
    o = construct.canonical_order(s, d, N)
    s, d, w = s[o], d[o], w[o]
    gs, gd = raw.group_of[s], raw.group_of[d]
    y = ((gs == gd) & (gs >= 0)).to(torch.float32)
    # positional neighbours (i, j), j in [i - k, i + k], of the targets j this rank owns, in whole-graph order
    k = int(neighbours)
    i = torch.arange(max(lo - k, 0), min(hi + k, N), dtype=torch.int64, device=device).repeat_interleave(2 * k + 1)
    j = i + torch.arange(-k, k + 1, dtype=torch.int64, device=device).repeat(i.numel() // (2 * k + 1))
    keep = (j >= lo) & (j < hi)
    x = torch.zeros(n_local, 1, dtype=torch.float32, device=device)
    x[: hi - lo] = 1.0
    return SimpleNamespace(
        x=x, edge_index=torch.stack([s, d - lo]).contiguous(), edge_attr=w.to(torch.float32).contiguous(), y=y,
        neighbour_edge_index=torch.stack([i[keep], j[keep] - lo]).contiguous(),
        n_local=n_local, n_pad=n_pad, n_global=N, lo=lo, hi=hi, rank=rank, world=world,
        bounds=None if bounds is None else [int(b) for b in bounds],
        e_sim_local=int(s.numel()), n_pos_local=int(y.sum().item()), e_sim_total=None, owned_mask=None,
        genome_of=raw.genome_of)


def simulate_graph(n: int, genomes: int, frac_pos: float, num_fragments: float = 10, n_shuffle: float = 2,
                   neighbours: int = 1, seed: int = 0, device="cpu", temperature: float = 0.8, mean_neg=None,
                   adjacent_only: bool = False):
    """whole simulated graph as the reference's `generate_graphs()` would emit it (dataset.py:157-158):
    x, edge_index (canonical order), edge_attr, y, neighbour_edge_index, class_balance."""
    # adjacent_only: skip the ortholog pairs of non-adjacent genomes — remove_trivial_cases drops every one of them
    # anyway (same graph), and at config-5 scale they are most of the raw relation
    raw = simulate_raw(n, genomes, frac_pos, num_fragments, n_shuffle, seed=seed, device=device, mean_neg=mean_neg,
                       blocks=list(range(int(genomes) - 1)) if adjacent_only else None)
    g = construct.build_from_raw(raw.num_nodes, raw.src, raw.dst, raw.score, raw.genome_of,
                                 group_of=raw.group_of, neighbours=neighbours, t=temperature)
    pos = g.y.sum()
    g.class_balance = ((g.y == 0).sum() / pos.clamp_min(1)).to(torch.float32)      # dataset.py:346
    g.genome_of = raw.genome_of
    return g


def simulate_subgraph_dataset(n: int, genomes: int, frac_pos: float, num_fragments: float = 10, n_shuffle: float = 2,
                              neighbours: int = 1, seed: int = 0, device="cpu", temperature: float = 0.8):
    """the reference's `--train` regime on a simulated pan-genome (dataset.py:137-147): one sub-graph per
    ortholog group, stored flat (pangnn_amd/subgraphs.py) so that 32-graph mini-batches are slices"""
    from . import subgraphs
    raw = simulate_raw(n, genomes, frac_pos, num_fragments, n_shuffle, seed=seed, device=device)
    s, d, sc = construct.remove_trivial_cases(raw.src, raw.dst, raw.score, raw.genome_of)
    s, d, w = construct.normalize_sim_scores(s, d, sc, raw.genome_of, t=temperature)
    nodes = torch.arange(raw.num_nodes, device=raw.src.device)
    return subgraphs.build_subgraphs(raw.num_nodes, s, d, w, raw.group_of, nodes, neighbours=neighbours,
                                     labels_group_of=raw.group_of)
