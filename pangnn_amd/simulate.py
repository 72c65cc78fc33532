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


def _synteny_order(n: int, genomes: int, frag_size: int, n_shuffle: int, gen):
    """new position of every gene after shuffle_synteny_blocks; returns pos_of[g, p] -> slot."""
    out = torch.empty(genomes, n, dtype=torch.int64)
    base = torch.arange(n)
    for g in range(genomes):
        if n_shuffle <= 1 or frag_size <= 0:
            out[g] = base
            continue
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


def simulate_raw(n: int, genomes: int, frac_pos: float, num_fragments: float = 10, n_shuffle: float = 2,
                 means=(200, 500), seed: int = 0, device="cpu"):
    """raw similarity relation (before remove_trivial_cases) + node metadata."""
    device = torch.device(device)
    gen = torch.Generator(device=device).manual_seed(seed)
    cgen = torch.Generator().manual_seed(seed + 1)
    G, N = int(genomes), int(n) * int(genomes)
    neg_mean, pos_mean = means
    pairs_per_group = G * (G - 1) // 2
    num_pos = pairs_per_group * n
    num_total = math.floor(num_pos / frac_pos)
    m = math.floor((num_total - num_pos) / N)                    # simulate.py:128

    pos_of = _synteny_order(n, G, math.floor(n / num_fragments), int(n_shuffle), cgen).to(device)
    node = (torch.arange(G, device=device).view(-1, 1) * n + pos_of)        # node id of gene (g, p)

    # ---- positives: all genome pairs at every position
    ga, gb = torch.triu_indices(G, G, offset=1, device=device)
    p_all = torch.arange(n, device=device)
    a = node[ga][:, p_all].reshape(-1)                           # [pairs * n]
    b = node[gb][:, p_all].reshape(-1)
    ps = _gamma_int(pos_mean, a.numel(), gen, device)
    adj = (gb - ga == 1)                                         # adjacent pairs can be hit by a negative
    adj_slot = torch.full((G,), -1, dtype=torch.int64, device=device)
    adj_slot[ga[adj]] = torch.nonzero(adj).view(-1)

    # ---- negatives: genes of genomes 0..G-2 -> k distinct positions of genome g+1
    n_src = (G - 1) * n
    k = _neg_binomial(0.2, float(m), n_src, gen, device).clamp_(1, n) if m > 0 else \
        torch.ones(n_src, dtype=torch.int64, device=device)
    owner, q = _distinct_positions(k, n, gen, device)
    sg, sp = owner // n, owner % n
    ns = _gamma_int(neg_mean, owner.numel(), gen, device)
    hit = q == sp                                                # negative lands on the own ortholog
    if bool(hit.any()):
        ps[adj_slot[sg[hit]] * n + sp[hit]] = ns[hit]            # overwrite that pair's score
    keep = ~hit
    na = node[sg[keep], sp[keep]]
    nb = node[sg[keep] + 1, q[keep]]
    nsc = ns[keep]

    src = torch.cat([a, b, na, nb])
    dst = torch.cat([b, a, nb, na])
    score = torch.cat([ps, ps, nsc, nsc])
    genome_of = torch.arange(N, device=device) // n
    group_of = torch.empty(N, dtype=torch.int64, device=device)
    group_of[node.reshape(-1)] = p_all.repeat(G)
    return SimpleNamespace(num_nodes=N, src=src, dst=dst, score=score, genome_of=genome_of, group_of=group_of,
                           mean_neg_per_gene=m)


def simulate_graph(n: int, genomes: int, frac_pos: float, num_fragments: float = 10, n_shuffle: float = 2,
                   neighbours: int = 1, seed: int = 0, device="cpu", temperature: float = 0.8):
    """whole simulated graph as the reference's `generate_graphs()` would emit it (dataset.py:157-158):
    x, edge_index (canonical order), edge_attr, y, neighbour_edge_index, class_balance."""
    raw = simulate_raw(n, genomes, frac_pos, num_fragments, n_shuffle, seed=seed, device=device)
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
