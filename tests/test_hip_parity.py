"""GPU suite (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on identical
inputs.  Integer / index results bit-exact; fp32 results within the north-star tolerance
(1e-4 on link-prediction logits; the same bound relative to magnitude for intermediate tensors)."""
import os

import numpy as np
import pytest
import torch

from conftest import copy_graph, random_graph, sub_graphs_from_golden, whole_graph_from_golden
from oracle import gcn_oracle as go

pytestmark = pytest.mark.gpu

ATOL = 1e-4          # north_star: link-prediction logits within 1e-4 fp32
RTOL = 1e-4


def dev():
    return torch.device("cuda:0")


def close(a, b, atol=ATOL, rtol=RTOL):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    ok = torch.allclose(a, b, atol=atol, rtol=rtol)
    if not ok:
        err = (a - b).abs()
        print("max abs err", err.max().item(), "max |ref|", b.abs().max().item())
    return ok


# ---------------------------------------------------------------- structure build (bit-exact)
@pytest.mark.parametrize("n,e,seed", [(1, 0, 0), (7, 1, 1), (100, 999, 2), (5000, 200000, 3), (33, 5000, 4)])
@pytest.mark.parametrize("group_by", [0, 1])
def test_csr_build_is_a_stable_sort(n, e, seed, group_by):
    from pangnn_amd.graph import build_csr
    ei, _ = random_graph(n, e, seed=seed, hub=min(e, 1500))
    csr = build_csr(ei.to(dev()), n, group_by)
    key = ei[group_by].numpy()
    perm = np.argsort(key, kind="stable")
    rowptr = np.searchsorted(key[perm], np.arange(n + 1), side="left")
    assert np.array_equal(csr.rowptr.cpu().numpy(), rowptr)
    assert np.array_equal(csr.perm.cpu().numpy().astype(np.int64), perm)
    assert np.array_equal(csr.other.cpu().numpy().astype(np.int64), ei[1 - group_by].numpy()[perm])


def test_csr_build_rejects_out_of_range_ids():
    from pangnn_amd.graph import build_csr
    ei = torch.tensor([[0, 1, 9], [1, 2, 0]], device=dev())
    with pytest.raises(ValueError):
        build_csr(ei, 5, 1)


SMALL_CASES = [(1, 1, 0, False), (7, 1, 1, False), (2, 2, 2, True), (100, 31, 3, True), (100, 32, 4, False),
               (100, 33, 5, True), (300, 511, 6, True), (300, 512, 7, False), (300, 513, 8, True), (2000, 4096, 9, True),
               (5000, 4097, 10, False), (33, 5000, 11, True), (65536, 16384, 12, False), (65536, 16384, 13, True),
               (40, 16384, 14, True), (60000, 9000, 15, True)]


@pytest.mark.parametrize("n,e,seed,sort_src", SMALL_CASES)
def test_small_structure_build_equals_the_general_build(n, e, seed, sort_src):
    """pangnn_structure_small (one launch: both CSR orders + both chunk plans of a mini-batch sized list) against
    pangnn_csr_build's radix sort and the index-op plan construction: every table, entry for entry — and against numpy's
    stable argsort.  Edge counts around the 32-edge tile, the 512-edge chunk and the power-of-two padding; hubs; isolated
    nodes at the end of the id range; source-sorted and unsorted lists."""
    from pangnn_amd import graph as G
    from pangnn_amd import functional as PF
    ei, _ = random_graph(n, e, seed=seed, hub=min(e, 1500))
    if sort_src:
        ei = ei[:, torch.argsort(ei[0], stable=True)]
    ei = ei.to(dev())
    ct = PF.d16_chunk(e)
    small = G.EdgeStructure(ei, n, hints={"sorted_by_src": sort_src})
    assert small._small_build() and small._small_built
    old, G.SMALL_STRUCTURE = G.SMALL_STRUCTURE, False
    try:
        gen = G.EdgeStructure(ei, n, hints={"sorted_by_src": sort_src})
        assert not gen._small_build()
        pairs = [(small.by_dst, gen.by_dst), (small.by_src, gen.by_src)]
        plans = [(small.csr_plan("dst", ct), gen.csr_plan("dst", ct)), (small.csr_plan("src", ct), gen.csr_plan("src", ct))]
        if sort_src:
            plans.append((small.runsum_plan(ct), gen.runsum_plan(ct)))
            assert small.runsum_plan(ct) is small.csr_plan("src", ct)
        else:
            assert small.runsum_plan(ct) is None and gen.runsum_plan(ct) is None
        # a chunk size the small build does not emit goes through the general plan code on the small build's tables
        plans.append((small.csr_plan("dst", 1), gen.csr_plan("dst", 1)))
    finally:
        G.SMALL_STRUCTURE = old
    for a, b in pairs:
        assert a.rowptr.dtype == b.rowptr.dtype and a.other.dtype == b.other.dtype and a.perm.dtype == b.perm.dtype
        assert torch.equal(a.rowptr, b.rowptr) and torch.equal(a.other, b.other) and torch.equal(a.perm, b.perm)
    for a, b in plans:
        assert a.n_parts == b.n_parts and a.chunk_tiles == b.chunk_tiles
        for f in ("part_off", "part_rowptr", "keys"):
            x, y = getattr(a, f), getattr(b, f)
            assert x.dtype == y.dtype and x.shape == y.shape and torch.equal(x, y), f
        assert a.n_parts_exact() == b.n_parts_exact() <= a.n_parts
    for group_by, csr in ((1, small.by_dst), (0, small.by_src)):
        key = ei[group_by].cpu().numpy()
        perm = np.argsort(key, kind="stable")
        assert np.array_equal(csr.perm.cpu().numpy().astype(np.int64), perm)
        assert np.array_equal(csr.rowptr.cpu().numpy(), np.searchsorted(key[perm], np.arange(n + 1), side="left"))


def test_small_structure_build_random_sizes_against_numpy():
    """60 random (N, E) inside the kernel's limits, random order, hubs, duplicate edges: both CSR orders against numpy's
    stable argsort, and the chunk plans against a direct numpy restatement of the plan definition"""
    from pangnn_amd import graph as G
    from pangnn_amd import functional as PF
    rng = np.random.default_rng(20261004)
    for trial in range(60):
        e = int(rng.integers(1, 16385)) if trial % 3 else int(2 ** rng.integers(0, 15))
        n = int(rng.integers(1, 65537)) if trial % 4 else int(rng.integers(1, 40))
        ei_np = rng.integers(0, n, size=(2, e), dtype=np.int64)
        if trial % 5 == 0:
            ei_np[1, : e // 2] = ei_np[1, 0]                                   # a hub target
        if trial % 2:
            ei_np = ei_np[:, np.argsort(ei_np[0], kind="stable")]
        st = G.EdgeStructure(torch.from_numpy(ei_np).to(dev()), n)
        assert st._small_build()
        ct = PF.d16_chunk(e)
        span = 32 * ct
        for group_by, csr, by in ((1, st.by_dst, "dst"), (0, st.by_src, "src")):
            key = ei_np[group_by]
            perm = np.argsort(key, kind="stable")
            ks = key[perm]
            assert np.array_equal(csr.perm.cpu().numpy().astype(np.int64), perm), (trial, n, e, by)
            assert np.array_equal(csr.other.cpu().numpy().astype(np.int64), ei_np[1 - group_by][perm])
            rowptr = np.searchsorted(ks, np.arange(n + 1), side="left")
            assert np.array_equal(csr.rowptr.cpu().numpy(), rowptr)
            plan = st.csr_plan(by, ct)
            start = (np.arange(e) % span == 0)
            start[1:] |= ks[1:] != ks[:-1]
            pid = np.cumsum(start) - 1
            assert np.array_equal(plan.keys.cpu().numpy().astype(np.int64), ks)
            assert np.array_equal(plan.part_off.cpu().numpy().astype(np.int64), pid[::span])
            pid_ext = np.concatenate([pid, [pid[-1] + 1]])
            assert np.array_equal(plan.part_rowptr.cpu().numpy(), pid_ext[rowptr]), (trial, n, e, by)
            assert plan.n_parts_exact() == pid[-1] + 1 <= plan.n_parts


def test_small_structure_build_limits_and_bad_ids():
    from pangnn_amd import graph as G
    from pangnn_amd import _lib
    lib = _lib.load()
    assert lib.pangnn_structure_small_supported(16384, 65536) and not lib.pangnn_structure_small_supported(16385, 10)
    assert not lib.pangnn_structure_small_supported(10, 65537) and not lib.pangnn_structure_small_supported(0, 10)
    ei, _ = random_graph(50, 20000, seed=1)
    big = G.EdgeStructure(ei.to(dev()), 50)
    assert not big._small_build() and big.by_dst.perm.shape[0] == 20000          # too many edges: the general build
    bad = torch.tensor([[0, 1, 9], [1, 2, 0]], device=dev())
    with pytest.raises(ValueError):
        G.EdgeStructure(bad, 5).by_dst
    with pytest.raises(ValueError):
        G.EdgeStructure(torch.tensor([[0, 1, 2], [1, -2, 0]], device=dev()), 5).by_src
    ok = G.EdgeStructure(torch.tensor([[0, 1, 4], [1, 2, 0]], device=dev()), 5)
    assert ok.by_dst.rowptr.tolist() == [0, 1, 2, 3, 3, 3]


# ---------------------------------------------------------------- gcn_norm
@pytest.mark.parametrize("weighted", [True, False])
def test_gcn_norm_matches_oracle(weighted):
    from pangnn_amd.graph import EdgeStructure
    n, e = 3000, 40000
    ei, w = random_graph(n, e, seed=5, hub=2500)
    st = EdgeStructure(ei.to(dev()), n)
    nrm = st.gcn_norm(w.to(dev()) if weighted else None)
    ref = go.gcn_norm(ei, w if weighted else None, n)
    assert close(nrm.orig, ref, atol=1e-7, rtol=1e-5)
    assert close(nrm.by_dst, ref[st.by_dst.perm.cpu().long()], atol=1e-7, rtol=1e-5)
    assert close(nrm.by_src, ref[st.by_src.perm.cpu().long()], atol=1e-7, rtol=1e-5)
    deg = torch.zeros(n).scatter_add_(0, ei[1], w if weighted else torch.ones(e))
    iso = deg == 0
    assert iso.any() and torch.equal(nrm.deg_inv_sqrt.cpu()[iso], torch.zeros(int(iso.sum())))


# ---------------------------------------------------------------- propagate (the ★ kernel)
@pytest.mark.parametrize("F", [16, 32, 64, 128, 256, 20, 3])
@pytest.mark.parametrize("n,e,hub", [(64, 0, None), (1, 5, None), (513, 7000, 3000), (2000, 30000, None),
                                     (3001, 9000, None), (700, 5000, 900)])      # last two: thin rows (< 8 per row)
def test_spmm_forward_backward_match_oracle(F, n, e, hub):
    import pangnn_amd
    from pangnn_amd import functional as PF
    from pangnn_amd.graph import EdgeStructure
    torch.manual_seed(F + n)
    ei, w = random_graph(n, e, seed=F + e, hub=hub)
    x = torch.randn(n, F)
    b = torch.randn(F)
    xg = x.clone().to(dev()).requires_grad_(True)
    bg = b.clone().to(dev()).requires_grad_(True)
    st = EdgeStructure(ei.to(dev()), n)
    out = PF.propagate(xg, bg, st, st.gcn_norm(w.to(dev())))
    xr, br = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = go.propagate_add(xr, ei, go.gcn_norm(ei, w, n)) + br
    assert close(out, ref)
    g = torch.randn(n, F)
    out.backward(g.to(dev()))
    ref.backward(g)
    assert close(xg.grad, xr.grad) and close(bg.grad, br.grad)


def test_spmm_is_bitwise_reproducible():
    from pangnn_amd import functional as PF
    from pangnn_amd.graph import EdgeStructure
    n, e = 4000, 300000
    ei, w = random_graph(n, e, seed=11, hub=5000)
    x = torch.randn(n, 128, device=dev())
    outs = []
    for _ in range(3):
        st = EdgeStructure(ei.to(dev()), n)          # rebuilt: the sort is stable, so is the sum order
        outs.append(PF.propagate(x, None, st, st.gcn_norm(w.to(dev()))))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_spmm_linearity_at_scale():
    """size-independent property at a size the oracle would not finish quickly:
    A(ax + by) == a A x + b A y, and A 1 == row sums of the normalised weights."""
    from pangnn_amd import functional as PF
    from pangnn_amd.graph import EdgeStructure
    n, e = 200000, 6000000
    gen = torch.Generator(device="cpu").manual_seed(0)
    ei = torch.randint(0, n, (2, e), generator=gen).to(dev())
    w = (torch.rand(e, generator=gen) * 80 + 1).to(dev())
    st = EdgeStructure(ei, n)
    nrm = st.gcn_norm(w)
    x, y = torch.randn(n, 128, device=dev()), torch.randn(n, 128, device=dev())
    lhs = PF.propagate(2.0 * x - 3.0 * y, None, st, nrm)
    rhs = 2.0 * PF.propagate(x, None, st, nrm) - 3.0 * PF.propagate(y, None, st, nrm)
    assert close(lhs, rhs, atol=1e-4, rtol=1e-4)
    ones = PF.propagate(torch.ones(n, 16, device=dev()), None, st, nrm)
    rowsum = torch.zeros(n, device=dev()).index_add_(0, ei[1], nrm.orig)
    assert close(ones[:, 0], rowsum, atol=1e-4, rtol=1e-4)
    # transpose identity <A x, y> == <x, A^T y>
    ax = PF.propagate(x, None, st, nrm)
    aty = PF.spmm_csr(st.by_src, nrm.by_src, y, n)
    l, r = (ax.double() * y.double()).sum().item(), (x.double() * aty.double()).sum().item()
    assert abs(l - r) <= 1e-5 * max(abs(l), abs(r), 1.0)


# ---------------------------------------------------------------- GCNConv module
@pytest.mark.parametrize("weighted", [True, False])
def test_gcnconv_module_matches_oracle_and_dense(weighted):
    import pangnn_amd
    torch.manual_seed(1)
    n, e = 300, 4000
    ei, w = random_graph(n, e, seed=7)
    conv = pangnn_amd.GCNConv(64, 128).to(dev())
    with torch.no_grad():
        conv.bias.uniform_(-1, 1)
    x = torch.randn(n, 64)
    out = conv(x.to(dev()), ei.to(dev()), w.to(dev()) if weighted else None)
    W, b = conv.lin.weight.detach().cpu(), conv.bias.detach().cpu()
    assert close(out, go.gcn_conv(x, ei, w if weighted else None, W, b))
    assert close(out, go.gcn_conv_dense(x.double(), ei, w.double() if weighted else None, W.double(), b.double()))


@pytest.mark.parametrize("k", [1, 2, 3])
@pytest.mark.parametrize("F", [64, 128])
@pytest.mark.parametrize("n", [1, 2, 5, 1000, 100003])
def test_band_propagate_is_the_generic_propagate_bit_for_bit(k, F, n):
    """the positional-neighbour graph of a whole genome set (dataset.py:356-361) is a band matrix: pangnn_band_propagate
    must give the generic CSR kernels' sums in the same order — forward, transposed (backward) and bias gradient —
    for f32 and bf16-stored rows"""
    from pangnn_amd import construct, functional as PF
    from pangnn_amd.graph import EdgeStructure
    if k >= n:
        pytest.skip("band wider than the graph")
    ei = construct.neighbour_edges(n, k, device=dev())
    st = EdgeStructure(ei, n)
    assert st.band_width() == k
    norm = st.gcn_norm(None)
    torch.manual_seed(n + k + F)
    for dt in (torch.float32, torch.bfloat16):
        x0 = torch.randn(n, F, device=dev()).to(dt)
        b0 = torch.randn(F, device=dev())
        go_ = torch.randn(n, F, device=dev())
        res = []
        for fn in (PF.band_propagate, PF.propagate):
            x, b = x0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            y = fn(x, b, st, norm)
            y.backward(go_)
            res.append((y.detach(), x.grad, b.grad))
        assert res[0][0].dtype == torch.float32 and res[0][1].dtype == dt
        if dt == torch.float32:          # same sums, same order as the generic thin-row kernel
            assert torch.equal(res[0][0], res[1][0]), ("forward", float((res[0][0] - res[1][0]).abs().max()))
            assert torch.equal(res[0][1], res[1][1]), ("backward", float((res[0][1] - res[1][1]).abs().max()))
        else:                            # the generic bf16-row kernel is the wave-per-row one (another association order)
            assert close(res[0][0], res[1][0], atol=1e-5, rtol=1e-5) and close(res[0][1].float(), res[1][1].float(), atol=2e-2, rtol=2e-2)
        # the generic path sums the bias gradient with torch (another association order)
        assert close(res[0][2], res[1][2], atol=1e-5 * (float(res[1][2].abs().max()) + 1e-12) + 1e-6, rtol=1e-5)
    ref = go.propagate_add(x0.float().cpu(), ei.cpu(), go.gcn_norm(ei.cpu(), None, n)) + b0.cpu()
    assert close(res[0][0], ref)


def test_band_detection_only_accepts_the_reference_pattern():
    from pangnn_amd import construct
    from pangnn_amd.graph import EdgeStructure
    n = 50
    ei = construct.neighbour_edges(n, 1, device=dev())
    assert EdgeStructure(ei, n).band_width() == 1
    assert EdgeStructure(construct.neighbour_edges(n, 3, device=dev()), n).band_width() == 3
    perm = torch.randperm(ei.shape[1], device=dev())
    assert EdgeStructure(ei[:, perm].contiguous(), n).band_width() == 0          # same edges, another order
    other = ei.clone(); other[0, 5] = (other[0, 5] + 7) % n
    assert EdgeStructure(other, n).band_width() == 0                              # same count, another edge
    assert EdgeStructure(ei[:, :-1].contiguous(), n).band_width() == 0
    # a sub-graph's neighbour graph (helper.py:366-417: no self loops, local ids) is not a band
    sg = sub_graphs_from_golden("cfg1_2genomes", 3)[0]
    assert EdgeStructure(sg.neighbour_edge_index.to(dev()), sg.x.shape[0]).band_width() == 0
    # the whole-graph fixture the reference built IS one
    g = whole_graph_from_golden("cfg1_2genomes")
    assert EdgeStructure(g.neighbour_edge_index.to(dev()), g.x.shape[0]).band_width() == 1


# ---------------------------------------------------------------- whole model on the golden graphs
def _pair(name_or_graph, dims, flags, seed=0, categorical=False):
    import pangnn_amd
    g = whole_graph_from_golden(name_or_graph) if isinstance(name_or_graph, str) else name_or_graph
    if flags.get("union_edge_weights"):
        g.edge_attr = g.union_edge_attr       # dataset.py:380: Data(x, ei, union_edge_weights, y)
    torch.manual_seed(seed)
    n = g.x.shape[0]
    oracle = go.AlternateGCNOracle(dims=dims, flags=go.default_flags(**flags), categorical_nodes=categorical,
                                   num_nodes=n)
    with torch.no_grad():
        for k, p in oracle.named_parameters():
            if k.endswith("bias"):
                p.uniform_(-0.5, 0.5)         # PyG initialises conv biases to 0; make them count
    model = pangnn_amd.AlternateGCN(dev(), None, categorical, dims=list(dims), num_nodes=n, **flags)
    model.load_state_dict(oracle.state_dict())      # identical key names: files interchange
    if categorical:
        g.x = torch.arange(n)
    return g, copy_graph(g, dev()), oracle, model


FLAG_SETS = [dict(), dict(skip_connections=True), dict(base_model=True), dict(union_edge_weights=True),
             dict(union_edge_weights=True, neighbours=4), dict(decoder="cosine"), dict(decoder="dot")]


# Tensors of the golden-graph model tests on which an activation-boundary flip on the HIP side has actually been
# OBSERVED (gpurun_out/r04a/grad_fp64.jsonl, the distances this function logs): only these keep the flip band
# max(4 e_o32 + 2e-5, 5e-4); every other (graph, flags, tensor) is held to the direct fp64 bound.
_ALL12 = ("conv_hidden.bias", "conv_hidden.lin.weight", "conv_in.bias", "conv_in.lin.weight", "conv_out.bias",
          "conv_out.lin.weight", "embedding.bias", "embedding.weight", "mlp.0.bias", "mlp.0.weight", "mlp.2.bias", "mlp.2.weight")
OBSERVED_FLIPS = {              # (graph, flags id, parameter): 3 of the 28 cases; the other 25 measure <= 2.7e-6 on every tensor
    # HIP side flips alone (fma(w, c, p + q) vs (p + q) + w c at a relu boundary): HIP 4e-5 .. 1.7e-4, fp32 oracle <= 7e-7
    **{("sim_200x4", "skip_connections=True", k): "hip" for k in (
        "conv_in.bias", "conv_in.lin.weight", "conv_out.bias", "conv_out.lin.weight", "embedding.bias", "embedding.weight",
        "mlp.0.bias", "mlp.2.bias", "mlp.2.weight")},
    # the fp64 referee sits on the other side of a boundary from BOTH fp32 evaluations (HIP == fp32 oracle to 3 digits):
    # 2.4e-5 .. 6.8e-5 and 1.7e-4 .. 8.4e-4 of the scale on either side
    **{("cfg2_sim_1000x5", "union_edge_weights=True-neighbours=4", k): "fp64" for k in _ALL12 if k not in (
        "conv_hidden.lin.weight", "mlp.0.weight")},
    **{("cfg3_5genomes", "union_edge_weights=True", k): "fp64" for k in _ALL12},
}
FP64_DIRECT = 2e-5              # HIP gradient vs the fp64 oracle, of the tensor's scale (4 x the 5e-6 DESIGN.md §2 reports)


def _log_grad_distances(tag, worst):
    """append the measured distances to $PANGNN_GRAD_LOG (one JSON object per line): the evidence the bounds rest on"""
    path = os.environ.get("PANGNN_GRAD_LOG")
    if path:
        import json
        with open(path, "a") as f:
            f.write(json.dumps({"case": tag, "dist": {k: [a, b] for k, (a, b) in worst.items()}}) + "\n")


def _check_logits_loss_grads_against_oracle(g, gd, oracle, model, tag=None, flips=()):
    """HIP logits within 1e-4 of the fp32 oracle (north star), loss 1e-5, parameter gradients within FP64_DIRECT of their
    scale of an fp64 run of the same oracle — the fp32 oracle's own distance to fp64 plays no part in that bound.
    `flips`: parameter names of this case on which a relu-boundary flip was observed (band instead of the direct bound)."""
    ref = oracle(g)
    out = model(gd)
    assert out.shape == ref.shape == (g.edge_index.shape[1],)
    assert close(out, ref)
    pw = torch.tensor(float((g.y == 0).sum() / g.y.sum()))
    lr = torch.nn.functional.binary_cross_entropy_with_logits(ref, g.y, pos_weight=pw)
    lo = torch.nn.functional.binary_cross_entropy_with_logits(out, gd.y, pos_weight=pw.to(dev()))
    assert close(lo, lr, atol=1e-5, rtol=1e-5)
    lr.backward()
    lo.backward()
    po = dict(oracle.named_parameters())
    # fp64 evaluation of the same oracle (referee for the gradient comparison below)
    import copy
    o64 = copy.deepcopy(oracle).double()
    o64.zero_grad()
    g64_ = copy.copy(g)
    g64_.x = g.x.double() if g.x.is_floating_point() else g.x
    g64_.edge_attr = g.edge_attr.double()
    torch.nn.functional.binary_cross_entropy_with_logits(o64(g64_), g.y.double(), pos_weight=pw.double()).backward()
    p64 = dict(o64.named_parameters())
    worst = {}
    for k, p in model.named_parameters():
        if po[k].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        # parameter gradients are fp32 sums over up to E edge terms taken in a different association order than the
        # oracle's.  Two fp32 evaluations cannot be held to 1e-4 of the tensor's scale against EACH OTHER: the fp32
        # oracle itself is up to 8e-4 away from an fp64 evaluation on these graphs (cfg3 union weights; 1.8e-4 on
        # cfg2 default) — activation-boundary flips and cancellation.  So (i) the direct bound stays at 1e-3, and
        # (ii) both are adjudicated against the fp64 oracle: where no activation flips, the HIP gradient is as close
        # to fp64 as the fp32 oracle is (x4 for the different association order, + 2e-5 of the scale); a flip on
        # either side moves a gradient by up to ~5e-4 of its scale (seen on the oracle: 8e-4, on the HIP side: 1e-4
        # for embedding.weight of sim_200x4 with skip connections, where fma(w, c, p + q) and (p + q) + w c round
        # differently), so that is the floor of the bound.
        scale = float(po[k].grad.abs().max()) + 1e-12
        g64 = p64[k].grad
        e_hip = float((p.grad.detach().cpu().double() - g64).abs().max()) / scale
        e_o32 = float((po[k].grad.double() - g64).abs().max()) / scale
        worst[k] = (e_hip, e_o32)
        if k in flips:       # a named, observed flip: as close as the fp32 oracle where nothing flips, inside the band otherwise
            assert e_hip <= max(4.0 * e_o32 + 2e-5, 5e-4), (k, e_hip, e_o32)
        else:                # the claim itself: the HIP gradient against fp64, whatever the fp32 oracle's own error is
            assert e_hip <= FP64_DIRECT, (tag, k, e_hip, e_o32)
        # the direct fp32-vs-fp32 bound only where the fp32 ORACLE is itself a sound reference for it: on the 1.5e6-edge
        # config-4-law graph its own sums (one fp32 chain per output over all edges) are up to 2e-3 of the scale from
        # fp64, further than the HIP sums, and the fp64 adjudication above is the whole statement
        if e_o32 <= 2.5e-4:
            assert close(p.grad, po[k].grad, atol=1e-3 * scale + 1e-7, rtol=1e-3), (k, e_hip, e_o32)
    print("gradient distance to the fp64 oracle, of the tensor's scale (HIP, fp32 oracle):",
          {k: (f"{a:.1e}", f"{b:.1e}") for k, (a, b) in worst.items()})
    _log_grad_distances(tag, worst)
    return out, ref, worst


@pytest.mark.parametrize("name,dims", [("sim_200x4", (64, 128)), ("cfg1_2genomes", (64, 128)),
                                        ("cfg2_sim_1000x5", (64, 64)), ("cfg3_5genomes", (64, 128))])
@pytest.mark.parametrize("flags", FLAG_SETS, ids=lambda f: "-".join(f"{k}={v}" for k, v in f.items()) or "default")
def test_alternate_gcn_logits_and_grads_match_oracle(name, dims, flags):
    g, gd, oracle, model = _pair(name, dims, flags)
    fid = "-".join(f"{k}={v}" for k, v in flags.items()) or "default"
    _check_logits_loss_grads_against_oracle(g, gd, oracle, model, tag=f"{name}/{fid}",
                                            flips={k for (n_, f_, k) in OBSERVED_FLIPS if (n_, f_) == (name, fid)})


@pytest.mark.parametrize("flags", [dict(), dict(skip_connections=True)], ids=["default", "skip"])
def test_config4_edge_law_matches_oracle(flags):
    """BASELINE config 4's own edge law at 1/50 scale — `--simulate_dataset 1000 20 0.2 100 20` (src/simulate.py:120-190:
    m = 38 negatives per gene, negative-binomial in-degree tail >= 1 000; N = 20 000, E ~ 1.5e6) — against the oracle:
    logits 1e-4, loss 1e-5, gradients adjudicated against the fp64 oracle.  (The golden graphs stop at E = 45 k and an
    in-degree of a few hundred; the full-size run below can only check oracle-free invariants.)"""
    from pangnn_amd import simulate
    g = simulate.simulate_graph(1000, 20, 0.2, 100, 20, seed=3, device="cpu")
    n, e = g.num_nodes, g.edge_index.shape[1]
    indeg = torch.bincount(g.edge_index[1], minlength=n)
    assert n == 20000 and 1.2e6 < e < 1.8e6 and int(indeg.max()) >= 600, (n, e, int(indeg.max()))
    g, gd, oracle, model = _pair(g, (64, 128), flags)
    # no flip band here: every HIP gradient within 2e-5 of its scale of the fp64 oracle (measured 1e-7 .. 6e-6), although the
    # fp32 ORACLE's own mlp.4.weight / mlp.0.weight sums are 2e-3 / 1e-2 away from fp64 on this graph
    out, ref, _ = _check_logits_loss_grads_against_oracle(g, gd, oracle, model,
                                                          tag="cfg4law/" + ("skip" if flags else "default"))
    # the one-pass training form (what bench.py times) on the same graph: same logits, same loss
    model.zero_grad()
    pw = torch.tensor(float((g.y == 0).sum() / g.y.sum()))
    loss, logits = model.loss_and_logits(gd, gd.y, pw.to(dev()))
    assert close(logits, ref)
    assert close(loss, torch.nn.functional.binary_cross_entropy_with_logits(ref, g.y, pos_weight=pw), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("skip", [False, True])
def test_three_decoder_forms_agree(skip):
    """fused MFMA kernel == re-associated pair_add + torch MLP == literal gather-concat-Linear"""
    g, gd, oracle, model = _pair("cfg1_2genomes", (64, 128), dict(skip_connections=skip))
    ref = oracle(g)
    outs = {}
    for mode in (True, "pair_add", False):
        model.fused_decoder = mode
        model.zero_grad()
        out = model(gd)
        out.sum().backward()
        outs[mode] = (out.detach(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
        assert close(out, ref), mode
    for mode in ("pair_add", False):
        assert close(outs[True][0], outs[mode][0], atol=2e-5, rtol=1e-5)
        for k, gk in outs[True][1].items():
            scale = float(gk.abs().max()) + 1e-12
            assert close(gk, outs[mode][1][k], atol=1e-3 * scale + 1e-7, rtol=1e-3), (mode, k)


@pytest.mark.parametrize("e", [0, 1, 31, 32, 33, 1000, 70001])
@pytest.mark.parametrize("skip", [False, True])
@pytest.mark.parametrize("mode", [1, 0], ids=["bf16x3", "f32mfma"])
def test_fused_decoder_kernel_vs_torch(e, skip, mode):
    """pangnn_decoder_mlp_{fwd,bwd}_f32 against the same MLP written with torch ops on the CPU,
    ragged tile tails included (tile = 32 edges)."""
    from pangnn_amd import functional as PF
    from pangnn_amd.graph import EdgeStructure
    torch.manual_seed(e + skip)
    n, d = 97, 64
    ei, w = random_graph(n, e, seed=e, isolated=0.0)
    if e == 0:
        ei = torch.zeros(2, 0, dtype=torch.long)
    P, Q = torch.randn(n, d), torch.randn(n, d)
    W2, b2, w3, b3, cv = torch.randn(d, d) / 8, torch.randn(d), torch.randn(d), torch.randn(1), torch.randn(d)
    extra = (w / 40) if skip else None
    leaves = [t.clone().requires_grad_(True) for t in (P, Q, W2, b2, w3, b3, cv)]
    Pr, Qr, W2r, b2r, w3r, b3r, cvr = leaves
    h1 = Pr[ei[0]] + Qr[ei[1]]
    if skip:
        h1 = h1 + extra.unsqueeze(1) * cvr
    ref = torch.relu(torch.relu(h1) @ W2r.t() + b2r) @ w3r + b3r
    gl = [t.clone().to(dev()).requires_grad_(True) for t in (P, Q, W2, b2, w3, b3, cv)]
    st = EdgeStructure(ei.to(dev()), n)
    old_mode, PF.DECODER_PRECISION = PF.DECODER_PRECISION, mode      # inference kernel and backward kernel of this mode
    try:
        out = PF.decoder_mlp(gl[0], gl[1], st, extra.to(dev()) if skip else None, gl[6] if skip else None,
                             gl[2], gl[3], gl[4], gl[5])
        assert out.shape == (e,)
        assert close(out, ref)
        go_ = torch.randn(e)
        ref.backward(go_)
        out.backward(go_.to(dev()))
    finally:
        PF.DECODER_PRECISION = old_mode
    for i, name in enumerate(["P", "Q", "W2", "b2", "w3", "b3", "cvec"]):
        if name == "cvec" and not skip:
            continue
        rg = leaves[i].grad if leaves[i].grad is not None else torch.zeros_like(leaves[i])
        scale = float(rg.abs().max()) + 1e-12
        assert close(gl[i].grad, rg, atol=1e-4 * scale + 1e-6, rtol=1e-3), name


def test_fused_decoder_bitwise_reproducible():
    from pangnn_amd import functional as PF
    from pangnn_amd.graph import EdgeStructure
    torch.manual_seed(0)
    n, e, d = 5000, 400000, 64
    ei, _ = random_graph(n, e, seed=1)
    st = EdgeStructure(ei.to(dev()), n)
    args = [torch.randn(n, d, device=dev()), torch.randn(n, d, device=dev())]
    par = [torch.randn(d, d, device=dev()) / 8, torch.randn(d, device=dev()), torch.randn(d, device=dev()),
           torch.randn(1, device=dev())]
    res = []
    for _ in range(2):
        leaves = [t.clone().requires_grad_(True) for t in args + par]
        out = PF.decoder_mlp(leaves[0], leaves[1], st, None, None, *leaves[2:])
        out.square().sum().backward()
        res.append([out.detach()] + [t.grad for t in leaves])
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_categorical_nodes_build_defined_semantics():
    g, gd, oracle, model = _pair("sim_200x4", (64, 128), dict(), categorical=True)
    assert close(model(gd), oracle(g))


@pytest.mark.parametrize("name", ["cfg1_2genomes", "cfg3_5genomes", "cfg2_sim_1000x5"])
def test_minibatch_of_32_subgraphs_matches_oracle(name):
    """the reference's actual training regime: DataLoader(batch_size=32) over per-group sub-graphs"""
    import pangnn_amd
    from pangnn_amd.data import Batch, Data
    subs = sub_graphs_from_golden(name, count=64)
    torch.manual_seed(0)
    oracle = go.AlternateGCNOracle(dims=(64, 128))
    model = pangnn_amd.AlternateGCN(dev(), None, False, dims=[64, 128])
    model.load_state_dict(oracle.state_dict())
    for lo in (0, 32):
        chunk = subs[lo:lo + 32]
        if not chunk:
            continue
        ref = oracle(go.collate(chunk))
        b = Batch.from_data_list([Data(s.x, s.edge_index, s.edge_attr, s.y,
                                       neighbour_edge_index=s.neighbour_edge_index) for s in chunk]).to(dev())
        assert close(model(b), ref)


def test_train_steps_track_the_oracle():
    from pangnn_amd.train import make_optimizer, train_step
    g, gd, oracle, model = _pair("cfg2_sim_1000x5", (64, 64), dict())
    pw = torch.tensor(float((g.y == 0).sum() / g.y.sum()))
    opt_o = torch.optim.Adam(oracle.parameters(), lr=1e-3)
    opt_m = make_optimizer(model)
    for step in range(5):
        lo, _ = go.train_step(oracle, opt_o, g, g.y, pw)
        lm, out = train_step(model, opt_m, gd, gd.y, pw.to(dev()))
        assert close(lm, lo, atol=1e-4, rtol=1e-4), step
    assert close(model(gd), oracle(g), atol=5e-4, rtol=5e-4)   # 5 Adam steps of drift allowed


# ---------------------------------------------------------------- EdgeConv / MessagePassing API
def test_edge_conv_max_aggregation_matches_oracle():
    import pangnn_amd
    torch.manual_seed(3)
    n, e, c, o = 200, 1500, 8, 12
    ei, _ = random_graph(n, e, seed=9)
    ref_m = go.EdgeConvOracle(c, o)
    m = pangnn_amd.EdgeConv(c, o).to(dev())
    m.load_state_dict(ref_m.state_dict())
    x = torch.randn(n, c)
    xr = x.clone().requires_grad_(True)
    xg = x.clone().to(dev()).requires_grad_(True)
    ref, out = ref_m(xr, ei), m(xg, ei.to(dev()))
    assert close(out, ref)
    gsel = torch.randn(n, o)
    ref.backward(gsel)
    out.backward(gsel.to(dev()))
    assert close(xg.grad, xr.grad)
    for (k, p), (_, q) in zip(m.named_parameters(), ref_m.named_parameters()):
        assert close(p.grad, q.grad, atol=1e-4, rtol=1e-3), k


def test_generic_message_passing_add():
    import pangnn_amd

    class WeightedSum(pangnn_amd.MessagePassing):
        def __init__(self):
            super().__init__(aggr="add")

        def forward(self, x, edge_index, w):
            return self.propagate(edge_index, x=x, w=w)

        def message(self, x_j, w):
            return w.view(-1, 1) * x_j

    n, e = 100, 900
    ei, w = random_graph(n, e, seed=2)
    x = torch.randn(n, 16)
    out = WeightedSum()(x.to(dev()), ei.to(dev()), w.to(dev()))
    ref = torch.zeros(n, 16).index_add_(0, ei[1], w.view(-1, 1) * x[ei[0]])
    assert close(out, ref, atol=1e-3, rtol=1e-5)


# ---------------------------------------------------------------- decoder gathers, direct C-ABI use
def test_edge_gather_concat_exact():
    from pangnn_amd import functional as PF
    from pangnn_amd.graph import EdgeStructure
    n, e, d = 500, 4000, 64
    ei, w = random_graph(n, e, seed=4)
    z = torch.randn(n, d)
    st = EdgeStructure(ei.to(dev()), n)
    a = PF.edge_gather_concat(z.to(dev()), st)
    assert torch.equal(a.cpu(), torch.cat([z[ei[0]], z[ei[1]]], dim=1))            # pure copies: bit-exact
    b = PF.edge_gather_concat(z.to(dev()), st, w.to(dev()))
    assert torch.equal(b.cpu(), torch.cat([z[ei[0]], z[ei[1]], w.unsqueeze(1)], dim=1))


# ---------------------------------------------------------------- partitioned shards on the HIP back end
@pytest.mark.parametrize("world", [2, 3])
def test_hip_ops_on_destination_shards_reassemble_the_whole_graph(world):
    """Every rank's arithmetic (HipOps on a rectangular shard: local targets, global sources) run one
    after the other on this single GPU, with the all-gathers replaced by the tensors they would
    return; the concatenated result must equal the whole-graph HIP result and the oracle."""
    import pangnn_amd
    from pangnn_amd import dist as pdist
    g = whole_graph_from_golden("cfg2_sim_1000x5")
    gd = copy_graph(g, dev())
    n, h, d = g.x.shape[0], 64, 64
    torch.manual_seed(0)
    conv = pangnn_amd.GCNConv(d, h).to(dev())
    with torch.no_grad():
        conv.bias.uniform_(-1, 1)
    x = torch.randn(n, d, device=dev())
    whole = conv(x, gd.edge_index, gd.edge_attr)
    ops = pdist.HipOps()
    n_local = (n + world - 1) // world
    n_pad = n_local * world
    shards = [pdist.partition_graph(gd, r, world) for r in range(world)]
    # stage 1 (per rank): local degree -> what the all-gather of deg^-1/2 would return
    sts = [ops.structure(s.edge_index, s.n_local, s.n_pad) for s in shards]
    from pangnn_amd import _lib
    lib = _lib.load()
    dis_parts = []
    for s, st in zip(shards, sts):
        dl = torch.empty(s.n_local, device=dev())
        c = st.by_dst
        _lib.check(lib.pangnn_gcn_degree_f32(c.rowptr.data_ptr(), c.perm.data_ptr(), s.edge_attr.data_ptr(),
                                             s.n_local, dl.data_ptr(), _lib.stream_ptr()))
        dis_parts.append(dl)
    dis_full = torch.cat(dis_parts)
    xw = torch.zeros(n_pad, h, device=dev())
    xw[:n] = conv.lin(x)
    outs = []
    for s, st in zip(shards, sts):
        nrm = ops.norm(st, s.edge_attr, lambda dloc: dis_full)
        outs.append(ops.propagate(xw, conv.bias, st, nrm)[: s.hi - s.lo])
    assert close(torch.cat(outs), whole, atol=1e-5, rtol=1e-5)
    assert close(torch.cat(outs), go.gcn_conv(x.cpu(), g.edge_index, g.edge_attr, conv.lin.weight.detach().cpu(),
                                              conv.bias.detach().cpu()))
    # decoder on shards: p indexed by global source, q by local target
    P, Q = torch.randn(n_pad, 64, device=dev()), torch.randn(n_pad, 64, device=dev())
    W2, b2, w3, b3 = (torch.randn(64, 64, device=dev()) / 8, torch.randn(64, device=dev()),
                      torch.randn(64, device=dev()), torch.randn(1, device=dev()))
    st_whole = pangnn_amd.EdgeStructure(gd.edge_index, n)
    from pangnn_amd import functional as PF
    whole_logits = PF.decoder_mlp(P[:n], Q[:n], st_whole, None, None, W2, b2, w3, b3)
    full = torch.empty_like(whole_logits)
    for s, st in zip(shards, sts):
        full[s.owned_mask] = ops.decoder(P, Q[s.lo:s.lo + s.n_local].contiguous(), st, None, None, W2, b2, w3, b3)
    assert torch.equal(full, whole_logits)          # same per-edge arithmetic, only ids remapped


# ---------------------------------------------------------------- node-level linear kernels
@pytest.mark.parametrize("k,m", [(64, 64), (64, 128), (128, 64), (128, 128)])
@pytest.mark.parametrize("n", [0, 1, 33, 1000, 40007])
def test_linear_kernels_match_torch(k, m, n):
    from pangnn_amd import functional as PF
    torch.manual_seed(n + k + m)
    x, w, b = torch.randn(n, k), torch.randn(m, k) / 8, torch.randn(m)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    xg, wg, bg = (t.clone().to(dev()).requires_grad_(True) for t in (x, w, b))
    ref = torch.nn.functional.linear(xr.double(), wr.double(), br.double())
    out = PF.linear(xg, wg, bg)
    assert out.shape == (n, m)
    assert close(out, ref, atol=1e-5, rtol=1e-5)
    g = torch.randn(n, m)
    ref.backward(g.double())
    out.backward(g.to(dev()))
    for a, r in ((xg, xr), (wg, wr), (bg, br)):
        scale = float(r.grad.abs().max()) + 1e-12 if n else 1.0
        assert close(a.grad, r.grad, atol=2e-5 * scale + 1e-7, rtol=1e-4)


@pytest.mark.parametrize("n", [1, 33, 40007])
@pytest.mark.parametrize("in_act", [0, 1])
@pytest.mark.parametrize("bf16", [False, True], ids=["f32", "bf16"])
def test_linear_wgrad_128x128_is_the_two_column_halves(n, in_act, bf16):
    """pangnn_linear_act_wgrad_mixed(K = M = 128) (conv_hidden of --union_edge_weights, gnn.py:128-139): rows [0, 64) and
    [64, 128) of dL/dW come from the 64-column windows of g through the <128, 64> kernel — bit for bit what that kernel gives on
    contiguous copies of the windows; also inside a wider matrix (ldg = 192); and within fp32 of an fp64 evaluation."""
    from pangnn_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(n + in_act)
    dt = torch.bfloat16 if bf16 else torch.float32
    wide = torch.randn(n, 192, generator=gen).to(dev()).to(dt)
    g = wide[:, 64:]                                           # [n, 128] window, ld = 192
    x = torch.randn(n, 128, generator=gen).to(dev()).to(dt)
    ws_bytes = lib.pangnn_linear_wgrad_workspace_bytes(128, 128)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev())

    def wgrad(gm, m):
        gw = torch.full((m, 128), float("nan"), device=dev())
        gb = torch.full((m,), float("nan"), device=dev())
        with torch.cuda.device(dev()):
            _lib.check(lib.pangnn_linear_act_wgrad_mixed(gm.data_ptr(), int(bf16), gm.stride(0), x.data_ptr(), int(bf16),
                                                         x.stride(0), n, 128, m, in_act, gw.data_ptr(), gb.data_ptr(),
                                                         ws.data_ptr(), ws_bytes, _lib.stream_ptr()), "linear_wgrad")
        return gw, gb

    gw, gb = wgrad(g, 128)
    lo, hi = wgrad(g[:, :64].contiguous(), 64), wgrad(g[:, 64:].contiguous(), 64)
    assert torch.equal(gw, torch.cat([lo[0], hi[0]])) and torch.equal(gb, torch.cat([lo[1], hi[1]]))
    xd = x.double()
    xd = torch.nn.functional.elu(xd) if in_act else xd
    ref = g.double().t() @ xd
    scale = float(ref.abs().max()) + 1e-12
    assert close(gw, ref, atol=2e-5 * scale, rtol=1e-4)
    assert close(gb, g.double().sum(0), atol=2e-5 * (float(g.double().sum(0).abs().max()) + 1e-12), rtol=1e-4)


def test_linear_strided_input_window():
    from pangnn_amd import functional as PF
    big = torch.randn(500, 192, device=dev())
    w = torch.randn(128, 64, device=dev())
    out = PF.linear(big[:, 64:128], w)                  # column window: ld = 192, no copy
    assert close(out, big[:, 64:128].cpu().double() @ w.cpu().double().t(), atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize("n", [0, 1, 257, 100003])
@pytest.mark.parametrize("pw", [None, 4.7])
def test_fused_bce_with_logits(n, pw):
    from pangnn_amd import functional as PF
    torch.manual_seed(n)
    x = torch.randn(n) * 6
    y = (torch.rand(n) < 0.2).float()
    pwt = None if pw is None else torch.tensor(pw)
    xr = x.clone().double().requires_grad_(True)
    if n:
        ref = torch.nn.functional.binary_cross_entropy_with_logits(xr, y.double(),
                                                                   pos_weight=None if pw is None else pwt.double())
        ref.backward()
    xg = x.clone().to(dev()).requires_grad_(True)
    out = PF.bce_with_logits(xg, y.to(dev()), None if pw is None else pwt.to(dev()), denom=max(n, 1))
    (out * 2.0).backward()
    if n:
        assert close(out, ref, atol=1e-6, rtol=1e-5)
        assert close(xg.grad, 2.0 * xr.grad, atol=1e-9, rtol=1e-4)
    else:
        assert float(out) == 0.0


def test_halo_gradient_accumulation_is_a_fixed_order_segment_sum():
    """HipOps.make_back_csr / accumulate_back: g_local[send_idx[k]] += back[k] without atomics"""
    from types import SimpleNamespace
    from pangnn_amd import dist as pdist
    ops = pdist.HipOps()
    torch.manual_seed(0)
    n_local, m, f = 1000, 7000, 64
    send_idx = torch.randint(0, n_local, (m,))
    back = torch.randn(m, f)
    g0 = torch.randn(n_local, f)
    ref = g0.clone().index_add_(0, send_idx, back)
    outs = []
    for _ in range(2):
        g = g0.clone().to(dev())
        plan = SimpleNamespace(send_idx=send_idx.to(dev()), n_local=n_local,
                               back_csr=ops.make_back_csr(send_idx.to(dev()), n_local))
        ops.accumulate_back(g, back.to(dev()), plan)
        outs.append(g)
    assert close(outs[0], ref, atol=1e-5, rtol=1e-5) and torch.equal(outs[0], outs[1])


def test_model_on_native_subgraph_batches_matches_oracle():
    """sub-graphs built by pangnn_amd/subgraphs.py on the GPU, batched as slices, through the HIP model;
    the oracle sees the same batch on the CPU"""
    import pangnn_amd
    from pangnn_amd import simulate
    ds = simulate.simulate_subgraph_dataset(300, 4, 0.3, 10, 2, seed=5, device=dev())
    assert len(ds) > 200
    torch.manual_seed(0)
    oracle = go.AlternateGCNOracle(dims=(64, 128))
    model = pangnn_amd.AlternateGCN(dev(), None, False, dims=[64, 128])
    model.load_state_dict(oracle.state_dict())
    for i0 in (0, 32, len(ds) - 7):
        b = ds.batch(i0, i0 + 32)
        ref = oracle(copy_graph(b, "cpu"))
        assert close(model(b), ref)
    # structure caches live on each batch object: a second pass over the same batch reuses them
    b = ds.batch(0, 32)
    a1 = model(b)
    assert hasattr(b, "_pangnn_structs") and torch.equal(a1, model(b))


def test_hip_graph_replay_equals_eager_steps():
    """GraphedTrainStep: the captured train step replayed N times == N eager steps (same kernels, same
    order => bitwise), for two different batches sharing one model / optimizer"""
    import pangnn_amd
    from pangnn_amd import simulate
    from pangnn_amd.train import GraphedTrainStep, make_optimizer, train_step
    ds = simulate.simulate_subgraph_dataset(300, 4, 0.3, 10, 2, seed=5, device=dev())
    batches = [ds.batch(0, 32), ds.batch(32, 64)]
    pw = ds.class_balance()

    def run_graphed():
        torch.manual_seed(0)
        model = pangnn_amd.AlternateGCN(dev(), None, False, dims=[64, 128])
        opt = make_optimizer(model, capturable=True)
        fns = [GraphedTrainStep(model, opt, b, b.y, pw, warmup=1) for b in batches]
        losses = []
        for k in range(6):
            l, _ = fns[k % 2]()
            losses.append(float(l))
        return losses, [p.detach().clone() for p in model.parameters()]

    # capture itself does not run the kernels, so eager needs exactly `warmup` extra steps per batch
    def run_eager():
        torch.manual_seed(0)
        model = pangnn_amd.AlternateGCN(dev(), None, False, dims=[64, 128])
        opt = make_optimizer(model, capturable=True)
        for b in batches:
            train_step(model, opt, b, b.y, pw)
        losses = []
        for k in range(6):
            b = batches[k % 2]
            l, _ = train_step(model, opt, b, b.y, pw)
            losses.append(float(l))
        return losses, [p.detach().clone() for p in model.parameters()]

    lg, pg = run_graphed()
    le, pe = run_eager()
    assert lg == le
    for a, b in zip(pg, pe):
        assert torch.equal(a, b)


def test_propagate_with_bf16_output_rounds_once_and_its_backward_gathers_bf16_rows():
    """config 5's propagate-first GCNConv under bf16 autocast: the propagate's result is what an autocast Linear consumes,
    i.e. bfloat16 — `out_dtype=torch.bfloat16` stores it that way (one rounding of the fp32 sums) and the transposed propagate gathers the
    bfloat16 gradient rows as stored: the same sums as gathering their fp32 copies (what the reference's cast backward hands
    on), half the bytes"""
    from pangnn_amd import functional as PF
    from pangnn_amd.graph import EdgeStructure
    n, e = 3000, 90000
    ei, w = random_graph(n, e, seed=21, hub=700)
    st = EdgeStructure(ei.to(dev()), n)
    norm = st.gcn_norm(w.to(dev()))
    torch.manual_seed(4)
    x0 = torch.randn(n, 64, device=dev()).bfloat16()
    g0 = torch.randn(n, 64, device=dev()).bfloat16()
    res = {}
    for out_bf16 in (False, True):
        x = x0.clone().requires_grad_(True)
        y = PF.propagate_any(x, None, st, norm, False, out_dtype=torch.bfloat16 if out_bf16 else None)
        y.backward(g0 if out_bf16 else g0.float())
        res[out_bf16] = (y.detach(), x.grad)
    assert res[True][0].dtype == torch.bfloat16 and res[False][0].dtype == torch.float32
    assert torch.equal(res[True][0], res[False][0].to(torch.bfloat16))                  # one rounding of the same sums
    assert res[True][1].dtype == res[False][1].dtype == torch.bfloat16
    # gathering bf16 rows == gathering their exact fp32 copies: the same products; the two kernels may associate differently
    assert close(res[True][1].float(), res[False][1].float(), atol=1e-2, rtol=1e-2)
    ref = go.propagate_add(g0.float().cpu(), ei.flip(0), go.gcn_norm(ei, w, n))        # transposed: swap the endpoints
    assert close(res[True][1].float(), ref, atol=2e-2 * float(ref.abs().max()), rtol=2e-2)


def _padded_reference_batch(ds, ids):
    """the disjoint union of the sub-graphs `ids` built with torch index ops on the host (what Batch.from_data_list does)"""
    h = ds._host()
    ei, nb, w, y, n = [], [], [], [], 0
    for i in ids:
        n0, e0, e1, b0, b1 = h.node[i], h.edge[i], h.edge[i + 1], h.nb[i], h.nb[i + 1]
        ei.append(ds.edge_index[:, e0:e1] - n0 + n)
        nb.append(ds.neighbour_edge_index[:, b0:b1] - n0 + n)
        w.append(ds.edge_attr[e0:e1]); y.append(ds.y[e0:e1])
        n += h.node[i + 1] - n0
    return torch.cat(ei, 1), torch.cat(nb, 1), torch.cat(w), torch.cat(y), n


def test_padded_collation_is_the_disjoint_union_plus_an_inert_tail():
    """pangnn_collate_subgraphs_padded on a SHUFFLED id list: the real part equals the index-op union entry for entry,
    the tail is self loops of the last (never real) node with weight 1 / label 0, live counts are the real counts"""
    from pangnn_amd import functional as PF
    from pangnn_amd import simulate
    ds = simulate.simulate_subgraph_dataset(300, 4, 0.3, 10, 2, seed=5, device=dev())
    spec = ds.padded_spec(32)
    buf = ds.padded_buffers(spec)
    gen = torch.Generator().manual_seed(1)
    for count in (32, 7, 1):
        ids = torch.randperm(ds.num_graphs, generator=gen)[:count].tolist()
        n, e, b = ds.set_graph_ids(buf, ids)
        ds.collate_padded(buf)
        ei, nb, w, y, n_ref = _padded_reference_batch(ds, ids)
        assert (n, e, b) == (n_ref, ei.shape[1], nb.shape[1])
        assert buf.live.tolist()[:5] == [e, b, n, count, 0]
        assert torch.equal(buf.edge_index[:, :e], ei) and torch.equal(buf.neighbour_edge_index[:, :b], nb)
        assert torch.equal(buf.edge_attr[:e], w) and torch.equal(buf.y[:e], y)
        for pads in (buf.edge_index[:, e:], buf.neighbour_edge_index[:, b:]):
            # self loops of the padded nodes [n, N_max), ids non-decreasing, spread evenly (no long padded row)
            assert torch.equal(pads[0], pads[1]) and int(pads.min()) >= n and int(pads.max()) < spec[1]
            assert bool((pads[0][1:] >= pads[0][:-1]).all())
            assert int(torch.bincount(pads[0] - n).max()) <= -(-pads.shape[1] // (spec[1] - n)) + 1
        assert bool((buf.edge_attr[e:] == 1).all()) and bool((buf.y[e:] == 0).all()) and bool((buf.x == 1).all())
        assert n < spec[1] and int(buf.edge_index[:, :e].max()) < n
        ptr = buf.ptr.tolist()
        assert ptr[0] == 0 and ptr[count] == n and all(v == n for v in ptr[count:])
        bid = buf.batch[:n]
        assert torch.equal(bid, torch.searchsorted(buf.ptr[1:count + 1].contiguous(), torch.arange(n, device=dev()), right=True))
        assert bool((buf.batch[n:] == count).all())
        # both CSR orders and the decoder's run-sum plans, written by the collation WITHOUT a sort (per-sub-graph ranks of the
        # data set), are the tables the one-launch sort-based build makes of the same collated lists, entry for entry
        assert buf.orders is not None
        from pangnn_amd.graph import EdgeStructure
        for name, ei in (("sim", buf.edge_index), ("nb", buf.neighbour_edge_index)):
            st = buf._pangnn_structs[name][1]
            ref = EdgeStructure(ei.clone(), spec[1], hints=buf._pangnn_hints[name])
            for a, b_ in ((st.by_dst, ref.by_dst), (st.by_src, ref.by_src)):
                assert torch.equal(a.rowptr, b_.rowptr) and torch.equal(a.other, b_.other) and torch.equal(a.perm, b_.perm)
            if name == "sim":
                ct = PF.d16_chunk(st.num_edges)
                for pa, pb in ((st.csr_plan("dst", ct), ref.csr_plan("dst", ct)), (st.runsum_plan(ct), ref.runsum_plan(ct))):
                    assert torch.equal(pa.part_off, pb.part_off) and torch.equal(pa.part_rowptr, pb.part_rowptr)
                    assert torch.equal(pa.keys, pb.keys) and pa.n_parts_exact() == pb.n_parts_exact()
    with pytest.raises(ValueError):
        ds.set_graph_ids(ds.padded_buffers((32, 10, 10, 10)), list(range(32)))          # does not fit: refused on the host


@pytest.mark.parametrize("flags", [dict(), dict(skip_connections=True)], ids=["default", "skip"])
def test_replayed_fresh_step_serves_every_batch(flags):
    """train.ReplayedFreshStep — the reference's DataLoader(batch_size=32, shuffle=True) loop (pangnn.py:152-216) as ONE
    captured HIP graph over fixed-shape padded buffers:
      (i)  replaying it over a shuffled schedule == running the same padded step eagerly, bit for bit (loss, logits,
           parameters after every step), including a batch that overflows the everyday buffers (worst-case slot) and a
           short last batch;
      (ii) the padded step == the unpadded fresh step (train_step on the index-op union) to fp32 re-association:
           loss 1e-6, logits 1e-5, parameter gradients 1e-5 of their scale."""
    import pangnn_amd
    from types import SimpleNamespace
    from pangnn_amd import simulate
    from pangnn_amd.train import ReplayedFreshStep, make_optimizer, train_step
    ds = simulate.simulate_subgraph_dataset(300, 4, 0.3, 10, 2, seed=5, device=dev())
    pw = ds.class_balance()
    gen = torch.Generator().manual_seed(3)
    perm = torch.randperm(ds.num_graphs, generator=gen).tolist()
    schedule = [perm[i:i + 32] for i in range(0, len(perm), 32)]
    h = ds._host()
    big = sorted(range(ds.num_graphs), key=lambda i: h.edge[i] - h.edge[i + 1])[:32]      # the 32 largest sub-graphs
    schedule = schedule[:4] + [big] + schedule[4:7] + [perm[:5]]

    def make(capture):
        torch.manual_seed(0)
        model = pangnn_amd.AlternateGCN(dev(), None, False, dims=[64, 128], **flags)
        opt = make_optimizer(model, capturable=True)
        return model, opt, ReplayedFreshStep(model, opt, ds, pw, 32, capture=capture, warmup=1, slack=1.2)

    ma, oa, sa = make(True)
    mb, ob, sb = make(False)
    assert not ds.fits(sa.spec, big) and ds.fits(sa.spec_worst, big)        # the schedule does exercise the second slot
    for k, ids in enumerate(schedule):
        la, xa = sa(ids)
        lb, xb = sb(ids)
        assert xa.shape == xb.shape == (sum(h.edge[i + 1] - h.edge[i] for i in ids),)
        assert torch.equal(la, lb) and torch.equal(xa, xb), (k, float(la), float(lb))
        for p, q in zip(ma.parameters(), mb.parameters()):
            assert torch.equal(p, q), k
    assert len(sa._slots) == 2

    # (ii) against the unpadded step on the same union, from the same parameters (one step each, gradients compared)
    torch.manual_seed(0)
    mc = pangnn_amd.AlternateGCN(dev(), None, False, dims=[64, 128], **flags)
    oc = make_optimizer(mc, capturable=True)
    for ids in (schedule[0], schedule[-1]):
        mc.load_state_dict(mb.state_dict())
        ei, nb, w, y, n = _padded_reference_batch(ds, ids)
        batch = SimpleNamespace(x=torch.ones(n, 1, device=dev()), edge_index=ei.contiguous(), edge_attr=w.contiguous(),
                                y=y.contiguous(), neighbour_edge_index=nb.contiguous())
        lc, xc = train_step(mc, oc, batch, batch.y, pw)
        lb, xb = sb(ids)
        assert close(lb, lc, atol=1e-6, rtol=1e-6) and close(xb, xc, atol=1e-5, rtol=1e-5)
        for (k, p), (_, q) in zip(mb.named_parameters(), mc.named_parameters()):
            if q.grad is None:                           # a layer this topology does not use
                assert p.grad is None, k
                continue
            scale = float(q.grad.abs().max()) + 1e-12
            assert close(p.grad, q.grad, atol=1e-5 * scale + 1e-9, rtol=1e-4), k


@pytest.mark.parametrize("name", ["cfg3_5genomes", "cfg2_sim_1000x5"])
def test_reference_built_subgraphs_train_through_one_captured_graph(name):
    """The reference's own sub-graphs (golden fixtures of `generate_sub_graphs`: its node numbering, its edge order — not
    source-sorted) -> SubGraphDataset.from_data_list -> train.ReplayedFreshStep: a shuffled DataLoader(batch_size=32) epoch
    replayed from ONE captured HIP graph tracks the oracle's train steps (oracle.collate + oracle.train_step, the PyG
    restatement) on the same batches: loss per step within 1e-4, logits of every batch within 5e-4 after the epoch's drift."""
    import pangnn_amd
    from pangnn_amd.subgraphs import SubGraphDataset
    from pangnn_amd.train import ReplayedFreshStep, make_optimizer
    subs = sub_graphs_from_golden(name, count=160)
    ds = SubGraphDataset.from_data_list(subs, device=dev())
    assert not ds._host().sorted_by_src                      # the reference's set-iteration edge order
    pw = ds.class_balance()
    torch.manual_seed(0)
    oracle = go.AlternateGCNOracle(dims=(64, 128))
    model = pangnn_amd.AlternateGCN(dev(), None, False, dims=[64, 128])
    model.load_state_dict(oracle.state_dict())
    opt_o = torch.optim.Adam(oracle.parameters(), lr=1e-3)
    opt_m = make_optimizer(model, capturable=True)
    step = ReplayedFreshStep(model, opt_m, ds, pw, 32, capture=True, warmup=1)
    perm = torch.randperm(len(subs), generator=torch.Generator().manual_seed(1)).tolist()
    for k in range(0, len(perm), 32):
        ids = perm[k:k + 32]
        batch = go.collate([subs[i] for i in ids])
        lo, out_o = go.train_step(oracle, opt_o, batch, batch.y, pw.cpu())
        lm, out_m = step(ids)
        assert out_m.shape == out_o.shape
        assert close(lm, lo, atol=1e-4, rtol=1e-4), k
        assert close(out_m, out_o, atol=5e-4, rtol=5e-4), k
    assert len(step._slots) >= 1


@pytest.mark.parametrize("flags", [dict(), dict(skip_connections=True)], ids=["default", "skip"])
def test_fused_loss_pass_equals_forward_criterion_backward(flags):
    """model.loss_and_logits (one decoder pass: logits + BCE + all gradients) vs model() + criterion +
    backward (three kernels), and vs the oracle"""
    from pangnn_amd.train import criterion
    g, gd, oracle, model = _pair("cfg3_5genomes", (64, 128), flags)
    pw = torch.tensor(float((g.y == 0).sum() / g.y.sum()))
    # reference: oracle
    lo = torch.nn.functional.binary_cross_entropy_with_logits(oracle(g), g.y, pos_weight=pw)
    lo.backward()
    # unfused HIP path (deferred_logits off: forward() really launches the inference decoder, criterion the loss kernel)
    model.deferred_logits = False
    out = model(gd)
    assert type(out) is torch.Tensor
    lu = criterion(out, gd.y, pw.to(dev()))
    lu.backward()
    gu = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    model.zero_grad()
    # fused
    lf, logits = model.loss_and_logits(gd, gd.y, pw.to(dev()))
    (lf * 1.0).backward()
    assert close(logits, out, atol=1e-6, rtol=1e-6) and close(logits, oracle(g))
    assert close(lf, lu, atol=1e-6, rtol=1e-6) and close(lf, lo, atol=1e-5, rtol=1e-5)
    po = dict(oracle.named_parameters())
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        scale = float(gu[k].abs().max()) + 1e-12
        assert close(p.grad, gu[k], atol=1e-5 * scale + 1e-9, rtol=1e-4), k
        assert close(p.grad, po[k].grad, atol=1e-3 * scale + 1e-7, rtol=1e-3), k
    # upstream gradient scaling flows through the stored gradients
    model.zero_grad()
    lf2, _ = model.loss_and_logits(gd, gd.y, pw.to(dev()))
    (lf2 * 3.0).backward()
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert close(p.grad, 3.0 * gu[k], atol=3e-5 * (float(gu[k].abs().max()) + 1e-12) + 1e-9, rtol=1e-4), k


def test_full_size_config4_invariants():
    """BASELINE config 4 at full size (N = 1e6, E ~ 7.5e7): properties that need no oracle run.
      * structure: rowptr ends at E, every row's edge ids ascend (stable sort), perm is a permutation
      * gcn_norm: sum_e norm_e^2 * deg_src * deg_dst / w_e^2 == E over non-isolated endpoints (definition)
      * one-pass training decoder == forward + criterion + backward (logits, loss, parameter gradients)
      * propagate: <A x, y> == <x, A^T y>"""
    import pangnn_amd
    from pangnn_amd import functional as PF
    from pangnn_amd import simulate
    from pangnn_amd.graph import structure_of
    from pangnn_amd.train import criterion
    g = simulate.simulate_graph(50000, 20, 0.2, 100, 20, seed=0, device=dev())
    n, e = g.num_nodes, g.edge_index.shape[1]
    assert n == 1_000_000 and 7.0e7 < e < 8.0e7 and g.neighbour_edge_index.shape[1] == 3 * n - 2
    st = structure_of(g.edge_index, n, holder=g, name="sim")
    for csr in (st.by_dst, st.by_src):
        assert int(csr.rowptr[0]) == 0 and int(csr.rowptr[-1]) == e
        assert bool((csr.rowptr[1:] >= csr.rowptr[:-1]).all())
        inner = torch.ones(e, dtype=torch.bool, device=dev())
        inner[csr.rowptr[:-1][csr.rowptr[:-1] < e]] = False            # first edge of each row
        assert bool((csr.perm[1:] > csr.perm[:-1])[inner[1:]].all())    # ascending original id inside a row
        assert int(torch.bincount(csr.perm.long(), minlength=e).max()) == 1
    nrm = st.gcn_norm(g.edge_attr)
    deg = torch.zeros(n, device=dev()).index_add_(0, g.edge_index[1], g.edge_attr)
    s, d = g.edge_index
    lhs = (nrm.orig.double() ** 2 * deg[s].double() * deg[d].double() / g.edge_attr.double() ** 2)
    ok = (deg[s] > 0) & (deg[d] > 0)
    assert abs(float(lhs[ok].sum()) / float(ok.sum()) - 1.0) < 1e-5
    x, y = torch.randn(n, 64, device=dev()), torch.randn(n, 64, device=dev())
    ax = PF.propagate(x, None, st, nrm)
    aty = PF.spmm_csr(st.by_src, nrm.by_src, y, n)
    _adjoint_identity(ax, y, x, aty)
    torch.manual_seed(0)
    model = pangnn_amd.AlternateGCN(dev(), None, False, dims=[64, 128], deferred_logits=False)
    out = model(g)
    lu = criterion(out, g.y, g.class_balance)
    lu.backward()
    gu = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    model.zero_grad()
    lf, logits = model.loss_and_logits(g, g.y, g.class_balance)
    lf.backward()
    assert close(logits, out, atol=1e-6, rtol=1e-6) and close(lf, lu, atol=1e-6, rtol=1e-6)
    assert bool(torch.isfinite(logits).all())
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert close(p.grad, gu[k], atol=1e-4 * (float(gu[k].abs().max()) + 1e-12) + 1e-10, rtol=1e-3), k



def _adjoint_identity(ax, y, x, aty):
    """<A x, y> == <x, A^T y>.  Both sides are sums of ~1e8 signed terms that largely cancel, so the bound is taken against
    the sum of the terms' MAGNITUDES (what the fp32 rounding of ax / aty — a few 1e-8 per element — is proportional to), not
    against the result: a bound relative to the result holds or fails with the random draw (round 5: it failed at 3e-6 of a
    result of 457 when the test ran in another order)."""
    l, r = (ax.double() * y.double()).sum().item(), (x.double() * aty.double()).sum().item()
    scale = max((ax.double().abs() * y.double().abs()).sum().item(), (x.double().abs() * aty.double().abs()).sum().item(), 1.0)
    assert abs(l - r) <= 1e-8 * scale, (l, r, scale)

def _chunks(e, step=1 << 22):
    for a in range(0, e, step):
        yield a, min(a + step, e)


@pytest.mark.parametrize("skip,pq16", [(False, False), (True, False), (True, True)], ids=["default", "skip", "skip-bf16-tables"])
def test_full_size_config4_against_a_torch_fp64_evaluation(skip, pq16):
    """BASELINE config 4 at FULL size against an independent float64 evaluation — plain torch ops on the GPU (index_select,
    matmul, index_add_ in float64, chunks of 4 M edges), the formulas of SURVEY.md §8 a4 / a5 / a7 / a9 and nothing of this
    package:
      * gcn_norm + the star propagate and its transpose: every one of the 1e6 x 64 output rows
      * the training decoder (S + T kernels): every logit, the loss, dL/dP, dL/dQ (all 2 x 1e6 x 64 values) and the gradients
        of W2, b2, w3, b3 (+ the skip feature's column) — bounds: logits 1e-4 (north_star), loss 1e-6, gradients FP64_DIRECT of
        their scale; also with --skip_connections (config 5's decoder instances) and with the P | Q tables stored as bfloat16
        (config 5's autocast: the float64 side reads the same bfloat16 values)
    The 1/50-scale oracle tests check the same arithmetic against oracle/; this is the full-size leg."""
    from pangnn_amd import functional as PF
    from pangnn_amd import simulate
    from pangnn_amd.graph import structure_of
    g = simulate.simulate_graph(50000, 20, 0.2, 100, 20, seed=0, device=dev())
    n, e = g.num_nodes, g.edge_index.shape[1]
    src, dst = g.edge_index
    st = structure_of(g.edge_index, n, holder=g, name="sim")
    nrm = st.gcn_norm(g.edge_attr)
    # ---- gcn_norm (PyG: deg = scatter_sum(w, col); dis = deg^-1/2, inf -> 0; norm = dis[row] w dis[col]) in float64
    w64 = g.edge_attr.double()
    deg = torch.zeros(n, dtype=torch.float64, device=dev()).index_add_(0, dst, w64)
    dis = deg.pow(-0.5)
    dis[torch.isinf(dis)] = 0.0
    norm64 = dis[src] * w64 * dis[dst]
    assert close(nrm.orig, norm64, atol=1e-6 * float(norm64.abs().max()), rtol=1e-5)
    # ---- propagate and its transpose, all rows (once: it does not depend on the decoder's parameters)
    torch.manual_seed(3)
    if not skip:
        x, y = torch.randn(n, 64, device=dev()), torch.randn(n, 64, device=dev())
        ax = PF.propagate(x, None, st, nrm)
        aty = PF.spmm_csr(st.by_src, nrm.by_src, y, n)
        ax64 = torch.zeros(n, 64, dtype=torch.float64, device=dev())
        aty64 = torch.zeros(n, 64, dtype=torch.float64, device=dev())
        for a, b in _chunks(e):
            ax64.index_add_(0, dst[a:b], norm64[a:b, None] * x[src[a:b]].double())
            aty64.index_add_(0, src[a:b], norm64[a:b, None] * y[dst[a:b]].double())
        for got, ref, tag in ((ax, ax64, "A x"), (aty, aty64, "A^T y")):
            err = float((got.double() - ref).abs().max()) / float(ref.abs().max())
            print(f"[full-size fp64] {tag}: max error / scale = {err:.2e}")
            assert err <= 2e-6, tag
        del ax, aty, ax64, aty64, x, y
    # ---- training decoder: P | Q tables and weights at the scale the model produces them
    torch.manual_seed(4)
    P, Q = torch.randn(n, 64, device=dev()) * 0.5, torch.randn(n, 64, device=dev()) * 0.5
    W2, b2 = torch.randn(64, 64, device=dev()) / 8, torch.randn(64, device=dev()) * 0.1
    w3, b3 = torch.randn(64, device=dev()) / 8, torch.randn(1, device=dev()) * 0.1
    pw = g.class_balance.reshape(1).float().contiguous()
    cv = torch.randn(64, device=dev()) * 0.2 if skip else None
    ex = (g.edge_attr / 40).contiguous() if skip else None
    if pq16:
        P, Q = P.to(torch.bfloat16), Q.to(torch.bfloat16)
    loss, logits, gp, gq, g_cv, g_w2, g_b2, g_w3, g_b3 = PF._decoder_train16(P, Q, st, ex, cv, W2, b2, w3, b3, y=g.y, pw=pw,
                                                                            denom=e)
    P, Q = P.float(), Q.float()                 # bfloat16 -> float32 is exact: the values the kernel gathered
    W64, b264, w364 = W2.double(), b2.double(), w3.double()
    gcv64 = torch.zeros(64, dtype=torch.float64, device=dev())
    pw64 = float(pw)
    loss64 = torch.zeros((), dtype=torch.float64, device=dev())
    gp64, gq64 = torch.zeros(n, 64, dtype=torch.float64, device=dev()), torch.zeros(n, 64, dtype=torch.float64, device=dev())
    gw264, gb264 = torch.zeros(64, 64, dtype=torch.float64, device=dev()), torch.zeros(64, dtype=torch.float64, device=dev())
    gw364, gb364 = torch.zeros(64, dtype=torch.float64, device=dev()), torch.zeros((), dtype=torch.float64, device=dev())
    gw2abs = torch.zeros(64, 64, dtype=torch.float64, device=dev())
    worst_logit = 0.0
    fragile_nodes = torch.zeros(n, dtype=torch.bool, device=dev())
    n_fragile_edges = 0
    for a, b in _chunks(e):
        s_, d_ = src[a:b], dst[a:b]
        h1p = (P[s_] + Q[d_]).double()            # the kernel's definition: h1 = relu(fl32(P + Q) ...); everything after in float64
        if skip:
            h1p = h1p + ex[a:b].double()[:, None] * cv.double()             # ... + w_e c (one more rounding in the kernel: fma)
        h1 = h1p.clamp_min(0)
        h2p = h1 @ W64.t() + b264
        h2 = h2p.clamp_min(0)
        # a relu whose argument is within fp32 rounding of 0 may fall on the other side in the kernel: one element of one
        # edge's dL/dh1 row then differs entirely.  Rows (nodes) that such an edge touches are checked with the loose bound.
        frag = (h2p.abs() < 2e-7).any(1)
        if skip:
            frag |= (h1p.abs() < 2e-7).any(1)
        fragile_nodes[s_[frag]] = True
        fragile_nodes[d_[frag]] = True
        n_fragile_edges += int(frag.sum())
        lg = h2 @ w364 + float(b3)
        worst_logit = max(worst_logit, float((logits[a:b].double() - lg).abs().max()))
        yy = g.y[a:b].double()
        # BCEWithLogitsLoss(pos_weight), mean over E: l = (1 - y) x + (1 + (pw - 1) y) softplus(-x)
        lw = 1.0 + (pw64 - 1.0) * yy
        loss64 += ((1.0 - yy) * lg + lw * torch.nn.functional.softplus(-lg)).sum()
        ge = ((1.0 - yy) - lw * torch.sigmoid(-lg)) / e                       # dL/dlogit
        gb364 += ge.sum()
        gw364 += (ge[:, None] * h2).sum(0)
        dh2 = ge[:, None] * w364 * (h2p > 0)
        gb264 += dh2.sum(0)
        gw264 += dh2.t() @ h1
        gw2abs += dh2.abs().t() @ h1                  # sum of |terms|: what the fp32 partial sums are made of
        dh1 = (dh2 @ W64) * (h1p > 0)
        gp64.index_add_(0, s_, dh1)
        gq64.index_add_(0, d_, dh1)
        if skip:
            gcv64 += (ex[a:b].double()[:, None] * dh1).sum(0)
    loss64 /= e
    print(f"[full-size fp64] decoder: max |logit error| = {worst_logit:.2e}, loss {float(loss):.8f} vs {float(loss64):.8f}")
    assert worst_logit <= 1e-4
    assert abs(float(loss) - float(loss64)) <= 1e-6 * max(1.0, abs(float(loss64)))
    n_frag = int(fragile_nodes.sum())
    print(f"[full-size fp64] {n_fragile_edges} edges with a pre-activation within 2e-7 of 0 touch {n_frag} of {n} nodes")
    assert n_frag <= n // 100
    for got, ref, tag in ((gp, gp64, "dL/dP"), (gq, gq64, "dL/dQ")):
        scale = float(ref.abs().max())
        d = (got.double() - ref).abs().amax(1) / scale
        err, err_frag = float(d[~fragile_nodes].max()), float(d[fragile_nodes].max()) if n_frag else 0.0
        print(f"[full-size fp64] {tag}: max error / scale = {err:.2e} ({err_frag:.2e} on the rows a fragile edge touches)")
        assert err <= FP64_DIRECT and err_frag <= 5e-2, tag
    # dL/dW2 is a cancelling sum (the two classes pull in opposite directions): its error against the sum of |terms|
    cond = float(gw2abs.max()) / float(gw264.abs().max())
    print(f"[full-size fp64] dL/dW2: sum |terms| / |sum| = {cond:.1f}; max error / max sum |terms| = "
          f"{float((g_w2.double() - gw264).abs().max()) / float(gw2abs.max()):.2e}")
    for got, ref, tag in ((g_w2, gw264, "dL/dW2"), (g_b2, gb264, "dL/db2"), (g_w3, gw364, "dL/dw3"),
                          (g_b3.reshape(()), gb364, "dL/db3")) + (((g_cv, gcv64, "dL/dcvec"),) if skip else ()):
        err = float((got.double() - ref).abs().max()) / (float(ref.abs().max()) + 1e-300)
        print(f"[full-size fp64] {tag}: max error / scale = {err:.2e}")
        assert err <= FP64_DIRECT, tag


class _ChunkedPropagateF64(torch.autograd.Function):
    """oracle.gcn_oracle.propagate_add (index_select -> mul -> index_add_) over 4 M-edge chunks, so that the [E, F] float64
    temporaries of a 7.5e7-edge graph never exist at once; edge weights are not differentiated"""

    @staticmethod
    def forward(ctx, x, edge_index, norm):
        ctx.save_for_backward(edge_index, norm)
        out = torch.zeros(x.shape, dtype=x.dtype, device=x.device)
        for a, b in _chunks(edge_index.shape[1]):
            out.index_add_(0, edge_index[1, a:b], norm[a:b, None] * x.index_select(0, edge_index[0, a:b]))
        return out

    @staticmethod
    def backward(ctx, g):
        edge_index, norm = ctx.saved_tensors
        gx = torch.zeros_like(g)
        for a, b in _chunks(edge_index.shape[1]):
            gx.index_add_(0, edge_index[0, a:b], norm[a:b, None] * g.index_select(0, edge_index[1, a:b]))
        return gx, None, None


@pytest.mark.parametrize("fuse_embedding", [True, False], ids=["first-layer-by-linearity", "layer-by-layer"])
def test_full_size_config4_train_step_against_the_oracle_model_in_float64(monkeypatch, fuse_embedding):
    """BASELINE config 4 at FULL size, the WHOLE train step (AlternateGCN default topology: embedding -> conv_in -> ELU ->
    conv_out -> ELU -> mlp decoder -> BCEWithLogits(pos_weight) -> backward) against oracle/gcn_oracle.py's model evaluated
    in float64 on the GPU: the oracle's own modules and gcn_norm, its propagate_add run in edge chunks (the same
    index_select / mul / index_add_), its mlp applied to 4 M edges at a time.  Every logit within 1e-4 (north_star), the loss
    within 1e-6, every parameter gradient within 1e-4 of its scale — for the default evaluation (conv_in(embedding(x)) by
    linearity, generated inside conv_out's dense kernels) and for the layer-by-layer one (`fuse_embedding=False`: the star
    propagate over 7.5e7 edges and its transpose run inside the step, as they do for any non-scalar feature).
    (Measured 5e-6 .. 2.2e-5: with pos_weight = neg / pos
    the freshly initialised model's gradient is the small difference of two large class sums — dL/db3 = sum_e dL/dlogit_e
    is 0.5 (neg - pos_weight pos) / E = 0 at sigmoid = 0.5 — so fp32 accumulation over 7.5e7 edges shows in the RESULT's
    scale; on random decoder inputs the same kernels measure 1e-7 .. 7e-6,
    test_full_size_config4_against_a_torch_fp64_evaluation.)"""
    import types
    import pangnn_amd
    from pangnn_amd import simulate
    g = simulate.simulate_graph(50000, 20, 0.2, 100, 20, seed=0, device=dev())
    n, e = g.num_nodes, g.edge_index.shape[1]
    torch.manual_seed(0)
    oracle = go.AlternateGCNOracle(dims=(64, 128), flags=go.default_flags(), num_nodes=n)
    with torch.no_grad():
        for k, p in oracle.named_parameters():
            if k.endswith("bias"):
                p.uniform_(-0.5, 0.5)         # PyG initialises conv biases to 0; make them count
    model = pangnn_amd.AlternateGCN(dev(), None, False, dims=[64, 128], num_nodes=n, fuse_embedding=fuse_embedding)
    model.load_state_dict(oracle.state_dict())
    loss, logits = model.loss_and_logits(g, g.y, g.class_balance)
    loss.backward()
    # ---- the oracle in float64 on the GPU
    oracle = oracle.double().to(dev())
    monkeypatch.setattr(go, "propagate_add", lambda x, ei, norm: _ChunkedPropagateF64.apply(x, ei, norm))
    g64 = types.SimpleNamespace(x=g.x.double(), edge_index=g.edge_index, edge_attr=g.edge_attr.double(),
                                neighbour_edge_index=g.neighbour_edge_index)
    z = oracle.encode(g64)
    zl = z.detach().requires_grad_(True)
    src, dst = g.edge_index
    pw = float(g.class_balance)
    loss64, worst = 0.0, 0.0
    for a, b in _chunks(e):
        lg = oracle.mlp(torch.cat([zl[src[a:b]], zl[dst[a:b]]], dim=1)).squeeze(-1)
        worst = max(worst, float((logits[a:b].double() - lg.detach()).abs().max()))
        yy = g.y[a:b].double()
        part = torch.nn.functional.binary_cross_entropy_with_logits(
            lg, yy, pos_weight=torch.tensor(pw, dtype=torch.float64, device=dev()), reduction="sum") / e
        part.backward()
        loss64 += float(part)
    z.backward(zl.grad)
    print(f"[full-size fp64 model, fuse_embedding={fuse_embedding}] max |logit error| = {worst:.2e}; loss {float(loss):.8f} vs {loss64:.8f}")
    assert worst <= 1e-4 and abs(float(loss) - loss64) <= 1e-6 * max(1.0, abs(loss64))
    ref = dict(oracle.named_parameters())
    for k, p in model.named_parameters():
        if ref[k].grad is None:                # conv_hidden / linear_out: not on the default topology's path
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        r = ref[k].grad
        err = float((p.grad.double() - r).abs().max()) / (float(r.abs().max()) + 1e-300)
        print(f"[full-size fp64 model] {k}: max error / scale = {err:.2e}")
        assert err <= 1e-4, k


@pytest.mark.parametrize("skip", [False, True], ids=["default", "skip"])
def test_full_size_S_and_T_kernels_agree_on_by_source_sums(skip):
    """BASELINE config 4 at full size, the two decoder kernels against EACH OTHER: dL/dP summed by source comes (i) out of
    the S kernel's own second product (per-(tile, source) run parts) and (ii) out of the T kernel's second product over
    the per-edge records S wrote, taken in by-source order.  Two kernels, two schedules, the same arithmetic in the same
    order: the [N, 64] results must be BIT-identical.  2.3e6 tiles per run — this is the check that exposes an
    intermittent per-tile miscompute (round 2's SLP build of the S kernel: dL/dh1 = v * (+0) for one edge of ~2 % of the
    half tiles in the skip-connection instances; tools/check_isa.py) that 1e4-tile unit cases miss."""
    from pangnn_amd import functional as PF
    from pangnn_amd import simulate
    from pangnn_amd.graph import structure_of
    g = simulate.simulate_graph(50000, 20, 0.2, 100, 20, seed=0, device=dev())
    n, e = g.num_nodes, g.edge_index.shape[1]
    st = structure_of(g.edge_index, n, holder=g, name="sim")
    assert st.runsum_plan() is not None                                   # canonical (source-sorted) order
    torch.manual_seed(1)
    P, Q = torch.randn(n, 64, device=dev()), torch.randn(n, 64, device=dev())
    W2, b2 = torch.randn(64, 64, device=dev()) / 8, torch.randn(64, device=dev())
    w3, b3 = torch.randn(64, device=dev()), torch.randn(1, device=dev())
    cv = torch.randn(64, device=dev()) if skip else None
    ex = (g.edge_attr / 40).contiguous() if skip else None
    pw = g.class_balance.reshape(1).float().contiguous()
    loss, logits, gp_s, gq, g_cv, g_w2, g_b2, g_w3, g_b3 = PF._decoder_train16(P, Q, st, ex, cv, W2, b2, w3, b3, y=g.y, pw=pw,
                                                                              denom=e)
    assert bool(torch.isfinite(loss).all()) and bool(torch.isfinite(gp_s).all())
    # the same sums from the records, through the T kernel over the by-source CSR (identity permutation here, but the
    # kernel gathers through it all the same) — and once more through S for run-to-run reproducibility
    rec = torch.empty(e, 8, dtype=torch.int32, device=dev())
    lib = PF._lib.load()
    plan = st.runsum_plan(PF.d16_chunk(e))
    parts = torch.empty(plan.n_parts, 64, device=dev())
    outs = [torch.empty_like(W2), torch.empty_like(w3), torch.empty_like(b3)]
    ws = torch.empty(lib.pangnn_decoder_train_workspace_bytes(), dtype=torch.uint8, device=dev())
    lg, ls = torch.empty(e, device=dev()), torch.empty(1, device=dev())
    PF._lib.check(lib.pangnn_decoder_train_mixed(
        P.data_ptr(), 64, Q.data_ptr(), 64, 0, n, st.edge_index.data_ptr(), e, e, PF._lib.ptr(ex), PF._lib.ptr(cv),
        W2.data_ptr(), b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), 64, g.y.data_ptr(), pw.data_ptr(), e, None, lg.data_ptr(),
        ls.data_ptr(), rec.data_ptr(), parts.data_ptr(), plan.part_off.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(),
        outs[2].data_ptr(), None if cv is None else torch.empty_like(cv).data_ptr(), None, ws.data_ptr(), ws.numel(),
        PF._lib.stream_ptr()), "pangnn_decoder_train_mixed")
    gp_s2 = PF._sum_parts(plan, parts, n, torch.empty(n, 64, device=dev()))
    assert torch.equal(gp_s2, gp_s) and torch.equal(lg, logits)
    gp_t = PF._dgrad_sum(rec, st, "src", W2, w3, n)
    bad = (gp_t != gp_s).any(1)
    assert not bool(bad.any()), f"{int(bad.sum())} of {n} source rows differ between the S and the T kernel"


# ---------------------------------------------------------------- edge cases of the whole module
def test_model_on_degenerate_graphs():
    """no similarity edges at all; a single node; isolated nodes; E not a multiple of the 32-edge tile"""
    import pangnn_amd
    from types import SimpleNamespace
    torch.manual_seed(0)
    oracle = go.AlternateGCNOracle(dims=(64, 128))
    model = pangnn_amd.AlternateGCN(dev(), None, False, dims=[64, 128])
    model.load_state_dict(oracle.state_dict())

    def mk(n, ei, w, nb):
        return SimpleNamespace(x=torch.ones(n, 1), edge_index=ei, edge_attr=w, neighbour_edge_index=nb)

    empty = torch.zeros(2, 0, dtype=torch.long)
    cases = [
        mk(5, empty, torch.zeros(0), torch.tensor([[0, 1, 2], [1, 2, 3]])),                  # E_sim = 0
        mk(1, empty, torch.zeros(0), torch.tensor([[0], [0]])),                              # one node, self loop
        mk(7, torch.tensor([[0, 1, 2], [1, 2, 0]]), torch.tensor([3.0, 81.0, 1.0]), empty),  # no neighbour edges
        mk(40, *random_graph(40, 33, seed=1)[:2], torch.tensor([[i for i in range(39)], [i + 1 for i in range(39)]])),
    ]
    for g in cases:
        ref = oracle(g)
        out = model(copy_graph(g, dev()))
        assert out.shape == ref.shape
        assert close(out, ref)


def test_operator_argument_errors_are_loud():
    import pangnn_amd
    conv = pangnn_amd.GCNConv(64, 64).to(dev())
    x = torch.randn(10, 64, device=dev())
    with pytest.raises(ValueError):
        conv(x, torch.tensor([[0, 1], [1, 2]], dtype=torch.int32, device=dev()))          # int32 edge_index
    with pytest.raises(ValueError):
        conv(x, torch.tensor([[0, 1], [1, 20]], device=dev()))                            # node id out of range
    with pytest.raises(ValueError):
        conv(x, torch.tensor([[0, 1], [1, 2]], device=dev()), torch.ones(5, device=dev()))  # weight length
    with pytest.raises(NotImplementedError):
        pangnn_amd.GCNConv(4, 4, add_self_loops=True)


@pytest.mark.parametrize("e", [1, 31, 32, 33, 64, 1000, 50001])
def test_decoder_run_sums_for_source_sorted_edges(e):
    """source-sorted edge list => dL/dP comes from the per-(tile, source) partial rows written by the backward
    kernel; must equal the generic segment-sum path (unsorted copy of the same edges) and torch"""
    from pangnn_amd import functional as PF
    from pangnn_amd.graph import EdgeStructure
    torch.manual_seed(e)
    n, d = 257, 64
    ei, _ = random_graph(n, e, seed=e, isolated=0.0, hub=min(e, 700))
    ei = ei[:, torch.argsort(ei[0] * n + ei[1])]                       # canonical (src, dst) order
    P, Q = torch.randn(n, d), torch.randn(n, d)
    W2, b2, w3, b3 = torch.randn(d, d) / 8, torch.randn(d), torch.randn(d), torch.randn(1)
    y = (torch.rand(e) < 0.3).float()
    pw = torch.tensor(3.0)
    st = EdgeStructure(ei.to(dev()), n)
    assert st.runsum_plan() is not None and st.runsum_plan().n_parts >= (e + 31) // 32
    perm = torch.randperm(e)
    st_u = EdgeStructure(ei[:, perm].contiguous().to(dev()), n)
    assert st_u.runsum_plan() is None or e <= 2
    res = []
    for s_, yy in ((st, y), (st_u, y[perm])):
        leaves = [t.clone().to(dev()).requires_grad_(True) for t in (P, Q, W2, b2, w3, b3)]
        loss, logits = PF.decoder_loss(leaves[0], leaves[1], s_, None, None, leaves[2], leaves[3], leaves[4], leaves[5],
                                       yy.to(dev()), pw.to(dev()), e)
        loss.backward()
        res.append((loss.detach(), [t.grad for t in leaves]))
    assert close(res[0][0], res[1][0], atol=1e-6, rtol=1e-5)
    for a, b in zip(res[0][1], res[1][1]):
        assert close(a, b, atol=1e-5 * (float(b.abs().max()) + 1e-12) + 1e-9, rtol=1e-4)
    # and against torch
    lv = [t.clone().requires_grad_(True) for t in (P, Q, W2, b2, w3, b3)]
    ref = torch.relu(torch.relu(lv[0][ei[0]] + lv[1][ei[1]]) @ lv[2].t() + lv[3]) @ lv[4] + lv[5]
    torch.nn.functional.binary_cross_entropy_with_logits(ref, y, pos_weight=pw).backward()
    assert close(res[0][1][0], lv[0].grad, atol=1e-4 * (float(lv[0].grad.abs().max()) + 1e-12) + 1e-8, rtol=1e-3)


# ---------------------------------------------------------------- bf16x3 matrix-pipe mode of the decoder backward
class _precision:
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        from pangnn_amd import functional as PF
        self.old, PF.DECODER_PRECISION = PF.DECODER_PRECISION, self.mode

    def __exit__(self, *a):
        from pangnn_amd import functional as PF
        PF.DECODER_PRECISION = self.old


@pytest.mark.parametrize("e", [1, 33, 1000, 70001, 131105, 262177, 524321, 1048609, 1048610])
@pytest.mark.parametrize("skip", [False, True])
@pytest.mark.parametrize("mode", [1, 0], ids=["bf16x3", "f32mfma"])
def test_decoder_training_kernels_vs_fp64(e, skip, mode):
    """both matrix-pipe modes of the training decoder against an fp64 torch evaluation, same bounds: precision = 1
    (bf16 terms of the fp32 operands on the matrix pipe, the default) must stay at fp32-level error.
    The edge counts above 1e5 sit just behind the steps of the S / T kernels' chunk-size function (2, 4, 8, 16 tiles per
    chunk from 4096, 8192, 16384, 32768 tiles on): runs of one source carried across the tiles of a chunk at every size,
    on 97 nodes (every run is thousands of edges long), sorted (odd E: run sums in S) and unsorted (even: both sums in T)."""
    from pangnn_amd import functional as PF
    from pangnn_amd.graph import EdgeStructure
    if e > 100000 and mode == 0:
        pytest.skip("chunked run sums are the default mode's")
    if e > 100000:
        assert PF.d16_chunk(e) == {131105: 2, 262177: 4, 524321: 8, 1048609: 16, 1048610: 16}[e]
    torch.manual_seed(e + skip)
    n, d = 97, 64
    ei, w = random_graph(n, e, seed=e, isolated=0.0)
    ei = ei[:, torch.argsort(ei[0] * n + ei[1])] if e % 2 else ei          # both the run-sum and generic paths
    P, Q = torch.randn(n, d), torch.randn(n, d)
    W2, b2, w3, b3, cv = torch.randn(d, d) / 8, torch.randn(d), torch.randn(d), torch.randn(1), torch.randn(d)
    extra = (w / 40) if skip else None
    y = (torch.rand(e) < 0.3).float()
    pw = torch.tensor(2.5)
    lv = [t.clone().double().requires_grad_(True) for t in (P, Q, W2, b2, w3, b3, cv)]
    h1 = lv[0][ei[0]] + lv[1][ei[1]]
    if skip:
        h1 = h1 + extra.double().unsqueeze(1) * lv[6]
    ref = torch.relu(torch.relu(h1) @ lv[2].t() + lv[3]) @ lv[4] + lv[5]
    lref = torch.nn.functional.binary_cross_entropy_with_logits(ref, y.double(), pos_weight=pw.double())
    lref.backward()
    st = EdgeStructure(ei.to(dev()), n)
    with _precision(mode):
        gl = [t.clone().to(dev()).requires_grad_(True) for t in (P, Q, W2, b2, w3, b3, cv)]
        loss, logits = PF.decoder_loss(gl[0], gl[1], st, extra.to(dev()) if skip else None, gl[6] if skip else None,
                                       gl[2], gl[3], gl[4], gl[5], y.to(dev()), pw.to(dev()), e)
        loss.backward()
        # the non-fused backward entry point in the same mode
        gl2 = [t.clone().to(dev()).requires_grad_(True) for t in (P, Q, W2, b2, w3, b3, cv)]
        out2 = PF.decoder_mlp(gl2[0], gl2[1], st, extra.to(dev()) if skip else None, gl2[6] if skip else None,
                              gl2[2], gl2[3], gl2[4], gl2[5])
        torch.nn.functional.binary_cross_entropy_with_logits(out2, y.to(dev()), pos_weight=pw.to(dev())).backward()
    # Bounds = what the exact three-term operand split delivers (tools/check_decoder16.py prints logits 2e-6 .. 1e-5,
    # gradients 6e-8 .. 2.5e-7 of the tensor's scale from 1 000 edges up) times 2 - 4, so that a dropped split term —
    # round 1's two-term dL/dh1 / hi+mid-only dL/dW2 sat at 2 - 5e-6 — FAILS here.  Below 1 000 edges a gradient is a
    # handful of terms whose common factor dL/dlogit carries the logit's own 2e-6 error (E = 1: every gradient 1.7e-6),
    # and b3's is a cancelling sum (E = 33: 4e-6), so those keep a looser bound.
    assert close(logits, ref, atol=1.6e-5, rtol=2e-6)                  # |logit| <= ~60 here; the gate is 1e-4
    assert close(loss, lref, atol=1e-6, rtol=1e-5)
    tol = 1e-6 if e >= 1000 else 8e-6
    for i, name in enumerate(["P", "Q", "W2", "b2", "w3", "b3", "cvec"]):
        if name == "cvec" and not skip:
            continue
        rg = lv[i].grad
        scale = float(rg.abs().max()) + 1e-30
        for got in (gl[i].grad, gl2[i].grad):
            err = float((got.detach().cpu().double() - rg).abs().max()) / scale
            assert err <= tol, (name, err, tol)


def test_bf16x3_mode_whole_model_and_f32_mode_agree():
    g, gd, oracle, model = _pair("cfg3_5genomes", (64, 128), dict())
    pw = torch.tensor(float((g.y == 0).sum() / g.y.sum()))
    res = {}
    for mode in (0, 1):
        with _precision(mode):
            model.zero_grad()
            loss, logits = model.loss_and_logits(gd, gd.y, pw.to(dev()))
            loss.backward()
            res[mode] = (loss.detach(), logits, {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    assert close(res[1][1], oracle(g))                                  # the 1e-4 logit gate
    assert close(res[1][1], res[0][1], atol=1e-5, rtol=1e-5)
    assert close(res[1][0], res[0][0], atol=1e-6, rtol=1e-6)
    for k, gk in res[0][2].items():
        scale = float(gk.abs().max()) + 1e-12
        assert close(res[1][2][k], gk, atol=1e-4 * scale + 1e-9, rtol=1e-3), k


# ---------------------------------------------------------------- embedding + first-layer propagate as one operator
@pytest.mark.parametrize("F", [16, 64, 128, 256])
@pytest.mark.parametrize("n", [0, 1, 5, 1000, 300007])
def test_weighted_colsum_kernel(F, n):
    from pangnn_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(n + F)
    g = torch.randn(n, F + 4, generator=gen)[:, 2:F + 2]          # strided rows
    r, s = torch.randn(n, generator=gen), torch.rand(n, generator=gen)
    gd, rd, sd = g.to(dev()), r.to(dev()), s.to(dev())
    outs = []
    for _ in range(2):
        out = torch.empty(2, F, device=dev())
        nb = lib.pangnn_weighted_colsum_workspace_bytes(F)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev())
        _lib.check(lib.pangnn_weighted_colsum_f32(gd.data_ptr(), gd.stride(0), rd.data_ptr(), sd.data_ptr(), n, F,
                                                  out.data_ptr(), ws.data_ptr(), nb, _lib.stream_ptr()), "colsum")
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])                           # fixed summation order
    ref = torch.stack([(r.double().unsqueeze(1) * g.double()).sum(0), (s.double().unsqueeze(1) * g.double()).sum(0)])
    scale = float(ref.abs().max()) + 1e-12
    assert close(outs[0], ref, atol=1e-4 * scale + 1e-7, rtol=1e-4)
    with pytest.raises(_lib.PangnnHipError):
        _lib.check(lib.pangnn_weighted_colsum_f32(gd.data_ptr(), gd.stride(0), rd.data_ptr(), sd.data_ptr(), n, 48,
                                                  out.data_ptr(), ws.data_ptr(), nb, _lib.stream_ptr()), "colsum")


@pytest.mark.parametrize("n,e,hub", [(1, 0, False), (7, 40, False), (900, 3500, False), (5000, 120000, True)])
@pytest.mark.parametrize("weighted", [True, False], ids=["weights", "unit"])
def test_node_actions_kernel_is_the_propagate_of_x_and_of_ones(n, e, hub, weighted):
    """pangnn_node_actions_f32: (r, s) = (A_hat x, A_hat 1) over the by-target CSR in one launch, against numpy in fp64 and
    against the generic propagate kernel on the two-column table it replaces"""
    from pangnn_amd import _lib
    from pangnn_amd import functional as PF
    from pangnn_amd.graph import structure_of
    lib = _lib.load()
    rng = np.random.default_rng(n + e)
    src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
    if hub and e:
        dst[: e // 3] = 3                                                   # one row far longer than a wave
    ei = torch.tensor(np.stack([src, dst]), dtype=torch.int64, device=dev())
    val = torch.tensor(rng.random(e) + 0.25, dtype=torch.float32, device=dev()) if weighted else None
    x = torch.tensor(rng.standard_normal(n), dtype=torch.float32, device=dev())
    st = structure_of(ei, n)
    csr = st.by_dst
    v_csr = None if val is None else val[csr.perm.long()].contiguous()
    r, s_ = torch.full((n,), float("nan"), device=dev()), torch.full((n,), float("nan"), device=dev())
    with torch.cuda.device(dev()):
        _lib.check(lib.pangnn_node_actions_f32(csr.rowptr.data_ptr(), _lib.ptr(csr.other), _lib.ptr(v_csr), x.data_ptr(), n,
                                               r.data_ptr(), s_.data_ptr(), _lib.stream_ptr()), "node_actions")
    v64 = np.ones(e) if val is None else val.double().cpu().numpy()
    r_ref, s_ref = np.zeros(n), np.zeros(n)
    np.add.at(r_ref, dst, v64 * x.double().cpu().numpy()[src])
    np.add.at(s_ref, dst, v64)
    assert close(r, torch.tensor(r_ref), atol=1e-5 * (1 + float(np.abs(r_ref).max(initial=0))), rtol=1e-5)
    assert close(s_, torch.tensor(s_ref), atol=1e-5 * (1 + float(np.abs(s_ref).max(initial=0))), rtol=1e-5)
    if e:
        tab = torch.zeros(n, 16, device=dev())
        tab[:, 0], tab[:, 1] = x, 1.0
        both = PF.spmm_csr(csr, v_csr, tab, n)
        assert close(r, both[:, 0], atol=1e-5 * (1 + float(both.abs().max())), rtol=1e-5)
        assert close(s_, both[:, 1], atol=1e-5 * (1 + float(both.abs().max())), rtol=1e-5)


@pytest.mark.parametrize("d,skip", [(64, False), (64, True), (16, True), (128, False)])
def test_pq_operands_kernel_is_the_sliced_form(d, skip):
    """pangnn_pq_operands_f32 (one launch) == cat / pad / column copy of mlp[0]'s parameters, bit for bit; gradients of
    functional.pq_operands == autograd of the sliced form"""
    from pangnn_amd import functional as PF
    gen = torch.Generator().manual_seed(d)
    w = torch.randn(d, 2 * d + int(skip), generator=gen).to(dev()).requires_grad_(True)
    b = torch.randn(d, generator=gen).to(dev()).requires_grad_(True)
    w_pq, b_pq, cvec = PF.pq_operands(w, b, d, skip)
    ref = (torch.cat([w[:, :d], w[:, d:2 * d]], dim=0), torch.cat([torch.zeros_like(b), b]), w[:, 2 * d] if skip else None)
    assert torch.equal(w_pq, ref[0]) and torch.equal(b_pq, ref[1]) and (cvec is None) == (not skip)
    if skip:
        assert torch.equal(cvec, ref[2])
    up = [torch.randn_like(t) for t in (w_pq, b_pq)] + ([torch.randn_like(cvec)] if skip else [])
    outs = [w_pq, b_pq] + ([cvec] if skip else [])
    gw, gb = torch.autograd.grad(outs, [w, b], up)
    gw_ref, gb_ref = torch.autograd.grad([t for t in ref if t is not None], [w, b], up)
    assert torch.equal(gw, gw_ref) and torch.equal(gb, gb_ref)


@pytest.mark.parametrize("F", [1, 64, 100, 128])
@pytest.mark.parametrize("n", [0, 1, 17, 900, 16384])
@pytest.mark.parametrize("bf16", [False, True], ids=["f32", "bf16"])
def test_colsum_small_kernel(F, n, bf16):
    from pangnn_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(F + n)
    g = torch.randn(n, F + 8, generator=gen).to(dev()).to(torch.bfloat16 if bf16 else torch.float32)[:, 4:F + 4]   # strided rows
    outs = []
    for _ in range(2):
        out = torch.full((F,), float("nan"), device=dev())
        with torch.cuda.device(dev()):
            _lib.check(lib.pangnn_colsum_small(g.data_ptr(), int(bf16), g.stride(0), n, F, out.data_ptr(), _lib.stream_ptr()),
                       "colsum_small")
        outs.append(out.clone())
    assert torch.equal(outs[0], outs[1])                                    # fixed order of additions
    ref = g.double().sum(0)
    atol = 1e-5 * (1 + (float(ref.abs().max()) if n else 0.0)) + 1e-6 * n ** 0.5
    assert close(outs[0], ref, atol=atol, rtol=1e-5)
    from pangnn_amd import functional as PF
    assert close(PF.colsum(g), ref, atol=atol, rtol=1e-5)


@pytest.mark.parametrize("k,m", [(64, 64), (64, 128), (128, 64), (128, 128)])
@pytest.mark.parametrize("n", [1, 33, 1000])
@pytest.mark.parametrize("gated,bf16", [(False, False), (True, False), (True, True)], ids=["plain", "gate", "gate-bf16"])
def test_linear_dgrad_from_the_layers_own_weight(k, m, n, gated, bf16):
    """pangnn_linear_dgrad_mixed (w [M][K] staged through the strides of its transpose) == the forward kernel on a
    transposed copy of w, bit for bit — the same operand images, the same products"""
    from pangnn_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(k + m + n)
    dt = torch.bfloat16 if bf16 else torch.float32
    g = torch.randn(n, m, generator=gen).to(dev()).to(dt)
    w = torch.randn(m, k, generator=gen).to(dev())
    gate = torch.randn(n, k, generator=gen).to(dev()).to(dt) if gated else None
    a = torch.full((n, k), float("nan"), dtype=dt, device=dev())
    b = torch.full((n, k), float("nan"), dtype=dt, device=dev())
    wt = w.t().contiguous()
    with torch.cuda.device(dev()):
        _lib.check(lib.pangnn_linear_dgrad_mixed(g.data_ptr(), int(bf16), g.stride(0), w.data_ptr(), a.data_ptr(), int(bf16),
                                                 a.stride(0), n, k, m, _lib.ptr(gate), int(bf16), k if gated else 0,
                                                 _lib.stream_ptr()), "linear_dgrad")
        _lib.check(lib.pangnn_linear_act_fwd_mixed(g.data_ptr(), int(bf16), g.stride(0), wt.data_ptr(), None, b.data_ptr(),
                                                   int(bf16), b.stride(0), n, m, k, 0, _lib.ptr(gate), int(bf16),
                                                   k if gated else 0, _lib.stream_ptr()), "linear_fwd")
    assert torch.equal(a, b)
    ref = g.double() @ w.double()
    if gated:
        gd = gate.double()
        ref = ref * torch.where(gd > 0, torch.ones_like(gd), gd.exp())
    assert close(a.float(), ref, atol=(0.3 if bf16 else 2e-4), rtol=(2e-2 if bf16 else 1e-4))


@pytest.mark.parametrize("F", [64, 128])
@pytest.mark.parametrize("n", [0, 1, 5, 1000, 300007])
@pytest.mark.parametrize("bf16", [False, True], ids=["f32", "bf16"])
def test_rank2_rows_and_weighted_colsum3_kernels(F, n, bf16):
    """pangnn_rank2_rows: out = r a^T + s c^T + bias (stored f32 / bf16); pangnn_weighted_colsum3: [r s 1]^T g"""
    from pangnn_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(F + n)
    r, s_ = torch.randn(n, generator=gen), torch.rand(n, generator=gen) + 0.5
    a, c, b = torch.randn(F, generator=gen), torch.randn(F, generator=gen), torch.randn(F, generator=gen)
    g = torch.randn(n, F, generator=gen)
    rd, sd, ad, cd, bd = (t.to(dev()) for t in (r, s_, a, c, b))
    dt = torch.bfloat16 if bf16 else torch.float32
    out = torch.full((n, F), float("nan"), dtype=dt, device=dev())
    with torch.cuda.device(dev()):
        _lib.check(lib.pangnn_rank2_rows(rd.data_ptr(), sd.data_ptr(), ad.data_ptr(), cd.data_ptr(), bd.data_ptr(),
                                         out.data_ptr(), int(bf16), F, n, F, _lib.stream_ptr()), "rank2_rows")
    ref = r.double()[:, None] * a.double() + s_.double()[:, None] * c.double() + b.double()
    assert close(out.float(), ref, atol=(4e-2 if bf16 else 1e-5), rtol=(8e-3 if bf16 else 1e-6))
    if bf16 and n:      # exactly the f32 result rounded once (round to nearest even)
        o32 = torch.empty(n, F, device=dev())
        with torch.cuda.device(dev()):
            _lib.check(lib.pangnn_rank2_rows(rd.data_ptr(), sd.data_ptr(), ad.data_ptr(), cd.data_ptr(), bd.data_ptr(),
                                             o32.data_ptr(), 0, F, n, F, _lib.stream_ptr()), "rank2_rows")
        assert torch.equal(out, o32.to(torch.bfloat16))
    gd = g.to(dev()).to(dt)
    sums = torch.full((3, F), float("nan"), device=dev())
    outs = []
    for _ in range(2):
        with torch.cuda.device(dev()):
            nb = lib.pangnn_weighted_colsum3_workspace_bytes(F)
            ws = torch.empty(nb, dtype=torch.uint8, device=dev())
            _lib.check(lib.pangnn_weighted_colsum3(gd.data_ptr(), int(bf16), gd.stride(0), rd.data_ptr(), sd.data_ptr(), n, F,
                                                   sums.data_ptr(), ws.data_ptr(), nb, _lib.stream_ptr()), "colsum3")
        outs.append(sums.clone())
    assert torch.equal(outs[0], outs[1])                                   # reproducible
    g64 = gd.double().cpu()
    ref3 = torch.stack([(r.double()[:, None] * g64).sum(0), (s_.double()[:, None] * g64).sum(0), g64.sum(0)])
    scale = float(ref3.abs().max()) + 1e-12
    assert close(outs[0], ref3, atol=1e-5 * scale + 1e-7, rtol=1e-4)
    with pytest.raises(_lib.PangnnHipError):
        _lib.check(lib.pangnn_rank2_rows(rd.data_ptr(), sd.data_ptr(), ad.data_ptr(), cd.data_ptr(), None, out.data_ptr(),
                                         int(bf16), 48 + 2, max(n, 1), 50, _lib.stream_ptr()), "rank2_rows")


@pytest.mark.parametrize("flags,dims", [(dict(), (64, 128)), (dict(base_model=True), (64, 128)),
                                        (dict(union_edge_weights=True), (64, 128)), (dict(), (64, 64)), (dict(), (128, 64))],
                         ids=["default", "base", "union", "default-64x64", "default-128x64"])
def test_fused_embedding_layer_equals_layerwise_form_and_oracle(flags, dims):
    """conv_in(embedding(x)) in its three forms — by linearity as r a^T + s c^T + b_in (default), round 2's per-step
    propagate of h0 with the embedding gradients by linearity ("propagate"), layer by layer (False: what the reference
    executes) — against each other and the oracle.  Non-constant scalar node feature: A_hat x and A_hat 1 differ, both
    embedding gradients are exercised; dims with D < H (propagate first) and D >= H (dense layer first)."""
    import pangnn_amd
    g, gd, oracle, model = _pair("cfg2_sim_1000x5", dims, flags, seed=5)
    gen = torch.Generator().manual_seed(9)
    g.x = torch.randn(g.x.shape[0], 1, generator=gen)
    gd.x = g.x.to(dev())
    assert model.fuse_embedding is True
    forms = {"rank2": model}
    for name, mode in (("propagate", "propagate"), ("layerwise", False)):
        forms[name] = pangnn_amd.AlternateGCN(dev(), None, False, dims=list(dims), fuse_embedding=mode, **flags)
        forms[name].load_state_dict(model.state_dict())
    pw = torch.tensor(float((g.y == 0).sum() / g.y.sum()))
    ref_logits = oracle(g)
    lr = torch.nn.functional.binary_cross_entropy_with_logits(ref_logits, g.y, pos_weight=pw)
    lr.backward()
    grads = {}
    for name, m in forms.items():
        gm = copy_graph(g, dev())
        loss, out = m.loss_and_logits(gm, gm.y, pw.to(dev()))
        loss.backward()
        assert close(out, ref_logits), name                                 # the 1e-4 logit gate, every form
        assert close(loss, lr, atol=1e-5, rtol=1e-5), name
        grads[name] = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    po = dict(oracle.named_parameters())
    for k, q in po.items():
        if q.grad is None:
            continue
        scale = float(q.grad.abs().max()) + 1e-12
        for name in forms:
            assert close(grads[name][k], q.grad, atol=1e-3 * scale + 1e-7, rtol=1e-3), (name, k)
        assert close(grads["rank2"][k], grads["layerwise"][k], atol=1e-3 * scale + 1e-7, rtol=1e-3), k
    # the cached A_hat x is keyed on the feature tensor: an in-place change must be seen
    plain = forms["layerwise"]
    gm = copy_graph(g, dev())
    model.zero_grad()
    l1, _ = model.loss_and_logits(gm, gm.y, pw.to(dev())); l1.backward()
    g1 = model.embedding.weight.grad.clone()
    gm.x.mul_(2.0)
    model.zero_grad()
    l2, _ = model.loss_and_logits(gm, gm.y, pw.to(dev())); l2.backward()
    plain.zero_grad()
    gp = copy_graph(g, dev()); gp.x = gp.x * 2.0
    l3, _ = plain.loss_and_logits(gp, gp.y, pw.to(dev())); l3.backward()
    scale = float(plain.embedding.weight.grad.abs().max()) + 1e-12
    assert close(model.embedding.weight.grad, plain.embedding.weight.grad, atol=1e-3 * scale + 1e-7, rtol=1e-3)
    assert not torch.equal(g1, model.embedding.weight.grad)
    # inference form (no autograd) through the same operator
    with torch.no_grad():
        assert close(model(copy_graph(g, dev())), ref_logits)


# ---------------------------------------------------------------- first layer generated inside the next dense layer
@pytest.mark.parametrize("h,m", [(64, 64), (64, 128), (128, 64)])
@pytest.mark.parametrize("n,bias", [(1, True), (31, False), (32, True), (33, False), (1000, True), (40007, False), (70016, True),
                                    (1000003, False)])
def test_first_layer_generated_inside_the_next_dense_layer(h, m, n, bias):
    """functional.embed_conv_in_linear (pangnn_embed_linear_fwd / _bwd: the [N, H] rows of conv_in(embedding(x)) generated
    inside the dense layer's kernels) against the two-operator form linear(embed_conv_in(...), in_act=1): the forward and
    dL/dW_out / dL/dbias_out bit for bit (same generated values, same products, same slab order), the first layer's four
    parameter gradients within fp32 re-association of the column sums — and everything against an fp64 evaluation.
    Row counts around the 32-row tile, one wave's share, and more tiles than one pass of the grid."""
    from pangnn_amd import functional as PF
    from pangnn_amd.graph import EdgeStructure
    if n > 100000 and (h, m) != (128, 64):
        pytest.skip("BASELINE's node count: the headline widths only")
    d = 64
    gen = torch.Generator().manual_seed(7 * n + h + m)
    e = max(4 * n, 8)
    ei, w = random_graph(n, e, seed=n + h, hub=min(e, 700))
    st = EdgeStructure(ei.to(dev()), n)
    norm = st.gcn_norm(w.to(dev()))
    x = torch.randn(n, 1, generator=gen).to(dev())
    P = lambda *s_: (torch.randn(*s_, generator=gen) * 0.3).to(dev()).requires_grad_(True)       # noqa: E731
    params = [P(d, 1), P(d), P(h, d), P(h), P(m, h), P(m) if bias else None]
    go_ = torch.randn(n, m, generator=gen).to(dev())

    def run(fused):
        for p_ in params:
            if p_ is not None:
                p_.grad = None
        we, be, win, bin_, wout, bout = params
        if fused:
            y = PF.embed_conv_in_linear(x, we, be, win, bin_, wout, bout, st, norm)
        else:
            y = PF.linear(PF.embed_conv_in(x, we, be, win, bin_, st, norm), wout, bout, 1)
        y.backward(go_)
        return y.detach(), [None if p_ is None else p_.grad.clone() for p_ in params]

    y1, g1 = run(True)
    y0, g0 = run(False)
    assert torch.equal(y1, y0)
    assert torch.equal(g1[4], g0[4]) and (not bias or torch.equal(g1[5], g0[5]))
    # fp64 evaluation
    r, s_ = PF._node_actions(x, st, norm)
    pd = [None if p_ is None else p_.detach().double().requires_grad_(True) for p_ in params]
    hh = r.double()[:, None] * (pd[2] @ pd[0]).T + s_.double()[:, None] * (pd[2] @ pd[1])[None, :] + pd[3]
    yd = torch.nn.functional.elu(hh) @ pd[4].T + (pd[5] if bias else 0.0)
    yd.backward(go_.double())
    assert close(y1, yd, atol=2e-5, rtol=1e-5)
    for k_ in range(6):
        if params[k_] is None:
            continue
        ref = pd[k_].grad
        scale = float(ref.abs().max()) + 1e-30
        assert close(g1[k_], ref, atol=2e-5 * scale, rtol=1e-4), k_
        assert close(g1[k_], g0[k_], atol=2e-5 * scale, rtol=1e-4), k_


@pytest.mark.parametrize("flags,dims", [(dict(), (64, 128)), (dict(base_model=True), (64, 128)), (dict(), (64, 64)),
                                        (dict(union_edge_weights=True), (64, 64)), (dict(skip_connections=True), (64, 128)),
                                        (dict(union_edge_weights=True, neighbours=4), (64, 64))],
                         ids=["default", "base", "default-64x64", "union-64x64", "skip", "union-3-hidden"])
def test_model_with_first_dense_layer_fused_equals_unfused_model(flags, dims):
    """AlternateGCN(fuse_first_dense=True) (default) against fuse_first_dense=False: loss, logits and every gradient behind
    the fused operator bit for bit; the first layer's own parameter gradients within re-association"""
    import pangnn_amd
    g, gd, oracle, model = _pair("cfg2_sim_1000x5", dims, flags, seed=2)
    gen = torch.Generator().manual_seed(4)
    gd.x = torch.randn(g.x.shape[0], 1, generator=gen).to(dev())
    assert model.fuse_first_dense is True
    plain = pangnn_amd.AlternateGCN(dev(), None, False, dims=list(dims), fuse_first_dense=False, **flags)
    plain.load_state_dict(model.state_dict())
    pw = (gd.y == 0).sum() / gd.y.sum()
    seen = []
    orig = PF_mod().embed_conv_in_linear          # the operator's entry (its implementation is the C++ op pangnn::embed_conv_in_linear)
    PF_mod().embed_conv_in_linear = lambda *a: (seen.append(1), orig(*a))[1]
    try:
        out = []
        for m_ in (model, plain):
            loss, logits = m_.loss_and_logits(copy_graph(gd, dev()), gd.y, pw)
            loss.backward()
            out.append((loss.detach(), logits, {k: p.grad for k, p in m_.named_parameters() if p.grad is not None}))
    finally:
        PF_mod().embed_conv_in_linear = orig
    assert len(seen) == 1                                   # the fused operator ran in `model`, not in `plain`
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    first = ("embedding.weight", "embedding.bias", "conv_in.lin.weight", "conv_in.bias")
    assert out[0][2].keys() == out[1][2].keys()
    for k, v in out[0][2].items():
        if k in first:
            scale = float(out[1][2][k].abs().max()) + 1e-30
            assert close(v, out[1][2][k], atol=2e-5 * scale, rtol=1e-4), k
        else:
            assert torch.equal(v, out[1][2][k]), k


def PF_mod():
    from pangnn_amd import functional
    return functional


# ---------------------------------------------------------------- dense layer with the preceding ELU folded in
@pytest.mark.parametrize("k,m", [(64, 64), (64, 128), (128, 64)])
@pytest.mark.parametrize("n", [1, 33, 1000, 40007])
def test_linear_with_folded_elu_matches_torch(k, m, n):
    from pangnn_amd import functional as PF
    gen = torch.Generator().manual_seed(n + k + m)
    x = torch.randn(n, k, generator=gen) * 2
    w = torch.randn(m, k, generator=gen) / 8
    b = torch.randn(m, generator=gen)
    gy = torch.randn(n, m, generator=gen)
    ref_in = [t.clone().double().requires_grad_(True) for t in (x, w, b)]
    ref = torch.nn.functional.linear(torch.nn.functional.elu(ref_in[0]), ref_in[1], ref_in[2])
    ref.backward(gy.double())
    leaves = [t.clone().to(dev()).requires_grad_(True) for t in (x, w, b)]
    out = PF.linear(leaves[0], leaves[1], leaves[2], in_act=1)
    out.backward(gy.to(dev()))
    assert close(out, ref)
    for a, r in zip(leaves, ref_in):
        scale = float(r.grad.abs().max()) + 1e-12
        assert close(a.grad, r.grad, atol=1e-4 * scale + 1e-7, rtol=1e-3)


def test_folded_activation_model_equals_layerwise_model():
    import pangnn_amd
    for flags in (dict(), dict(base_model=True), dict(union_edge_weights=True), dict(skip_connections=True)):
        g, gd, oracle, model = _pair("cfg2_sim_1000x5", (64, 128), flags, seed=3)
        plain = pangnn_amd.AlternateGCN(dev(), None, False, dims=[64, 128], fold_activation=False, **flags)
        plain.load_state_dict(model.state_dict())
        pw = torch.tensor(2.0, device=dev())
        res = []
        for m in (model, plain):
            gm = copy_graph(g, dev())
            loss, out = m.loss_and_logits(gm, gm.y, pw)
            loss.backward()
            res.append((loss.detach(), out, {k: p.grad for k, p in m.named_parameters() if p.grad is not None}))
            assert close(m(gm), out, atol=1e-5, rtol=1e-5)          # inference path (activation applied explicitly)
        assert close(res[0][1], oracle(g))
        assert close(res[0][0], res[1][0], atol=1e-6, rtol=1e-5)
        assert close(res[0][1], res[1][1], atol=1e-5, rtol=1e-5)
        for k, gk in res[1][2].items():
            scale = float(gk.abs().max()) + 1e-12
            assert close(res[0][2][k], gk, atol=1e-4 * scale + 1e-9, rtol=1e-3), (flags, k)


# ---------------------------------------------------------------- propagate over bfloat16-stored rows
@pytest.mark.parametrize("F", [32, 64, 128, 256])
@pytest.mark.parametrize("n,e,hub", [(1, 5, None), (513, 7000, 3000), (2000, 30000, None)])
def test_spmm_bf16_rows_match_oracle_on_rounded_inputs(F, n, e, hub):
    """bf16 storage of the gathered rows, fp32 weights / accumulation / result: equal (to fp32 rounding) to the
    oracle run on the bf16-rounded features; the gradient w.r.t. the rows comes back in bf16"""
    from pangnn_amd import functional as PF
    from pangnn_amd.graph import EdgeStructure
    ei, w = random_graph(n, e, seed=F + n, hub=hub)
    torch.manual_seed(F)
    x = torch.randn(n, F)
    xb = x.to(torch.bfloat16)
    b = torch.randn(F)
    norm_ref = go.gcn_norm(ei, w.double(), n, dtype=torch.float64)
    ref = go.propagate_add(xb.double(), ei, norm_ref) + b.double()
    st = EdgeStructure(ei.to(dev()), n)
    norm = st.gcn_norm(w.to(dev()))
    xd = xb.to(dev()).requires_grad_(True)
    out = PF.propagate(xd, b.to(dev()), st, norm)
    assert out.dtype == torch.float32
    assert close(out, ref, atol=1e-4, rtol=1e-4)
    g = torch.randn(n, F)
    out.backward(g.to(dev()))
    assert xd.grad.dtype == torch.bfloat16
    gref = go.propagate_add(g.double(), ei.flip(0), norm_ref)             # transposed propagate
    assert close(xd.grad.float(), gref, atol=2e-2 * (float(gref.abs().max()) + 1e-12), rtol=2e-2)


def test_gcnconv_under_bf16_autocast_propagates_bf16_rows():
    """`accelerate` bf16 mixed precision: the reference's GCNConv then runs its Linear in bf16 and the propagate on
    bf16 rows with fp32 accumulation; here the rows are stored in bf16 (the dense part stays fp32, i.e. more
    accurate than the reference), so the result agrees with the oracle under CPU autocast to bf16 resolution"""
    import pangnn_amd
    g = whole_graph_from_golden("cfg2_sim_1000x5")
    n = g.x.shape[0]
    torch.manual_seed(4)
    x = torch.randn(n, 128)
    conv_o = go.GCNConvOracle(128, 64)
    with torch.no_grad():
        conv_o.bias.uniform_(-0.5, 0.5)
    conv = pangnn_amd.GCNConv(128, 64).to(dev())
    conv.load_state_dict(conv_o.state_dict())
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ref = conv_o(x, g.edge_index, g.edge_attr).float()
    exact = conv_o(x, g.edge_index, g.edge_attr)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = conv(x.to(dev()), g.edge_index.to(dev()), g.edge_attr.to(dev()))
    assert out.dtype == torch.float32
    scale = float(exact.abs().max())
    assert float((out.cpu() - ref).abs().max()) < 3e-2 * scale            # bf16 resolution vs the autocast oracle
    assert float((out.cpu() - exact).abs().max()) < 1e-2 * scale          # only the stored rows are rounded
    assert float((out.cpu() - exact).abs().max()) > 1e-6 * scale          # ... and they are


def test_gcn_norm_cache_survives_address_reuse():
    """The per-structure norm cache is keyed on the weight tensor's address / version / shape.  A weight tensor
    that is freed and whose address the caching allocator hands to a NEW same-shape tensor must not return the old
    normalisation (round-1 ADVICE: the cache entry now keeps the keyed tensor alive)."""
    from pangnn_amd.graph import EdgeStructure
    n, e = 300, 5000
    ei, w = random_graph(n, e, seed=5)
    st = EdgeStructure(ei.to(dev()), n)
    seen = set()
    for k in range(6):
        wk = (w * (k + 1)).to(dev())                 # fresh tensor each round; the previous one is dropped
        seen.add(wk.data_ptr())
        got = st.gcn_norm(wk).orig.cpu()
        want = go.gcn_norm(ei, w * (k + 1), n)
        assert close(got, want, atol=1e-6, rtol=1e-5), k
        del wk
    # and through the identity-keyed structure cache with a non-contiguous edge_index
    from pangnn_amd.graph import structure_of, clear_cache
    clear_cache()
    big = torch.stack([ei[0], ei[0], ei[1]]).to(dev())
    view = big[::2]                                  # rows 0 and 2: non-contiguous [2, E]
    st2 = structure_of(view, n)
    assert torch.equal(st2.edge_index.cpu(), ei)
    clear_cache()


def test_config5_slice_skip_connections_categorical_bf16_autocast():
    """BASELINE.json config 5 on a 1-GPU-sized slice: `--skip_connections --categorical_node` under bf16 mixed
    precision.  The oracle under CPU bf16 autocast (what `accelerate` makes of the reference: its Linear layers run in
    bf16) is the reference; the product stores only the propagated rows in bf16 and keeps the dense layers and the
    decoder in fp32-level arithmetic, so it must (i) agree with the autocast oracle to bf16 resolution, (ii) be at
    least as close to the fp32 oracle as the autocast oracle is, (iii) train: three steps lower the loss like the
    oracle's."""
    _check_config5_flags_against_autocast_oracle(
        *_pair("cfg2_sim_1000x5", (64, 64), dict(skip_connections=True), categorical=True))


def _check_config5_flags_against_autocast_oracle(g, gd, oracle, model):
    pw = torch.tensor(float((g.y == 0).sum() / g.y.sum()))
    exact = oracle(g)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ref = oracle(g).float()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(gd)
        loss, logits = model.loss_and_logits(gd, gd.y, pw.to(dev()))
    assert out.dtype == torch.float32 and torch.allclose(out, logits, atol=1e-4, rtol=1e-4)    # inference form vs one-pass training form
    scale = float(exact.abs().max())
    e_ref = float((ref - exact).abs().max())
    e_out = float((out.cpu() - exact).abs().max())
    assert float((out.cpu() - ref).abs().max()) < 5e-2 * scale          # bf16 resolution
    assert e_out <= e_ref + 1e-3 * scale                                 # no further from fp32 than the reference's own autocast
    assert e_out > 1e-6 * scale                                          # the bf16 row storage is really on
    # a few training steps under autocast track the oracle's
    from pangnn_amd.train import make_optimizer, train_step
    opt_m, opt_o = make_optimizer(model), torch.optim.Adam(oracle.parameters(), lr=1e-3)
    lm, lo = [], []
    for _ in range(3):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            l_, _ = train_step(model, opt_m, gd, gd.y, pw.to(dev()))
        lm.append(float(l_))
        with torch.autocast("cpu", dtype=torch.bfloat16):
            l2, _ = go.train_step(oracle, opt_o, g, g.y, pw)
        lo.append(float(l2))
    assert lm[-1] < lm[0] and abs(lm[-1] - lo[-1]) < 5e-2 * abs(lo[0])


def test_config5_edge_law_matches_autocast_oracle():
    """BASELINE config 5's own edge law (`--simulate_dataset 200000 50 0.1 500 50`: m = floor(49 / 2 * 9) = 220 negatives
    per gene towards the next genome, src/simulate.py:120-133) on 1 000 genes x 6 genomes, with config 5's flags
    (`--skip_connections --categorical_node`, src/gnn.py:93,111,173) and hidden_dim 128 under bf16 autocast, against the
    oracle under CPU bf16 autocast — the checks of the test above on the graph law the flags are quoted with."""
    from pangnn_amd import simulate
    g = simulate.simulate_graph(1000, 6, 0.1, 500, 50, seed=5, device="cpu", mean_neg=220, adjacent_only=True)
    n, e = g.num_nodes, g.edge_index.shape[1]
    assert n == 6000 and 1.5e6 < e < 3.0e6, (n, e)                       # ~ 2 * 5 * 1000 * 220 directed negatives
    assert float(g.y.mean()) < 0.02
    _check_config5_flags_against_autocast_oracle(
        *_pair(g, (64, 128), dict(skip_connections=True), categorical=True))


def test_full_size_config5_slice_invariants():
    """One GPU's share of BASELINE config 5 at full size (`bench.py --workload cfg5slice`: 6 of the 50 genomes, 200 000
    genes each, m = 220: N = 1.2e6, E ~ 4.4e8) with `--skip_connections --categorical_node` under bf16 autocast —
    properties that need no oracle run (the pattern of test_full_size_config4_invariants):
      * structure: rowptr ends at E, ascending edge ids inside a row, perm is a permutation (both CSR orders)
      * only adjacent genomes are connected (what the 8-way partition's one-genome halo relies on)
      * propagate on bfloat16-stored rows: <A x, y> == <x, A^T y>
      * one-pass training decoder == forward + criterion + backward; everything finite"""
    import pangnn_amd
    from pangnn_amd import functional as PF
    from pangnn_amd import simulate
    from pangnn_amd.graph import structure_of
    from pangnn_amd.train import criterion
    g = simulate.simulate_graph(200000, 6, 0.1, 500, 50, seed=0, device=dev(), mean_neg=220, adjacent_only=True)
    n, e = g.num_nodes, g.edge_index.shape[1]
    assert n == 1_200_000 and 4.0e8 < e < 4.8e8 and g.neighbour_edge_index.shape[1] == 3 * n - 2
    gs, gt = g.genome_of[g.edge_index[0]], g.genome_of[g.edge_index[1]]
    assert bool(((gs - gt).abs() == 1).all())
    del gs, gt
    st = structure_of(g.edge_index, n, holder=g, name="sim")
    for csr in (st.by_dst, st.by_src):
        assert int(csr.rowptr[0]) == 0 and int(csr.rowptr[-1]) == e
        assert bool((csr.rowptr[1:] >= csr.rowptr[:-1]).all())
        inner = torch.ones(e, dtype=torch.bool, device=dev())
        inner[csr.rowptr[:-1][csr.rowptr[:-1] < e]] = False            # first edge of each row
        assert bool((csr.perm[1:] > csr.perm[:-1])[inner[1:]].all())    # ascending original id inside a row
        del inner
        seen = torch.zeros(e, dtype=torch.uint8, device=dev())
        seen[csr.perm.long()] = 1
        assert int(seen.sum(dtype=torch.int64)) == e                    # a permutation of the edge ids
        del seen
    nrm = st.gcn_norm(g.edge_attr)
    x = torch.randn(n, 64, device=dev()).bfloat16()
    y = torch.randn(n, 64, device=dev()).bfloat16()
    ax = PF.spmm_csr(st.by_dst, nrm.by_dst, x, n)                       # pangnn_spmm_csr_bf16: rows gathered as stored
    aty = PF.spmm_csr(st.by_src, nrm.by_src, y, n)
    assert ax.dtype == torch.float32
    _adjoint_identity(ax, y, x, aty)
    del x, y, ax, aty
    torch.manual_seed(0)
    model = pangnn_amd.AlternateGCN(dev(), None, True, dims=[64, 128], num_nodes=n, skip_connections=True,
                                    deferred_logits=False)
    g.x = torch.arange(n, device=dev())
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(g)
        lu = criterion(out, g.y, g.class_balance)
    lu.backward()
    gu = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    model.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lf, logits = model.loss_and_logits(g, g.y, g.class_balance)
    lf.backward()
    assert bool(torch.isfinite(logits).all()) and bool(torch.isfinite(lf))
    # Under autocast the two forms are not bit-identical: the training form folds the last ELU into the P|Q layer
    # (in-kernel expm1), the forward form applies torch's ELU first — an fp32 ulp apart, which now and then flips the
    # bf16 rounding of a stored P|Q element (2^-8 of the element).  Bound = a fraction of bf16 resolution.
    assert close(logits, out, atol=1e-3, rtol=1e-3) and close(lf, lu, atol=1e-5, rtol=1e-4)
    assert float((logits - out).abs().mean()) < 2e-5
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert bool(torch.isfinite(p.grad).all()), k
            # gradients of bf16-stored tensors are bf16 (one ulp = 2^-8 of the element), rounded at different points
            assert close(p.grad, gu[k], atol=1e-2 * (float(gu[k].abs().max()) + 1e-12) + 1e-12, rtol=5e-2), k


# ---------------------------------------------------------------- bf16-storage modes (config 5: bf16 mixed precision)
@pytest.mark.parametrize("k,m", [(64, 64), (64, 128), (128, 64), (128, 128)])
@pytest.mark.parametrize("n", [1, 33, 40007])
@pytest.mark.parametrize("x_bf16,y_bf16", [(True, False), (False, True), (True, True)])
@pytest.mark.parametrize("in_act", [0, 1])
def test_linear_bf16_storage_is_the_f32_kernel_with_one_rounding(k, m, n, x_bf16, y_bf16, in_act):
    """pangnn_linear_act_{fwd,wgrad}_mixed: a matrix stored as bfloat16 is read exactly (bf16 -> f32 is a shift) and
    a bf16 result is the f32 kernel's result rounded to nearest even once — so every output must equal, BIT FOR BIT,
    what the f32 entry points give on the up-converted inputs (followed by torch's RNE cast where the storage is
    bf16).  Gradients of bf16 tensors are bf16 (autograd's rule, and the reference's under autocast)."""
    from pangnn_amd import functional as PF
    torch.manual_seed(n + k + m + in_act)
    bf, f32 = torch.bfloat16, torch.float32
    x = torch.randn(n, k, device=dev()).to(bf if x_bf16 else f32)
    w, b = torch.randn(m, k, device=dev()) / 8, torch.randn(m, device=dev())
    g = torch.randn(n, m, device=dev()).to(bf if y_bf16 else f32)
    xs = x.clone().requires_grad_(True)
    ws, bs = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    out = PF.linear(xs, ws, bs, in_act, bf if y_bf16 else None)
    assert out.dtype == (bf if y_bf16 else f32)
    out.backward(g)
    # the f32 kernels on the same VALUES
    xr = x.float().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = PF.linear(xr, wr, br, in_act)
    ref.backward(g.float())
    assert torch.equal(out, ref.detach().to(out.dtype))
    assert xs.grad.dtype == x.dtype and torch.equal(xs.grad, xr.grad.to(x.dtype))
    assert torch.equal(ws.grad, wr.grad) and torch.equal(bs.grad, br.grad)


@pytest.mark.parametrize("e", [1, 33, 1000, 70001])
@pytest.mark.parametrize("skip", [False, True])
def test_decoder_on_bf16_tables_equals_decoder_on_upconverted_tables(e, skip):
    """pangnn_decoder_train_mixed / pangnn_decoder_mlp_infer_mixed with bfloat16 P | Q: the gather reads half the
    bytes, the arithmetic is unchanged — logits, loss and every gradient are bit-identical to the f32 entry points on
    the up-converted tables (sorted lists: run-sum path; unsorted: generic path)."""
    from pangnn_amd import functional as PF
    from pangnn_amd.graph import EdgeStructure
    torch.manual_seed(e + skip)
    n, d = 97, 64
    ei, w = random_graph(n, e, seed=e, isolated=0.0)
    ei = ei[:, torch.argsort(ei[0] * n + ei[1])] if e % 2 else ei
    pq = torch.randn(n, 2 * d, device=dev()).to(torch.bfloat16)
    W2, b2, w3, b3, cv = (t.to(dev()) for t in (torch.randn(d, d) / 8, torch.randn(d), torch.randn(d), torch.randn(1),
                                                 torch.randn(d)))
    extra = (w / 40).to(dev()) if skip else None
    y = (torch.rand(e) < 0.3).float().to(dev())
    pw = torch.tensor(2.5, device=dev())
    st = EdgeStructure(ei.to(dev()), n)
    res = []
    for tab in (pq, pq.float()):
        leaf = tab.clone().requires_grad_(True)
        ws = [t.clone().requires_grad_(True) for t in (W2, b2, w3, b3, cv)]
        loss, logits = PF.decoder_loss_pq(leaf, st, extra, ws[4] if skip else None, ws[0], ws[1], ws[2], ws[3], y, pw, e)
        loss.backward()
        with torch.no_grad():
            inf = PF.decoder_mlp_pq(tab, st, extra, cv if skip else None, W2, b2, w3, b3)
        res.append((loss.detach(), logits, inf, leaf.grad, [t.grad for t in ws[: 5 if skip else 4]]))
    (l16, lg16, inf16, g16, gw16), (l32, lg32, inf32, g32, gw32) = res
    assert torch.equal(l16, l32) and torch.equal(lg16, lg32) and torch.equal(inf16, inf32) and torch.equal(inf16, lg16)
    assert g16.dtype == torch.bfloat16 and torch.equal(g16, g32.to(torch.bfloat16))
    for a, b in zip(gw16, gw32):
        assert torch.equal(a, b)
    # separate p and q tensors (the partitioned model's form) take the same path
    p16, q16 = pq[:, :d].contiguous(), pq[:, d:].contiguous()
    lsep, lgsep = PF.decoder_loss(p16, q16, st, extra, cv if skip else None, W2, b2, w3, b3, y, pw, e)
    assert torch.equal(lgsep, lg16) and torch.equal(lsep, l16)


def test_model_under_bf16_autocast_stores_linear_outputs_as_bf16():
    """Under bf16 autocast the reference's Linear layers return bf16 tensors (src/gnn.py:93,111,173 with accelerate's
    mixed precision); here the same tensors — the 128-wide hidden pre-activation, the 64-wide x W^T rows that are
    propagated, and the decoder's P | Q — are WRITTEN as bfloat16 by the linear kernel and read as stored by the next
    kernel (linear / propagate / decoder gather); no fp32 copy of them exists."""
    import pangnn_amd
    from pangnn_amd import functional as PF
    g, gd, oracle, model = _pair("cfg2_sim_1000x5", (64, 128), dict())
    from torch.utils._python_dispatch import TorchDispatchMode
    seen, first = [], []

    class Spy(TorchDispatchMode):
        """the dense layers are the dispatcher op pangnn::linear (C++ implementation): watch the op, not a Python function"""
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            if func is torch.ops.pangnn.linear.default:
                seen.append((args[0].dtype, tuple(args[1].shape), out.dtype))
            if func is torch.ops.pangnn.embed_conv_in.default:
                first.append(out.dtype)
            return out
    with Spy(), torch.autocast("cuda", dtype=torch.bfloat16):
        loss, logits = model.loss_and_logits(gd, gd.y, None)
        loss.backward()
    bf, f32 = torch.bfloat16, torch.float32
    # conv_in(embedding(x)) (the rank-2 operator writes the hidden pre-activation as bf16), conv_out's dense part (bf16 in
    # -> bf16 rows), P|Q (f32 z -> bf16)
    assert first == [bf]
    assert (bf, (64, 128), bf) in seen and (f32, (128, 64), bf) in seen
    assert logits.dtype == f32 and all(p.grad is None or p.grad.dtype == f32 for p in model.parameters())
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ref = oracle(g).float()
    exact = oracle(g)
    scale = float(exact.abs().max())
    assert float((logits.cpu() - ref).abs().max()) < 5e-2 * scale
    assert float((logits.cpu() - exact).abs().max()) <= float((ref - exact).abs().max()) + 1e-3 * scale


# ---------------------------------------------------------------- long rows
@pytest.mark.parametrize("bf16", [False, True], ids=["f32", "bf16-rows"])
def test_hub_rows_are_split_into_segments_and_summed_in_order(bf16):
    """A hub of the similarity graph (one target with 60 000 in-edges among 3 000 nodes, and a source with 20 000 out-edges):
    the propagate runs the wave-per-row kernel over segments of about sqrt(longest row) entries (graph.long_segment) and adds a row's partials with
    the contiguous part sum (graph.CSR.long_rows) — same sums as the oracle, forward and transposed, through both routes
    (ctypes and the C++ op), bitwise reproducible; a graph without long rows has no segment table."""
    from pangnn_amd import functional as PF
    from pangnn_amd import graph as G
    from pangnn_amd.graph import structure_of
    n, e = 3000, 120_000
    gen = torch.Generator().manual_seed(11)
    src = torch.randint(0, n, (e,), generator=gen)
    dst = torch.randint(0, n, (e,), generator=gen)
    dst[:60_000] = 17                                   # the hub target
    src[60_000:80_000] = 23                             # a hub source (long row of the transposed structure)
    w = torch.rand(e, generator=gen) * 80 + 1
    ei = torch.stack([src, dst])
    x = torch.randn(n, 64, generator=gen)
    eid, wd = ei.to(dev()), w.to(dev())
    xd = x.to(dev()).bfloat16() if bf16 else x.to(dev())
    xr = xd.float().cpu()
    st = structure_of(eid, n)
    assert st.by_dst.long_rows() is not None and st.by_src.long_rows() is not None
    seg_ptr, parts_rowptr = st.by_dst.long_rows()
    lens = (seg_ptr[1:] - seg_ptr[:-1])
    seg = G.long_segment(int((st.by_dst.rowptr[1:] - st.by_dst.rowptr[:-1]).max()))
    assert seg == 256 and int(lens.max()) <= seg and int(parts_rowptr[-1]) == seg_ptr.shape[0] - 1
    assert int(parts_rowptr[18] - parts_rowptr[17]) == -(-int(st.by_dst.rowptr[18] - st.by_dst.rowptr[17]) // seg)
    nrm = st.gcn_norm(wd)
    bias = torch.randn(64, device=dev())
    norm_ref = go.gcn_norm(ei, w, n)
    want = go.propagate_add(xr, ei, norm_ref) + bias.cpu()
    tol = dict(atol=2e-3, rtol=2e-3)                    # 60 000-term fp32 sums against the oracle's index_add order
    y1 = PF.spmm_csr(st.by_dst, nrm.by_dst, xd, n, bias=bias)
    y2 = torch.ops.pangnn.gcn_propagate(xd, bias, eid, wd, False, False)
    assert torch.equal(y1, y2) and torch.equal(y1, PF.spmm_csr(st.by_dst, nrm.by_dst, xd, n, bias=bias))
    assert close(y1, want, **tol)
    hub = float((y1[17].cpu() - want[17]).abs().max() / want[17].abs().max())
    assert hub < 1e-4                                   # the hub row itself: relative to its own scale
    # transposed propagate (the backward of the layer): by-source order, its own segment table
    g = torch.randn(n, 64, device=dev())
    gx = PF.spmm_csr(st.by_src, nrm.by_src, g, n)
    want_t = torch.zeros(n, 64).index_add_(0, ei[0], g.cpu()[ei[1]] * norm_ref[:, None])
    assert close(gx, want_t, **tol)
    xg = xd.float().clone().requires_grad_(True)
    torch.ops.pangnn.gcn_propagate(xg, None, eid, wd, False, False).backward(g)
    assert torch.equal(xg.grad, gx)
    # accumulate into a given buffer
    acc = torch.ones(n, 64, device=dev())
    PF.spmm_csr(st.by_src, nrm.by_src, g, n, out=acc, accumulate=True)
    assert close(acc, gx + 1.0, atol=1e-5, rtol=1e-6)
    # no long rows: no table
    ei_s, _ = random_graph(500, 20_000, seed=2)
    assert structure_of(ei_s.to(dev()), 500).by_dst.long_rows() is None
