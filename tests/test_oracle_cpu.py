"""CPU suite (-m "not gpu"): oracle vs golden fixtures, oracle self-consistency, host logic, and
that the C-ABI library loads and exports every symbol include/pangnn_hip.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, sub_graphs_from_golden, random_graph
from oracle import construct_oracle as co
from oracle import gcn_oracle as go

FIXTURES = ["sim_200x4", "cfg1_2genomes", "cfg2_sim_1000x5", "cfg3_5genomes"]


# ---------------------------------------------------------------- construction oracle (PINNED)
@pytest.mark.parametrize("name", FIXTURES)
def test_remove_trivial_cases_matches_reference(name):
    f = load_golden(name)
    s, d, w = co.remove_trivial_cases(f["raw_src"], f["raw_dst"], f["raw_score"], f["genome_of"])
    assert np.array_equal(s, f["flt_src"]) and np.array_equal(d, f["flt_dst"]) and np.array_equal(w, f["flt_score"])


@pytest.mark.parametrize("name", FIXTURES)
def test_normalize_sim_scores_matches_reference_bitwise(name):
    f = load_golden(name)
    s, d, w = co.normalize_sim_scores(f["flt_src"], f["flt_dst"], f["flt_score"], f["genome_of"])
    o1, o2 = co.canonical_order(s, d), co.canonical_order(f["nrm_src"], f["nrm_dst"])
    assert np.array_equal(s[o1], f["nrm_src"][o2]) and np.array_equal(d[o1], f["nrm_dst"][o2])
    assert np.array_equal(w[o1], f["nrm_weight"][o2])          # float64, bit for bit
    assert w.min() >= 1.0 and w.max() <= 81.0 + 1e-9             # preprocessing.py:541 invariant


@pytest.mark.parametrize("name", FIXTURES)
def test_whole_graph_tensors_match_reference(name):
    f = load_golden(name)
    ei, w, y, nb = co.whole_graph(int(f["num_nodes"]), f["nrm_src"], f["nrm_dst"], f["nrm_weight"],
                                  f["grp_src"], f["grp_dst"])
    o = co.canonical_order(f["whole_edge_index"][0], f["whole_edge_index"][1])
    assert np.array_equal(ei, f["whole_edge_index"][:, o])       # bit-exact edge_index
    assert np.array_equal(w, f["whole_edge_attr"][o])            # fp32 weights, bit-exact
    assert np.array_equal(y, f["whole_y"][o])
    assert np.array_equal(nb, f["whole_neighbour_edge_index"])   # neighbour graph incl. self loops, order too
    n = int(f["num_nodes"])
    assert nb.shape[1] == n * 3 - 2                              # N(2n+1) - n(n+1), n = 1


def test_golden_shapes_are_the_surveyed_ones():
    # SURVEY.md §4 [probe] table
    expect = {"cfg1_2genomes": (1895, 2017, 5683), "cfg3_5genomes": (4773, 12855, 14317),
              "cfg2_sim_1000x5": (5000, 44903, 14998)}
    for name, (n, e, enb) in expect.items():
        f = load_golden(name)
        assert int(f["num_nodes"]) == n and f["whole_edge_index"].shape[1] == e
        assert f["whole_neighbour_edge_index"].shape[1] == enb


# ---------------------------------------------------------------- arithmetic oracle (UNPINNED: self-checks)
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_gcn_conv_gather_scatter_equals_dense_formulation(dtype):
    torch.manual_seed(0)
    n, e, fi, fo = 57, 400, 12, 9
    ei, w = random_graph(n, e, seed=1)
    x = torch.randn(n, fi, dtype=dtype)
    W = torch.randn(fo, fi, dtype=dtype)
    b = torch.randn(fo, dtype=dtype)
    for weight in (w.to(dtype), None):
        a = go.gcn_conv(x, ei, weight, W, b)
        d = go.gcn_conv_dense(x, ei, weight, W, b)
        tol = 1e-12 if dtype == torch.float64 else 2e-5
        assert torch.allclose(a, d, atol=tol, rtol=tol)


def test_gcn_norm_isolated_and_empty():
    ei = torch.tensor([[0, 1], [2, 2]])
    nrm = go.gcn_norm(ei, torch.tensor([4.0, 12.0]), 4)
    # deg[2] = 16 -> dis 0.25 ; sources 0 and 1 have in-degree 0 -> dis 0 -> norm 0
    assert torch.equal(nrm, torch.zeros(2))
    assert go.gcn_norm(torch.zeros(2, 0, dtype=torch.long), None, 3).numel() == 0


def test_oracle_gradients_fp64_gradcheck():
    torch.manual_seed(0)
    n, e = 9, 30
    ei, w = random_graph(n, e, seed=3, isolated=0.2)
    x = torch.randn(n, 5, dtype=torch.float64, requires_grad=True)
    W = torch.randn(4, 5, dtype=torch.float64, requires_grad=True)
    b = torch.randn(4, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda x, W, b: go.gcn_conv(x, ei, w.double(), W, b), (x, W, b))


def test_collate_offsets_index_attributes():
    subs = sub_graphs_from_golden("cfg1_2genomes", count=5)
    b = go.collate(subs)
    n = sum(g.x.shape[0] for g in subs)
    assert b.x.shape[0] == n and b.ptr[-1] == n
    off = subs[0].x.shape[0]
    e0 = subs[0].edge_index.shape[1]
    e1 = subs[1].edge_index.shape[1]
    assert torch.equal(b.edge_index[:, e0:e0 + e1], subs[1].edge_index + off)
    nb0 = subs[0].neighbour_edge_index.shape[1]
    nb1 = subs[1].neighbour_edge_index.shape[1]
    assert torch.equal(b.neighbour_edge_index[:, nb0:nb0 + nb1], subs[1].neighbour_edge_index + off)
    assert int(b.edge_index.max()) < n and int(b.neighbour_edge_index.max()) < n


def test_edge_conv_oracle_empty_rows_are_zero():
    torch.manual_seed(0)
    m = go.EdgeConvOracle(3, 4)
    x = torch.randn(5, 3)
    ei = torch.tensor([[0, 1, 2], [1, 1, 3]])
    out = m(x, ei)
    assert torch.equal(out[0], torch.zeros(4)) and torch.equal(out[4], torch.zeros(4))
    assert out[1].abs().sum() > 0


# ---------------------------------------------------------------- host logic of the product (no GPU compute)
def test_product_state_dict_keys_and_shapes_match_reference_layout():
    import pangnn_amd
    for skip in (False, True):
        m = pangnn_amd.AlternateGCN(dims=[64, 128], skip_connections=skip)
        o = go.AlternateGCNOracle(dims=(64, 128), flags=go.default_flags(skip_connections=skip))
        sd, so = m.state_dict(), o.state_dict()
        assert list(sd.keys()) == list(so.keys())
        assert [tuple(v.shape) for v in sd.values()] == [tuple(v.shape) for v in so.values()]
        assert list(sd.keys())[:4] == ["embedding.weight", "embedding.bias", "conv_in.bias", "conv_in.lin.weight"]
        assert sd["mlp.0.weight"].shape == (64, 129 if skip else 128)
    assert sum(p.numel() for p in pangnn_amd.AlternateGCN(dims=[64, 128]).parameters()) == 53953


def test_args_namespace_adapter():
    import pangnn_amd
    from types import SimpleNamespace
    a = SimpleNamespace(union_edge_weights=True, base_model=False, skip_connections=True, decoder="cosine",
                        neighbours=4, unrelated=1)
    m = pangnn_amd.AlternateGCN(None, None, False, dims=[8, 16], args=a)
    assert m.flags.union_edge_weights and m.flags.skip_connections and m.flags.decoder == "cosine"
    assert m.flags.neighbours == 4
    with pytest.raises(TypeError):
        pangnn_amd.AlternateGCN(dims=[8, 16], not_a_flag=True)


def test_batch_from_data_list_matches_oracle_collate():
    from pangnn_amd.data import Batch, Data
    subs = sub_graphs_from_golden("cfg3_5genomes", count=32)
    b = Batch.from_data_list([Data(g.x, g.edge_index, g.edge_attr, g.y,
                                   neighbour_edge_index=g.neighbour_edge_index) for g in subs])
    o = go.collate(subs)
    for k in ("x", "edge_index", "edge_attr", "y", "neighbour_edge_index", "ptr", "batch"):
        assert torch.equal(getattr(b, k), getattr(o, k)), k
    assert b.num_graphs == 32


def test_product_refuses_cpu_tensors():
    import pangnn_amd
    from pangnn_amd._lib import PangnnHipError
    m = pangnn_amd.GCNConv(4, 4)
    with pytest.raises(PangnnHipError):
        m(torch.randn(3, 4), torch.tensor([[0, 1], [1, 2]]))
    # every operator of the functional layer: no silent torch-on-CPU path (shapes the HIP kernels do not cover go to
    # hipBLASLt on the GPU, never to the host)
    from pangnn_amd import functional as PF
    for k, mm in ((64, 128), (5, 7)):
        with pytest.raises(PangnnHipError):
            PF.linear(torch.randn(3, k), torch.randn(mm, k), torch.randn(mm))
    with pytest.raises(PangnnHipError):
        PF.bce_with_logits(torch.randn(5), torch.ones(5))


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pangnn_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(pangnn_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 12
    lib = ctypes.CDLL(os.path.join(ROOT, "pangnn_amd", "libpangnn_hip.so"))
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/pangnn_hip.h but not exported"
    from pangnn_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared               # python binding covers the header 1:1
    lib.pangnn_abi_version.restype = ctypes.c_int
    assert lib.pangnn_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define PANGNN_ABI_VERSION (\d+)", hdr).group(1)) == 3


def test_shipped_code_objects_pass_the_isa_gate():
    """tools/check_isa.py (run by csrc/Makefile at every link) on the built objects: no packed-f32 instruction that takes
    the high dword of src1 for its low result — the form gfx950 executes wrongly beside another wave's MFMAs (DESIGN.md §4,
    tools/pk_opsel_probe.hip).  Also checks that the gate FIRES on an instruction of that form."""
    import subprocess
    import sys
    objs = sorted(os.path.join(ROOT, "pangnn_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "pangnn_amd", "csrc"))
                  if f.endswith(".o") and not f.startswith("t_"))      # t_*.o: host objects of libpangnn_torch.so
    assert len(objs) >= 6, "build first: python -c 'import __graft_entry__ as g; g.build()'"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_isa.py")] + objs, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa
    assert check_isa.BAD.search("v_pk_mul_f32 v[120:121], v[168:169], v[224:225] op_sel:[0,1]")
    assert check_isa.BAD.search("v_pk_fma_f32 v[4:5], v[0:1], v[2:3], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,1,1]")
    assert check_isa.BAD.search("v_pk_add_f32 v[24:25], v[20:21], v[20:21] op_sel:[0,1] op_sel_hi:[1,0]")
    assert not check_isa.BAD.search("v_pk_mul_f32 v[118:119], v[118:119], v[224:225] op_sel_hi:[1,0]")
    assert not check_isa.BAD.search("v_pk_fma_f32 v[118:119], v[114:115], v[120:121], v[118:119] op_sel:[1,0,0]")
    assert not check_isa.BAD.search("v_pk_min_u16 v2, v2, v1")


def test_c_abi_argument_errors_without_gpu():
    """argument validation happens before any HIP call, so it is checkable on a CPU-only box"""
    from pangnn_amd import _lib
    lib = _lib.load()
    rc = lib.pangnn_spmm_csr_f32(None, None, None, None, 64, 10, None, None, 64, 5, -1, 64, 0, None)
    assert rc == -1 and b"null pointer" in lib.pangnn_last_error()
    rc = lib.pangnn_csr_build(None, 5, 10, 4, 1, None, None, None, None, 0, None)
    assert rc == -1                                          # ld < E
    rc = lib.pangnn_csr_build(None, 2**31 + 5, 2**31 + 5, 4, 1, ctypes.c_void_p(16), None, None, None, 0, None)
    assert rc == -2                                          # E does not fit int32
    assert lib.pangnn_spmm_csr_f32(None, None, None, None, 64, 10, None, None, 64, 0, -1, 64, 0, None) == 0


def test_c_restatement_agrees_with_torch_restatement():
    """oracle/propagate_oracle.c (built by __graft_entry__.build) vs oracle/gcn_oracle.py"""
    so = os.path.join(ROOT, "oracle", "_build", "libpropagate_oracle.so")
    if not os.path.exists(so):
        import __graft_entry__ as ge
        ge.build()
    lib = ctypes.CDLL(so)
    n, e, f = 300, 5000, 24
    ei, w = random_graph(n, e, seed=21, hub=700)
    x, b = torch.randn(n, f), torch.randn(f)
    ei_c = ei.contiguous()
    norm = torch.empty(e)
    out = torch.empty(n, f)
    P = ctypes.c_void_p
    lib.oracle_gcn_norm_f32(P(ei_c.data_ptr()), ctypes.c_int64(e), ctypes.c_int64(n), P(w.data_ptr()),
                            P(norm.data_ptr()))
    lib.oracle_propagate_f32(P(ei_c.data_ptr()), ctypes.c_int64(e), ctypes.c_int64(n), P(norm.data_ptr()),
                             P(x.data_ptr()), ctypes.c_int64(f), P(b.data_ptr()), P(out.data_ptr()))
    ref_norm = go.gcn_norm(ei, w, n)
    assert torch.allclose(norm, ref_norm, rtol=1e-5, atol=1e-8)
    assert torch.allclose(out, go.propagate_add(x, ei, ref_norm) + b, rtol=1e-4, atol=1e-4)
