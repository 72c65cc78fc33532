"""torch.ops.pangnn.* (csrc/torch_ops.cpp: TORCH_LIBRARY(pangnn, ...) over the C ABI; pangnn_amd/torch_ops.py: fake
kernels, autograd, autocast).  CPU suite: registration, schemas, fake-tensor tracing, loud failure on CPU tensors.
GPU suite: the dispatcher ops against the ctypes path (bit-identical) and the oracle, autocast, torch.compile."""
import pytest
import torch

from conftest import random_graph
from oracle import gcn_oracle as go

OPS = ["csr_from_coo", "gcn_norm", "spmm", "propagate", "edge_gather_concat", "segment_sum_rows", "segment_max_rows",
       "segment_max_bwd"]


def test_ops_are_registered_with_schemas():
    import pangnn_amd  # noqa: F401
    for name in OPS:
        op = getattr(torch.ops.pangnn, name)
        assert str(op.default._schema).startswith(f"pangnn::{name}(")
    assert "Tensor? bias" in str(torch.ops.pangnn.propagate.default._schema)


def test_fake_kernels_trace_without_a_gpu():
    import pangnn_amd  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        dev = "cuda"
        ei = torch.empty(2, 50, dtype=torch.int64, device=dev)
        rp, ot, pm = torch.ops.pangnn.csr_from_coo(ei, 20, 1)
        assert rp.shape == (21,) and ot.dtype == torch.int32 and pm.shape == (50,)
        dis, ns, no = torch.ops.pangnn.gcn_norm(rp, ot, pm, None)
        assert dis.shape == (20,) and ns.shape == (50,) and no.dtype == torch.float32
        x = torch.empty(20, 64, device=dev, dtype=torch.bfloat16)
        y = torch.ops.pangnn.propagate(rp, ot, ns, rp, ot, ns, x, None)
        assert y.shape == (20, 64) and y.dtype == torch.float32
        c = torch.ops.pangnn.edge_gather_concat(torch.empty(20, 64, device=dev), ei, torch.empty(50, device=dev))
        assert c.shape == (50, 129)
        mx, arg = torch.ops.pangnn.segment_max_rows(rp, pm, torch.empty(50, 8, device=dev), 20)
        assert mx.shape == (20, 8) and arg.dtype == torch.int32


def test_ops_refuse_cpu_tensors():
    import pangnn_amd  # noqa: F401
    ei = torch.tensor([[0, 1], [1, 0]])
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.pangnn.csr_from_coo(ei, 2, 1)


@pytest.mark.gpu
def test_dispatcher_propagate_equals_ctypes_path_and_oracle():
    from pangnn_amd import functional as PF, torch_ops
    from pangnn_amd.graph import EdgeStructure
    dev = torch.device("cuda:0")
    n, e, f = 500, 6000, 64
    ei, w = random_graph(n, e, seed=11)
    st = EdgeStructure(ei.to(dev), n)
    norm = st.gcn_norm(w.to(dev))
    torch.manual_seed(0)
    x0, b0 = torch.randn(n, f), torch.randn(f)
    go_ = torch.randn(n, f)
    res = []
    for fn in (lambda x, b: PF._Propagate.apply(x, b, st, norm, None), lambda x, b: torch_ops.propagate(x, b, st, norm)):
        x, b = x0.clone().to(dev).requires_grad_(True), b0.clone().to(dev).requires_grad_(True)
        y = fn(x, b)
        y.backward(go_.to(dev))
        res.append((y.detach(), x.grad, b.grad))
    for a, c in zip(res[0], res[1]):
        assert torch.equal(a, c)                                   # same kernels: bit-identical
    ref = go.propagate_add(x0, ei, go.gcn_norm(ei, w, n)) + b0
    assert torch.allclose(res[1][0].cpu(), ref, atol=1e-4, rtol=1e-4)
    # raw ops: structure + norm through the dispatcher
    rp, ot, pm = torch.ops.pangnn.csr_from_coo(ei.to(dev), n, 1)
    assert torch.equal(rp, st.by_dst.rowptr) and torch.equal(ot, st.by_dst.other) and torch.equal(pm, st.by_dst.perm)
    dis, ns, no = torch.ops.pangnn.gcn_norm(rp, ot, pm, w.to(dev))
    assert torch.equal(ns, norm.by_dst) and torch.equal(no, norm.orig)


@pytest.mark.gpu
def test_dispatcher_ops_reject_operands_on_another_device_or_of_the_wrong_type():
    """the implementations forward raw data_ptr()s: an index / weight tensor left on the host (or of the wrong dtype,
    or of the wrong length) must raise before any kernel sees its pointer (csrc/torch_ops.cpp: operand checks)"""
    from pangnn_amd.graph import EdgeStructure
    dev = torch.device("cuda:0")
    n, e, f = 50, 400, 64
    ei, w = random_graph(n, e, seed=5)
    st = EdgeStructure(ei.to(dev), n)
    norm = st.gcn_norm(w.to(dev))
    d, s_ = st.by_dst, st.by_src
    x = torch.randn(n, f, device=dev)
    ops = torch.ops.pangnn
    good = ops.spmm(d.rowptr, d.other, norm.by_dst, x, None, n)
    bad_calls = [
        lambda: ops.spmm(d.rowptr.cpu(), d.other, norm.by_dst, x, None, n),              # rowptr on the host
        lambda: ops.spmm(d.rowptr, d.other.cpu(), norm.by_dst, x, None, n),              # ids on the host
        lambda: ops.spmm(d.rowptr, d.other, norm.by_dst.cpu(), x, None, n),              # weights on the host
        lambda: ops.spmm(d.rowptr, d.other, norm.by_dst, x, torch.zeros(f), n),          # bias on the host
        lambda: ops.spmm(d.rowptr, d.other.long(), norm.by_dst, x, None, n),             # ids of the wrong dtype
        lambda: ops.spmm(d.rowptr.int(), d.other, norm.by_dst, x, None, n),              # rowptr of the wrong dtype
        lambda: ops.spmm(d.rowptr, d.other, norm.by_dst[:-1], x, None, n),               # one weight short
        lambda: ops.spmm(d.rowptr, d.other, norm.by_dst, x, None, n + 1),                # more rows than rowptr holds
        lambda: ops.propagate(d.rowptr, d.other, norm.by_dst, s_.rowptr, s_.other.cpu(), norm.by_src, x, None),
        lambda: ops.propagate(d.rowptr, d.other, norm.by_dst, s_.rowptr, s_.other, norm.by_src.cpu(), x, None),
        lambda: ops.gcn_norm(d.rowptr, d.other.cpu(), d.perm, None),
        lambda: ops.gcn_norm(d.rowptr, d.other, d.perm.cpu(), None),
        lambda: ops.gcn_norm(d.rowptr, d.other, d.perm, w),                               # weights on the host
        lambda: ops.segment_sum_rows(d.rowptr, d.perm.cpu(), torch.randn(e, f, device=dev), 0, f, n),
        lambda: ops.segment_max_rows(d.rowptr.cpu(), d.perm, torch.randn(e, 8, device=dev), n),
        lambda: ops.segment_max_bwd(torch.randn(n, 8, device=dev), torch.zeros(n, 8, dtype=torch.int32), d.rowptr, e),
        lambda: ops.edge_gather_concat(x, ei, None),                                      # edge_index on the host
        lambda: ops.edge_gather_concat(x, ei.to(dev), w),                                 # extra on the host
    ]
    for i, call in enumerate(bad_calls):
        with pytest.raises(RuntimeError):
            call()
            pytest.fail(f"bad call {i} did not raise")
    torch.cuda.synchronize()
    assert torch.equal(good, ops.spmm(d.rowptr, d.other, norm.by_dst, x, None, n))      # the GPU is still healthy


@pytest.mark.gpu
def test_autocast_policy_and_compile():
    from pangnn_amd import torch_ops
    from pangnn_amd.graph import EdgeStructure
    dev = torch.device("cuda:0")
    n, e, f = 300, 4000, 64
    ei, w = random_graph(n, e, seed=3)
    st = EdgeStructure(ei.to(dev), n)
    norm = st.gcn_norm(w.to(dev))
    x = torch.randn(n, f, device=dev)
    full = torch_ops.propagate(x, None, st, norm)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y16 = torch_ops.propagate(x, None, st, norm)               # rows gathered as bfloat16, fp32 sums
    assert y16.dtype == torch.float32
    assert torch.equal(y16, torch_ops.propagate(x.to(torch.bfloat16), None, st, norm))
    assert torch.allclose(y16, full, atol=3e-2, rtol=3e-2) and not torch.equal(y16, full)
    with torch.autocast("cuda", dtype=torch.float16):
        y_h = torch_ops.propagate(x, None, st, norm)               # no fp16 row format: fp32
    assert torch.equal(y_h, full)

    def fn(xx):
        return torch_ops.propagate(xx, None, st, norm).relu().sum()

    xc = x.clone().requires_grad_(True)
    out = torch.compile(fn, backend="aot_eager")(xc)               # traces through the fake kernel + autograd formula
    out.backward()
    xe = x.clone().requires_grad_(True)
    fn(xe).backward()
    assert torch.allclose(out, fn(x)) and torch.allclose(xc.grad, xe.grad)


@pytest.mark.gpu
def test_model_through_dispatcher_ops_is_bit_identical():
    import pangnn_amd
    from pangnn_amd import functional as PF
    from conftest import copy_graph, whole_graph_from_golden
    dev = torch.device("cuda:0")
    g = copy_graph(whole_graph_from_golden("cfg2_sim_1000x5"), dev)
    torch.manual_seed(0)
    model = pangnn_amd.AlternateGCN(dev, None, False, dims=[64, 64], fuse_embedding=False)
    pw = (g.y == 0).sum() / g.y.sum()
    outs = []
    for flag in (False, True):
        old, PF.USE_DISPATCHER_OPS = PF.USE_DISPATCHER_OPS, flag
        try:
            model.zero_grad()
            loss, logits = model.loss_and_logits(g, g.y, pw)
            loss.backward()
            outs.append((loss.detach().clone(), logits.clone(), [p.grad.clone() for p in model.parameters()
                                                                 if p.grad is not None]))
        finally:
            PF.USE_DISPATCHER_OPS = old
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for a, b in zip(outs[0][2], outs[1][2]):
        assert torch.equal(a, b)
