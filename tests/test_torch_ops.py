"""torch.ops.pangnn.* (csrc/torch_ops.cpp: TORCH_LIBRARY(pangnn, ...) over the C ABI; pangnn_amd/torch_ops.py: fake
kernels, autograd, autocast).  CPU suite: registration, schemas, fake-tensor tracing, loud failure on CPU tensors.
GPU suite: the dispatcher ops against the ctypes path (bit-identical) and the oracle, autocast, torch.compile."""
import pytest
import torch

from conftest import random_graph
from oracle import gcn_oracle as go

OPS = ["csr_from_coo", "gcn_norm", "spmm", "propagate", "edge_gather_concat", "segment_sum_rows", "segment_max_rows",
       "segment_max_bwd",
       # the per-step operators of the train step (torch_ops.py, round 3)
       "linear", "linear_backward", "gcn_propagate", "gcn_propagate_backward", "embed_conv_in", "embed_conv_in_backward",
       "embed_conv_in_linear", "embed_conv_in_linear_backward",
       "embed_propagate", "embed_propagate_backward", "decoder_loss", "decoder_mlp", "decoder_mlp_backward",
       "bce_with_logits"]


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


def test_step_operators_trace_with_fake_tensors():
    """forward and backward ops of the train step: shapes / dtypes from the fake kernels alone (what torch.compile and
    FakeTensor tracing see); no GPU, no launch"""
    import pangnn_amd  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    ops = torch.ops.pangnn
    with FakeTensorMode():
        dev, n, e = "cuda", 20, 50
        f = lambda *s: torch.empty(*s, device=dev)                     # noqa: E731
        ei = torch.empty(2, e, dtype=torch.int64, device=dev)
        x, w, b = f(n, 64), f(128, 64), f(128)
        y = ops.linear(x, w, b, 1, 1)
        assert y.shape == (n, 128) and y.dtype == torch.bfloat16
        gx, gw, gb = ops.linear_backward(y, x, w, 1, True, True)
        assert gx.shape == x.shape and gw.shape == w.shape and gb.shape == b.shape
        assert ops.linear_backward(y, x, w, 0, False, False)[0].numel() == 0
        z = ops.gcn_propagate(x.to(torch.bfloat16), b[:64], ei, f(e), True, 0)
        assert z.shape == (n, 64) and z.dtype == torch.float32
        gz, gbias = ops.gcn_propagate_backward(z, ei, None, True, True, 1)
        assert gz.dtype == torch.bfloat16 and gbias.shape == (64,)
        xt, ew, eb = f(n, 1), f(64, 1), f(64)
        h = ops.embed_conv_in(xt, ew, eb, w, b, ei, None, 0)
        assert h.shape == (n, 128) and h.dtype == torch.float32
        assert [tuple(t.shape) for t in ops.embed_conv_in_backward(h, xt, ew, eb, w, ei, None, True)] == \
            [(64, 1), (64,), (128, 64), (128,)]
        yl = ops.embed_conv_in_linear(xt, ew, eb, w, b, f(64, 128), None, ei, None)
        assert yl.shape == (n, 64) and yl.dtype == torch.float32
        assert [tuple(t.shape) for t in ops.embed_conv_in_linear_backward(yl, xt, ew, eb, w, b, f(64, 128), ei, None, False)] == \
            [(64, 1), (64,), (128, 64), (128,), (64, 128), (0,)]
        a = ops.embed_propagate(xt, ew, eb, ei, None)
        assert a.shape == (n, 64)
        assert [tuple(t.shape) for t in ops.embed_propagate_backward(a, xt, ei, None)] == [(64, 1), (64,)]
        pq, w2, b2, w3, b3, cv = f(n, 128), f(64, 64), f(64), f(64), f(1), f(64)
        out = ops.decoder_loss(pq, ei, f(e), cv, w2, b2, w3, b3, f(e), f(1), e, None)
        assert [tuple(t.shape) for t in out] == [(), (e,), (n, 128), (64,), (64, 64), (64,), (64,), (1,)]
        assert ops.decoder_loss(pq, ei, None, None, w2, b2, w3, b3, f(e), None, e, None)[3].numel() == 0
        lg = ops.decoder_mlp(pq.to(torch.bfloat16), ei, None, None, w2, b2, w3, b3)
        assert lg.shape == (e,) and lg.dtype == torch.float32
        assert [tuple(t.shape) for t in ops.decoder_mlp_backward(lg, pq, ei, f(e), cv, w2, b2, w3, b3)] == \
            [(n, 128), (64,), (64, 64), (64,), (64,), (1,)]
        loss, g = ops.bce_with_logits(lg, f(e), None, e)
        assert loss.shape == () and g.shape == (e,)


def test_dispatcher_route_is_the_default_and_auto_follows_observers():
    """PANGNN_DISPATCHER_OPS: the registered ops are the default route for every call (round 4); `auto` = round 3's
    default: torch.ops.pangnn.* under a tracer / dispatch mode, the direct autograd.Functions for unobserved eager calls;
    a partitioned shard's rectangular structure and `0` never take the ops"""
    import os
    from types import SimpleNamespace
    from pangnn_amd import functional as PF
    from torch._subclasses.fake_tensor import FakeTensorMode
    from torch.utils._python_dispatch import TorchDispatchMode
    if os.environ.get("PANGNN_DISPATCHER_OPS", "1") != "1":
        pytest.skip("route forced by the environment")
    assert PF.USE_DISPATCHER_OPS is True and not PF.observed() and PF._via_ops()
    assert not PF._via_ops(SimpleNamespace(num_src=7, num_nodes=5))              # rectangular (a shard): direct route
    old, PF.USE_DISPATCHER_OPS = PF.USE_DISPATCHER_OPS, "auto"
    try:
        assert not PF._via_ops()
        with FakeTensorMode():
            assert PF.observed() and PF._via_ops()

        class Spy(TorchDispatchMode):
            def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                return func(*args, **(kwargs or {}))

        with Spy():
            assert PF._via_ops()
        PF.USE_DISPATCHER_OPS = False
        with Spy():
            assert not PF._via_ops()
    finally:
        PF.USE_DISPATCHER_OPS = old
    assert PF._via_ops()


def test_ops_refuse_cpu_tensors():
    import pangnn_amd  # noqa: F401
    ei = torch.tensor([[0, 1], [1, 0]])
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.pangnn.csr_from_coo(ei, 2, 1)
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.pangnn.linear(torch.randn(4, 64), torch.randn(64, 64), None, 0, False)
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.pangnn.gcn_propagate(torch.randn(3, 64), None, torch.tensor([[0, 1], [1, 2]]), None, True, False)
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.pangnn.decoder_mlp(torch.randn(3, 128), torch.tensor([[0, 1], [1, 2]]), None, None, torch.randn(64, 64),
                                     torch.randn(64), torch.randn(64), torch.randn(1))


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
    for a, c in zip(res[0][:2], res[1][:2]):
        assert torch.equal(a, c)                                   # same kernels: bit-identical
    # the bias gradient is a column sum: functional's one-launch short-matrix kernel vs torch's reduction inside the raw op's
    # autograd formula (which must stay traceable): same sum, different order of additions
    assert torch.allclose(res[0][2], res[1][2], atol=1e-4, rtol=1e-5)
    assert torch.allclose(res[0][2].cpu(), go_.sum(0), atol=1e-4, rtol=1e-5)
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
        y_h = torch_ops.propagate(x, None, st, norm)               # rows gathered as float16 (round 5), fp32 sums
    assert y_h.dtype == torch.float32
    assert torch.equal(y_h, torch_ops.propagate(x.to(torch.float16), None, st, norm))
    assert torch.allclose(y_h, full, atol=4e-3, rtol=4e-3) and not torch.equal(y_h, full)

    def fn(xx):
        return torch_ops.propagate(xx, None, st, norm).relu().sum()

    xc = x.clone().requires_grad_(True)
    out = torch.compile(fn, backend="aot_eager")(xc)               # traces through the fake kernel + autograd formula
    out.backward()
    xe = x.clone().requires_grad_(True)
    fn(xe).backward()
    assert torch.allclose(out, fn(x)) and torch.allclose(xc.grad, xe.grad)


MODEL_CASES = {
    "default": dict(),
    "skip": dict(skip_connections=True),
    "union": dict(union_edge_weights=True, neighbours=3),
    "base": dict(base_model=True),
    "layerwise": dict(_fuse=False),
    "two_operators": dict(_first_dense=False),
    "round2_first_layer": dict(_fuse="propagate"),
    "wide": dict(_dims=[64, 128], skip_connections=True),
    "bf16": dict(_dims=[64, 128], _autocast=True),
    "infer_decoder": dict(_fused_decoder_only=True),
}


def _model_and_graph(case, dev):
    import pangnn_amd
    from conftest import copy_graph, whole_graph_from_golden
    kw = dict(MODEL_CASES[case])
    fuse, dims, first_dense = kw.pop("_fuse", True), kw.pop("_dims", [64, 64]), kw.pop("_first_dense", True)
    autocast, via_forward = kw.pop("_autocast", False), kw.pop("_fused_decoder_only", False)
    g = copy_graph(whole_graph_from_golden("cfg2_sim_1000x5"), dev)
    if kw.get("union_edge_weights"):
        g.edge_attr = g.union_edge_attr       # dataset.py:380: Data(x, ei, union_edge_weights, y)
    torch.manual_seed(0)
    # deferred_logits off: `via_forward` means forward() launching the inference decoder op and criterion the loss op
    model = pangnn_amd.AlternateGCN(dev, None, False, dims=dims, fuse_embedding=fuse, fuse_first_dense=first_dense,
                                    deferred_logits=False, **kw)
    pw = (g.y == 0).sum() / g.y.sum()
    return model, g, pw, autocast, via_forward


def _one_step(model, g, pw, autocast, via_forward, fn=None):
    from pangnn_amd.train import criterion
    model.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        if via_forward:                                   # forward() + criterion: decoder_mlp / bce ops and their backward
            logits = model(g)
            loss = criterion(logits, g.y, pw)
        else:
            loss, logits = (fn or model.loss_and_logits)(g, g.y, pw)
    loss.backward()
    return (loss.detach().clone(), logits.detach().clone(),
            {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})


def _assert_same_step(a, b):
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert a[2].keys() == b[2].keys() and len(a[2]) >= 8
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(MODEL_CASES))
def test_model_through_dispatcher_ops_is_bit_identical(case):
    """the default route (torch.ops.pangnn.* for every per-step operator) against the ctypes autograd.Functions: the
    same kernels behind both, so loss, logits and every parameter gradient agree bit for bit"""
    from pangnn_amd import functional as PF
    dev = torch.device("cuda:0")
    model, g, pw, autocast, via_forward = _model_and_graph(case, dev)
    outs = []
    for flag in (False, True):
        old, PF.USE_DISPATCHER_OPS = PF.USE_DISPATCHER_OPS, flag
        try:
            outs.append(_one_step(model, g, pw, autocast, via_forward))
        finally:
            PF.USE_DISPATCHER_OPS = old
    _assert_same_step(outs[0], outs[1])


@pytest.mark.gpu
def test_an_observer_sees_every_step_operator_as_a_registered_op():
    """under a TorchDispatchMode (default routing) every per-step operator of a train step is a torch.ops.pangnn.* call"""
    from torch.utils._python_dispatch import TorchDispatchMode
    dev = torch.device("cuda:0")
    model, g, pw, *_ = _model_and_graph("default", dev)
    seen = []

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            if func.namespace == "pangnn":
                seen.append(func._schema.name.split("::")[1])
            return func(*args, **(kwargs or {}))

    with Spy():
        loss, _ = model.loss_and_logits(g, g.y, pw)
        loss.backward()
    for name in ("embed_conv_in_linear", "gcn_propagate", "linear", "decoder_loss", "embed_conv_in_linear_backward",
                 "gcn_propagate_backward", "linear_backward"):
        assert name in seen, (name, seen)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["default", "skip", "union", "bf16", "infer_decoder"])
def test_compiled_step_is_one_graph_and_bit_identical(case):
    """torch.compile(backend="aot_eager", fullgraph=True) of the model's loss_and_logits (forward()+criterion for the
    inference decoder): dynamo traces the whole forward into ONE graph of torch.ops.pangnn.* calls, AOTAutograd derives the
    backward graph from the registered formulas (backward ops are registered ops too), and the compiled step returns what
    the eager step returns, bit for bit."""
    from pangnn_amd.train import criterion
    dev = torch.device("cuda:0")
    model, g, pw, autocast, via_forward = _model_and_graph(case, dev)
    eager = _one_step(model, g, pw, autocast, via_forward)
    graphs = []

    def backend(gm, example_inputs):
        from torch._dynamo.backends.debugging import aot_eager
        graphs.append([n.target for n in gm.graph.nodes if n.op == "call_function"])
        return aot_eager(gm, example_inputs)

    torch._dynamo.reset()
    if via_forward:
        def step(graph, y, w):
            return criterion(model(graph), y, w), None
        compiled = torch.compile(step, backend=backend, fullgraph=True)
        model.zero_grad()
        loss, _ = compiled(g, g.y, pw)
        loss.backward()
        assert torch.equal(loss.detach(), eager[0])
        grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
        assert grads.keys() == eager[2].keys()
        for k in grads:
            assert torch.equal(grads[k], eager[2][k]), k
    else:
        compiled = torch.compile(model.loss_and_logits, backend=backend, fullgraph=True)
        _assert_same_step(eager, _one_step(model, g, pw, autocast, False, fn=compiled))
    assert len(graphs) == 1, f"{len(graphs)} graphs"
    names = {str(t) for t in graphs[0]}
    assert any("pangnn" in t for t in names), names
    torch._dynamo.reset()


@pytest.mark.gpu
def test_opcheck_of_the_step_operators():
    """torch.library.opcheck: schema (no undeclared mutation / aliasing), fake kernel against the real one, autograd
    registration, and AOT dispatch of each forward op on real operands"""
    from pangnn_amd.graph import structure_of
    dev = torch.device("cuda:0")
    n, e = 300, 2400
    ei, w = random_graph(n, e, seed=2)
    ei, w = ei.to(dev), w.to(dev)
    structure_of(ei, n)
    torch.manual_seed(1)
    r = lambda *s: torch.randn(*s, device=dev)                         # noqa: E731
    p = lambda *s: (torch.randn(*s, device=dev) * 0.2).requires_grad_(True)   # noqa: E731
    ops = torch.ops.pangnn
    tests = ("test_schema", "test_faketensor", "test_autograd_registration", "test_aot_dispatch_static")
    y = (torch.rand(e, device=dev) < 0.3).float()
    cases = [
        (ops.linear, (p(n, 64), p(128, 64), p(128), 1, 0)),
        (ops.linear, (r(n, 128).to(torch.bfloat16), p(64, 128), None, 0, 1)),
        (ops.linear, (r(n, 128).to(torch.float16), p(64, 128), None, 1, 2)),
        (ops.gcn_propagate, (p(n, 64), p(64), ei, w, True, 0)),
        (ops.gcn_propagate, (p(n, 128), None, ei, None, False, 0)),
        (ops.embed_conv_in, (r(n, 1), p(64, 1), p(64), p(128, 64), p(128), ei, w, False)),
        (ops.embed_conv_in_linear, (r(n, 1), p(64, 1), p(64), p(128, 64), p(128), p(64, 128), None, ei, w)),
        (ops.embed_conv_in_linear, (r(n, 1), p(64, 1), p(64), p(64, 64), None, p(64, 64), p(64), ei, None)),
        (ops.embed_propagate, (r(n, 1), p(64, 1), p(64), ei, w)),
        (ops.decoder_mlp, (p(n, 128), ei, w, p(64), p(64, 64), p(64), p(64), p(1))),
        (ops.decoder_loss, (p(n, 128), ei, None, None, p(64, 64), p(64), p(64), p(1), y, None, e, None)),
        (ops.bce_with_logits, (p(e), y, r(1).abs(), e)),
    ]
    for op, args in cases:
        torch.library.opcheck(op, args, test_utils=tests)


@pytest.mark.gpu
def test_native_structure_registry_builds_on_miss_keys_on_identity_and_forgets():
    """csrc/graph_ops.cpp: an op called on raw tensors nobody prepared finds nothing in the registry, calls the one Python
    hook (pangnn::_prepare_structure), and carries on; entries are keyed on the tensor's IDENTITY (address, version counter,
    storage): an equal copy is another entry, an in-place change of the edge list is another structure, a freed tensor's
    address handed to a new tensor is not a stale hit; clear_cache / forget empty it."""
    import pangnn_amd  # noqa: F401
    from pangnn_amd import graph as G
    ops = torch.ops.pangnn
    dev = torch.device("cuda")
    n = 300
    ei, w = random_graph(n, 2000, seed=3)
    ei, w = ei.to(dev), w.to(dev)
    x = torch.randn(n, 64, device=dev)

    def ref(ei_, w_):
        return go.propagate_add(x.cpu(), ei_.cpu(), go.gcn_norm(ei_.cpu(), None if w_ is None else w_.cpu(), n))

    G.clear_cache()
    assert ops._registry_size(ei) == 0
    y = ops.gcn_propagate(x, None, ei, w, False, 0)                 # miss -> hook -> built, pushed, used
    assert ops._registry_size(ei) == 1 and torch.allclose(y.cpu(), ref(ei, w), rtol=1e-4, atol=1e-4)
    assert torch.equal(ops.gcn_propagate(x, None, ei, w, False, 0), y) and ops._registry_size(ei) == 1      # hit
    y_unit = ops.gcn_propagate(x, None, ei, None, False, 0)         # another normalisation of the same structure
    assert ops._registry_size(ei) == 1 and torch.allclose(y_unit.cpu(), ref(ei, None), rtol=1e-4, atol=1e-4)
    ei2 = ei.clone()                                                     # equal content, another tensor: another entry
    assert torch.equal(ops.gcn_propagate(x, None, ei2, w, False, 0), y) and ops._registry_size(ei) == 2
    k = int(((ei[0] < 200) & (ei[1] < 200)).nonzero()[10])               # an edge between nodes that have in-edges (norm != 0)
    ei2[0, k] = (ei2[0, k] + 7) % 200                                    # in-place: the version counter moves on
    y3 = ops.gcn_propagate(x, None, ei2, w, False, 0)
    assert torch.allclose(y3.cpu(), ref(ei2, w), rtol=1e-4, atol=1e-4) and not torch.equal(y3, y)
    # backward through the C++ autograd formula (transposed propagate over the by-source order, built on ITS miss)
    xg = x.clone().requires_grad_(True)
    b = torch.zeros(64, device=dev, requires_grad=True)
    ops.gcn_propagate(xg, b, ei, w, False, 0).square().sum().backward()
    xr = x.cpu().clone().requires_grad_(True)
    (go.propagate_add(xr, ei.cpu(), go.gcn_norm(ei.cpu(), w.cpu(), n))).square().sum().backward()
    assert torch.allclose(xg.grad.cpu(), xr.grad, rtol=1e-3, atol=1e-3) and b.grad.shape == (64,)
    # a freed tensor's address handed to a new tensor of the same shape must not hit the dead entry
    ptr = ei2.data_ptr()
    del ei2
    for _ in range(8):
        t = torch.randint(0, n, (2, 2000), device=dev)
        if t.data_ptr() == ptr:
            break
    out = ops.gcn_propagate(x, None, t, None, False, 0)
    assert torch.allclose(out.cpu(), ref(t, None), rtol=1e-4, atol=1e-4)
    G.forget(G.structure_key(ei, n), ei)
    G.clear_cache()
    assert ops._registry_size(ei) == 0
    # the decoder ops need no table for inference and the run-sum plans for training: both through the hook
    pq = torch.randn(n, 128, device=dev)
    w2, b2, w3, b3 = torch.randn(64, 64, device=dev) / 8, torch.randn(64, device=dev) / 8, torch.randn(64, device=dev) / 8, torch.zeros(1, device=dev)
    lg = ops.decoder_mlp(pq, ei, None, None, w2, b2, w3, b3)
    yl = (torch.rand(ei.shape[1], device=dev) < 0.3).float()
    out = ops.decoder_loss(pq, ei, None, None, w2, b2, w3, b3, yl, None, ei.shape[1], None)
    assert torch.equal(out[1], lg) and out[2].shape == pq.shape and ops._registry_size(ei) == 1
    h1 = torch.relu(pq[ei[0], :64] + pq[ei[1], 64:])
    want = torch.relu(h1 @ w2.t() + b2) @ w3 + b3
    assert torch.allclose(lg, want, rtol=1e-3, atol=1e-3)
