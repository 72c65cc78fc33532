"""The float16 row format (`--mixed_precision fp16`, /root/reference/src/setup.py:50: accelerate's float16 autocast makes the
reference's Linear outputs — and hence the rows PyG's propagate gathers — float16 tensors).  Storage only: every product and sum
is fp32, exactly as for bfloat16 rows (PANGNN_DTYPE_F16 next to PANGNN_DTYPE_BF16 in include/pangnn_hip.h).  The tests mirror
the bfloat16 ones: a 16-bit operand is read exactly, a 16-bit result is the f32 kernel's result rounded to nearest even once."""
import pytest
import torch

from conftest import copy_graph, random_graph, whole_graph_from_golden
from oracle import gcn_oracle as go

pytestmark = pytest.mark.gpu
F16, BF16, F32 = torch.float16, torch.bfloat16, torch.float32


def dev():
    return torch.device("cuda:0")


def close(a, b, atol, rtol):
    return torch.allclose(a.detach().cpu().double(), b.detach().cpu().double(), atol=atol, rtol=rtol)


@pytest.mark.parametrize("k,m", [(64, 64), (64, 128), (128, 64), (128, 128)])
@pytest.mark.parametrize("n", [1, 33, 40007])
@pytest.mark.parametrize("x16,y16", [(True, False), (False, True), (True, True)])
@pytest.mark.parametrize("in_act", [0, 1])
def test_linear_f16_storage_is_the_f32_kernel_with_one_rounding(k, m, n, x16, y16, in_act):
    """pangnn_linear_act_{fwd,wgrad}_mixed / pangnn_linear_dgrad_mixed with PANGNN_DTYPE_F16: bit for bit the f32 entry points
    on the up-converted inputs followed by torch's round-to-nearest-even cast where the storage is float16; gradients of
    float16 tensors are float16 (autograd's rule, and the reference's under autocast)"""
    from pangnn_amd import functional as PF
    torch.manual_seed(n + k + m + in_act)
    x = torch.randn(n, k, device=dev()).to(F16 if x16 else F32)
    w, b = torch.randn(m, k, device=dev()) / 8, torch.randn(m, device=dev())
    g = torch.randn(n, m, device=dev()).to(F16 if y16 else F32)
    xs = x.clone().requires_grad_(True)
    ws, bs = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    out = PF.linear(xs, ws, bs, in_act, F16 if y16 else None)
    assert out.dtype == (F16 if y16 else F32)
    out.backward(g)
    xr = x.float().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = PF.linear(xr, wr, br, in_act)
    ref.backward(g.float())
    assert torch.equal(out, ref.detach().to(out.dtype))
    assert xs.grad.dtype == x.dtype and torch.equal(xs.grad, xr.grad.to(x.dtype))
    assert torch.equal(ws.grad, wr.grad) and torch.equal(bs.grad, br.grad)


def test_one_call_never_mixes_the_two_16_bit_formats():
    """bfloat16 x with a float16 result (or gate) is refused by the C ABI; functional.linear routes such a call around it"""
    from pangnn_amd import _lib, functional as PF
    lib = _lib.load()
    x = torch.randn(64, 64, device=dev()).to(BF16)
    w = torch.randn(64, 64, device=dev())
    y = torch.empty(64, 64, device=dev(), dtype=F16)
    with torch.cuda.device(dev()):
        rc = lib.pangnn_linear_act_fwd_mixed(x.data_ptr(), 1, 64, w.data_ptr(), None, y.data_ptr(), 2, 64, 64, 64, 64, 0, None, 0, 0,
                                             _lib.stream_ptr())
        assert rc == -1 and b"bfloat16 or all float16" in lib.pangnn_last_error()
        g = torch.randn(64, 64, device=dev()).to(F16)
        ws_b = lib.pangnn_linear_wgrad_workspace_bytes(64, 64)
        ws = torch.empty(ws_b, dtype=torch.uint8, device=dev())
        gw = torch.empty(64, 64, device=dev())
        rc = lib.pangnn_linear_act_wgrad_mixed(g.data_ptr(), 2, 64, x.data_ptr(), 1, 64, 64, 64, 64, 0, gw.data_ptr(), None,
                                               ws.data_ptr(), ws_b, _lib.stream_ptr())
        assert rc == -1
    out = PF.linear(x, w, None, 0, F16)                         # the library route (torch), not an error
    assert out.dtype == F16 and close(out, x.float() @ w.t(), atol=2e-2, rtol=2e-3)


@pytest.mark.parametrize("F", [32, 64, 128, 256])
@pytest.mark.parametrize("n,e,hub", [(1, 5, None), (513, 7000, 3000), (2000, 30000, None)])
def test_spmm_f16_rows_match_oracle_on_rounded_inputs(F, n, e, hub):
    """float16 storage of the gathered rows, fp32 weights / accumulation / result: equal (to fp32 rounding) to the oracle on
    the float16-rounded features, and to the bfloat16-row kernel's sums bit for bit when the values fit both formats; the
    gradient w.r.t. the rows comes back in float16"""
    from pangnn_amd import functional as PF
    from pangnn_amd.graph import EdgeStructure
    ei, w = random_graph(n, e, seed=F + n, hub=hub)
    torch.manual_seed(F)
    x = torch.randn(n, F)
    xh = x.to(F16)
    b = torch.randn(F)
    norm_ref = go.gcn_norm(ei, w.double(), n, dtype=torch.float64)
    ref = go.propagate_add(xh.double(), ei, norm_ref) + b.double()
    st = EdgeStructure(ei.to(dev()), n)
    norm = st.gcn_norm(w.to(dev()))
    xd = xh.to(dev()).requires_grad_(True)
    out = PF.propagate(xd, b.to(dev()), st, norm)
    assert out.dtype == F32
    assert close(out, ref, atol=1e-4, rtol=1e-4)
    g = torch.randn(n, F)
    out.backward(g.to(dev()))
    assert xd.grad.dtype == F16
    gref = go.propagate_add(g.double(), ei.flip(0), norm_ref)
    assert close(xd.grad.float(), gref, atol=2e-3 * (float(gref.abs().max()) + 1e-12), rtol=2e-3)
    # values representable in both 2-byte formats (8 significant bits, small exponents): the two kernels differ in the
    # conversion instruction only
    xq = x.to(BF16).to(F16)
    if bool((xq.float() == x.to(BF16).float()).all()):
        a = PF.spmm_csr(st.by_dst, norm.by_dst, xq.to(dev()), n)
        c = PF.spmm_csr(st.by_dst, norm.by_dst, x.to(BF16).to(dev()), n)
        assert torch.equal(a, c)


@pytest.mark.parametrize("F", [64, 128])
def test_band_propagate_reads_f16_rows_exactly(F):
    """pangnn_band_propagate with x_dtype = PANGNN_DTYPE_F16 == the same kernel on the up-converted rows, bit for bit"""
    from pangnn_amd import _lib
    lib = _lib.load()
    n, k = 5000, 1
    gen = torch.Generator().manual_seed(F)
    x = torch.randn(n, F, generator=gen).to(dev()).to(F16)
    dis = (torch.rand(n, generator=gen) + 0.5).to(dev())
    bias = torch.randn(F, generator=gen).to(dev())
    outs = []
    for xs, code in ((x, 2), (x.float(), 0)):
        out = torch.full((n, F), float("nan"), device=dev())
        cs = torch.empty(F, device=dev())
        wsb = lib.pangnn_band_propagate_workspace_bytes(F)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
        with torch.cuda.device(dev()):
            _lib.check(lib.pangnn_band_propagate(xs.data_ptr(), code, xs.stride(0), dis.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                                 F, n, F, k, cs.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr()), "band")
        outs.append((out, cs))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("F", [64, 128])
@pytest.mark.parametrize("n", [1, 5, 1000, 300007])
def test_rank2_rows_and_column_sums_in_f16(F, n):
    """pangnn_rank2_rows / pangnn_embed_conv_in_rows store float16 = the f32 rows rounded once; pangnn_weighted_colsum3 and
    pangnn_colsum_small read float16 rows exactly (== the f32 kernels on the up-converted matrix, bit for bit)"""
    from pangnn_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(F + n)
    r, s_ = torch.randn(n, generator=gen).to(dev()), (torch.rand(n, generator=gen) + 0.5).to(dev())
    a, c, b = (torch.randn(F, generator=gen).to(dev()) for _ in range(3))
    o16 = torch.full((n, F), float("nan"), dtype=F16, device=dev())
    o32 = torch.full((n, F), float("nan"), device=dev())
    with torch.cuda.device(dev()):
        for o, code in ((o16, 2), (o32, 0)):
            _lib.check(lib.pangnn_rank2_rows(r.data_ptr(), s_.data_ptr(), a.data_ptr(), c.data_ptr(), b.data_ptr(), o.data_ptr(),
                                             code, F, n, F, _lib.stream_ptr()), "rank2_rows")
    assert torch.equal(o16, o32.to(F16))
    g = torch.randn(n, F, generator=gen).to(dev()).to(F16)
    sums = []
    wsb = lib.pangnn_weighted_colsum3_workspace_bytes(F)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
    with torch.cuda.device(dev()):
        for gs, code in ((g, 2), (g.float(), 0)):
            out = torch.empty(3, F, device=dev())
            _lib.check(lib.pangnn_weighted_colsum3(gs.data_ptr(), code, gs.stride(0), r.data_ptr(), s_.data_ptr(), n, F,
                                                   out.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr()), "colsum3")
            sums.append(out)
    assert torch.equal(sums[0], sums[1])
    if n <= 4096:
        cs = []
        with torch.cuda.device(dev()):
            for gs, code in ((g, 2), (g.float(), 0)):
                out = torch.empty(F, device=dev())
                _lib.check(lib.pangnn_colsum_small(gs.data_ptr(), code, gs.stride(0), n, F, out.data_ptr(), _lib.stream_ptr()),
                           "colsum_small")
                cs.append(out)
        assert torch.equal(cs[0], cs[1])


def test_propagate_with_f16_output_rounds_once_and_its_backward_gathers_f16_rows():
    from pangnn_amd import functional as PF
    from pangnn_amd.graph import EdgeStructure
    n = 3000
    ei, w = random_graph(n, 40000, seed=5)
    st = EdgeStructure(ei.to(dev()), n)
    norm = st.gcn_norm(w.to(dev()))
    torch.manual_seed(0)
    x0 = torch.randn(n, 64, device=dev()).to(F16)
    g0 = torch.randn(n, 64, device=dev()).to(F16)
    res = {}
    for od in (None, F16):
        x = x0.clone().requires_grad_(True)
        y = PF.propagate_any(x, None, st, norm, False, out_dtype=od)
        y.backward(g0 if od is not None else g0.float())
        res[od] = (y.detach(), x.grad)
    assert res[F16][0].dtype == F16 and res[None][0].dtype == F32
    assert torch.equal(res[F16][0], res[None][0].to(F16))
    assert res[F16][1].dtype == res[None][1].dtype == F16
    assert close(res[F16][1].float(), res[None][1].float(), atol=2e-3 * float(res[None][1].float().abs().max()), rtol=2e-3)


@pytest.mark.parametrize("flags", [dict(), dict(union_edge_weights=True), dict(base_model=True), dict(skip_connections=True)],
                         ids=["default", "union", "base", "skip"])
def test_model_under_fp16_autocast_stores_linear_outputs_as_f16(flags):
    """Under float16 autocast the reference's Linear layers return float16 tensors (src/gnn.py:93,111 with accelerate's fp16
    mixed precision): the rows conv_in writes, the x W^T rows that are propagated, the decoder's P | Q — WRITTEN as float16 by
    the kernels that produce them and read as stored by the next one (linear / propagate / decoder gather).  Logits and
    gradients sit at float16 resolution of the fp32 run — nearer to it than the oracle under float16 CPU autocast (the
    reference's own arithmetic) is."""
    import pangnn_amd
    from torch.utils._python_dispatch import TorchDispatchMode
    g = whole_graph_from_golden("cfg2_sim_1000x5")
    if flags.get("union_edge_weights"):
        g.edge_attr = g.union_edge_attr
    gd = copy_graph(g, "cuda")
    torch.manual_seed(3)
    oracle = go.AlternateGCNOracle(dims=(64, 128), flags=go.default_flags(**flags), categorical_nodes=False,
                                   num_nodes=g.x.shape[0])
    with torch.no_grad():
        for k, p in oracle.named_parameters():
            if k.endswith("bias"):
                p.uniform_(-0.5, 0.5)
    model = pangnn_amd.AlternateGCN("cuda", None, False, dims=[64, 128], **flags)
    model.load_state_dict(oracle.state_dict())
    seen, first = [], []

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            if func is torch.ops.pangnn.linear.default:
                seen.append((args[0].dtype, tuple(args[1].shape), out.dtype))
            if func is torch.ops.pangnn.embed_conv_in.default:
                first.append(out.dtype)
            return out
    with Spy(), torch.autocast("cuda", dtype=F16):
        loss, logits = model.loss_and_logits(gd, gd.y, None)
        loss.backward()
    assert first == [F16]
    if not flags.get("base_model"):
        assert any(o == F16 for _, _, o in seen), seen            # a GCNConv's dense part wrote float16 rows
    assert (F32, (128, 64), F16) in seen                          # P | Q = z [W_a ; W_b]^T: fp32 z in, float16 tables out
    assert logits.dtype == F32 and all(p.grad is None or p.grad.dtype == F32 for p in model.parameters())
    grads16 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    model.zero_grad()
    loss32, logits32 = model.loss_and_logits(gd, gd.y, None)
    loss32.backward()
    scale = float(logits32.abs().max())
    err = float((logits - logits32).abs().max())
    assert 0 < err < 4e-3 * scale, (err, scale)                    # float16 rows are on, and at float16 resolution
    assert abs(float(loss) - float(loss32)) < 2e-3 * abs(float(loss32))
    for k, p in model.named_parameters():
        if p.grad is not None:
            gs = float(p.grad.abs().max()) + 1e-12
            assert float((grads16[k] - p.grad).abs().max()) < 2e-2 * gs, k
    with torch.autocast("cpu", dtype=F16):
        try:
            ref = oracle(g).float()
        except RuntimeError:
            ref = None                                             # CPU float16 autocast lacks a kernel on this torch build
    if ref is not None:
        exact = oracle(g)
        assert float((logits.cpu() - exact).abs().max()) <= float((ref - exact).abs().max()) + 1e-3 * scale


@pytest.mark.parametrize("e", [1, 33, 1000, 70001])
@pytest.mark.parametrize("skip", [False, True])
def test_decoder_on_f16_tables_equals_decoder_on_upconverted_tables(e, skip):
    """pangnn_decoder_train_mixed / pangnn_decoder_mlp_infer_mixed with float16 P | Q (the S and inference kernels compiled for
    IEEE-half tables, decoder16_f16.o): the gather reads half the bytes, the arithmetic is unchanged — logits, loss and every
    gradient are bit-identical to the f32 entry points on the up-converted tables (sorted lists: run-sum path; unsorted:
    generic path); the table's gradient comes back as float16 = the f32 gradient rounded once"""
    from pangnn_amd import functional as PF
    from pangnn_amd.graph import EdgeStructure
    torch.manual_seed(e + skip)
    n, d = 97, 64
    ei, w = random_graph(n, e, seed=e, isolated=0.0)
    ei = ei[:, torch.argsort(ei[0] * n + ei[1])] if e % 2 else ei
    pq = torch.randn(n, 2 * d, device=dev()).to(F16)
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
    assert g16.dtype == F16 and torch.equal(g16, g32.to(F16))
    for a, b in zip(gw16, gw32):
        assert torch.equal(a, b)
    p16, q16 = pq[:, :d].contiguous(), pq[:, d:].contiguous()
    lsep, lgsep = PF.decoder_loss(p16, q16, st, extra, cv if skip else None, W2, b2, w3, b3, y, pw, e)
    assert torch.equal(lgsep, lg16) and torch.equal(lsep, l16)
    # the bfloat16 tables of the same values (8 significant bits fit both formats) give the same bits: only the conversion differs
    pq8 = pq.to(BF16)
    if bool((pq8.to(F16).float() == pq8.float()).all()):
        with torch.no_grad():
            a = PF.decoder_mlp_pq(pq8, st, extra, cv if skip else None, W2, b2, w3, b3)
            b = PF.decoder_mlp_pq(pq8.to(F16), st, extra, cv if skip else None, W2, b2, w3, b3)
        assert torch.equal(a, b)


def test_by_source_sums_over_row_windows_equal_the_rows_of_the_full_sum():
    """functional._decoder_train16(p_windows=...) (the partitioned decoder: an edge range whose sources lie in known row ranges):
    the windows' sums are the corresponding rows of the full dL/dP, nothing is written in between; out_logits / out_q +
    accumulate_q write where they are told"""
    from pangnn_amd import functional as PF
    from pangnn_amd.graph import EdgeStructure
    torch.manual_seed(0)
    n, d, e = 400, 64, 30000
    src = torch.sort(torch.randint(100, 300, (e,))).values            # sources only in rows [100, 300): source-sorted
    dst = torch.randint(0, n, (e,))
    ei = torch.stack([src, dst]).to(dev())
    st = EdgeStructure(ei, n, n)
    p, q = torch.randn(n, d, device=dev()), torch.randn(n, d, device=dev())
    w2, b2, w3, b3 = (torch.randn(d, d, device=dev()) / 8, torch.randn(d, device=dev()), torch.randn(d, device=dev()),
                      torch.randn(1, device=dev()))
    y = (torch.rand(e, device=dev()) < 0.3).float()
    pw = torch.tensor([2.0], device=dev())
    full = PF._decoder_train16(p, q, st, None, None, w2, b2, w3, b3, y=y, pw=pw, denom=e)
    lo_w, hi_w = torch.full((50, d), 7.0, device=dev()), torch.full((120, d), 7.0, device=dev())
    logits = torch.full((e + 8,), 9.0, device=dev())
    gq0 = torch.randn(n, d, device=dev())
    gq = gq0.clone()
    win = PF._decoder_train16(p, q, st, None, None, w2, b2, w3, b3, y=y, pw=pw, denom=e,
                              p_windows=[(100, 150, lo_w), (180, 300, hi_w)], out_logits=logits[4:e + 4], out_q=gq, accumulate_q=True)
    assert isinstance(win[2], list) and win[2][0] is lo_w and win[2][1] is hi_w
    # (a row's parts are added by whichever propagate kernel the entry density of the call selects — wave per row or lane group
    # per row —, so a window and the full call may associate a multi-part row's few terms differently: fp32 rounding apart)
    scale = float(full[2].abs().max())
    assert close(lo_w, full[2][100:150], atol=1e-6 * scale, rtol=1e-5) and close(hi_w, full[2][180:300], atol=1e-6 * scale, rtol=1e-5)
    assert torch.equal(logits[4:e + 4], full[1]) and bool((logits[:4] == 9.0).all()) and bool((logits[e + 4:] == 9.0).all())
    assert torch.equal(win[0], full[0])
    assert torch.allclose(gq, gq0 + full[3], atol=1e-6, rtol=1e-6)
    assert bool((full[2][:100] == 0).all()) and bool((full[2][300:] == 0).all())      # rows without edges: zero in the full sum
