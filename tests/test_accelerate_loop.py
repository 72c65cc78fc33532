"""The reference's own training loop, restated literally, on the GPU under a real `accelerate.Accelerator`
(/root/reference/pangnn.py:25 Accelerator(mixed_precision), :87-98 model / Adam / BCEWithLogitsLoss(pos_weight = HOST scalar),
:122,155 accelerator.prepare, :152-153 DataLoader(batch_size=32), :194-222 the step: zero_grad, output = model(batch),
loss = criterion(output, labels), accelerator.backward(loss), optimizer.step(), loss.item(), sigmoid(output.detach()),
:339-341 unwrap_model(model).state_dict()) against the same loop run by the CPU oracle.

What a maintainer gets after INTEGRATION.md §1's import swap is exactly this call shape.  `model(batch)` in training mode
returns a `pangnn_amd.DeferredLogits` handle; torch's own `BCEWithLogitsLoss` resolves it through the one-pass training decoder
(S + T kernels: logits, loss and every gradient in one sweep), so this loop runs the kernels of `loss_and_logits` — asserted
here bit for bit — without a line of it changing."""
import os

import pytest
import torch
from conftest import copy_graph, sub_graphs_from_golden, whole_graph_from_golden

from oracle import gcn_oracle as go

pytestmark = pytest.mark.gpu


def _accelerator(mixed):
    from accelerate import Accelerator
    from accelerate.state import AcceleratorState
    AcceleratorState._reset_state(True)
    return Accelerator(mixed_precision=mixed)


def _batches(name):
    """(list of CPU batches as the oracle sees them, list of pangnn_amd Data for the loader, class balance)"""
    from pangnn_amd.data import Data
    if name == "cfg2_whole":
        g = whole_graph_from_golden("cfg2_sim_1000x5")
        cb = float((g.y == 0).sum() / g.y.sum())
        return [g, g, g], None, cb
    subs = sub_graphs_from_golden(name, count=96)
    chunks = [subs[i:i + 32] for i in range(0, len(subs), 32)][:3]
    y = torch.cat([s.y for s in subs])
    cb = float((y == 0).sum() / y.sum())
    data = [Data(s.x, s.edge_index, s.edge_attr, s.y, neighbour_edge_index=s.neighbour_edge_index) for s in subs]
    return [go.collate(c) for c in chunks], data, cb


def _spy_on_row_storage(monkeypatch):
    """dtype of every tensor the dense layers / the first layer STORE (functional.linear / embed_conv_in results): the
    observer is a plain function wrapper — a dispatch mode would switch the model to its traceable plain-tensor route"""
    from pangnn_amd import functional as PF
    seen = []
    for fn in ("linear", "embed_conv_in"):
        orig = getattr(PF, fn)

        def wrap(*a, _orig=orig, _fn=fn, **k):
            out = _orig(*a, **k)
            seen.append((_fn, out.dtype))
            return out
        monkeypatch.setattr(PF, fn, wrap)
    return seen


@pytest.mark.parametrize("mixed", ["no", "bf16"])
@pytest.mark.parametrize("name", ["cfg1_2genomes", "cfg3_5genomes", "cfg2_whole"])
def test_reference_loop_under_accelerate_tracks_the_oracle_loop(name, mixed, monkeypatch):
    import pangnn_amd
    from pangnn_amd import DeferredLogits
    from pangnn_amd.data import DataLoader
    cpu_batches, data, class_balance = _batches(name)
    dims = [64, 64] if name == "cfg2_whole" else [64, 128]          # config 2 is quoted with hidden_dim = 64

    # ---- the oracle's loop (CPU; under bf16: its forward inside CPU autocast, as accelerate wraps only forward)
    torch.manual_seed(0)
    oracle = go.AlternateGCNOracle(dims=tuple(dims))
    init = {k: v.clone() for k, v in oracle.state_dict().items()}
    opt_o = torch.optim.Adam(oracle.parameters(), lr=0.001)
    pw = torch.tensor(class_balance)
    ref = []
    for b in cpu_batches:
        opt_o.zero_grad()
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=mixed == "bf16"):
            out = oracle(b)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(out.float(), b.y, pos_weight=pw)
        loss.backward()
        opt_o.step()
        ref.append((loss.item(), out.detach().float()))

    # ---- the reference's loop, literally
    accelerator = _accelerator(mixed)                                                           # pangnn.py:25
    device = accelerator.device
    assert device.type == "cuda"
    model = pangnn_amd.AlternateGCN(device=device, dataset=None, categorical_nodes=False, dims=dims)   # pangnn.py:87
    model.load_state_dict(init)
    optimizer = torch.optim.Adam(model.parameters(), lr=0.001)                                   # pangnn.py:88
    criterion = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor(class_balance))               # pangnn.py:98 (host scalar)
    if data is not None:
        loader = DataLoader(data, batch_size=32, shuffle=False, pin_memory=True)                 # pangnn.py:152
        model, optimizer, loader = accelerator.prepare(model, optimizer, loader)                 # pangnn.py:155
    else:
        model, optimizer = accelerator.prepare(model, optimizer)                                 # pangnn.py:122
        whole = copy_graph(cpu_batches[0], device)
        loader = [whole, whole, whole]
    seen = _spy_on_row_storage(monkeypatch) if mixed == "bf16" else None
    train_loss, got = 0.0, []
    for batch_num, batch in enumerate(loader):
        if batch_num >= 3:
            break
        model.train()                                                                            # pangnn.py:190
        labels = batch.y
        optimizer.zero_grad()                                                                    # pangnn.py:194
        output = model(batch)                                                                    # pangnn.py:200
        assert type(output) is DeferredLogits and output.pending
        loss = criterion(output, labels)                                                         # pangnn.py:203
        assert output.route == "fused"
        accelerator.backward(loss)                                                               # pangnn.py:207
        optimizer.step()                                                                         # pangnn.py:216
        train_loss += loss.item()                                                                # pangnn.py:218
        probabilities = torch.sigmoid(output.detach())                                           # pangnn.py:220
        assert probabilities.shape == labels.shape and output.route == "fused"
        got.append((loss.item(), output.detach().cpu()))
    state = accelerator.unwrap_model(model).state_dict()                                         # pangnn.py:339-341
    assert len(got) == len(ref) == 3

    if mixed == "no":
        # fp32: north_star's 1e-4 on the logits of the first step, loss 1e-5; later steps carry Adam's drift (lr 1e-3 times a
        # sign-like first update: an element whose gradient is rounding noise may step the other way), as
        # test_train_steps_track_the_oracle allows
        for step, ((lm, om), (lo, oo)) in enumerate(zip(got, ref)):
            tol = 1e-4 if step == 0 else 5e-4
            assert torch.allclose(om, oo, atol=tol, rtol=tol), (step, float((om - oo).abs().max()))
            assert abs(lm - lo) <= (1e-5 if step == 0 else 1e-4) * max(1.0, abs(lo)), (step, lm, lo)
        ref_state = oracle.state_dict()
        assert list(state.keys()) == list(ref_state.keys())
        worst, moved = 0.0, 0
        for k, v in state.items():
            d = (v.cpu() - ref_state[k]).abs()
            # after 3 Adam steps a parameter has moved by <= 3e-3; elements whose gradient is at rounding level may differ by
            # that much, everything else agrees to 2e-5
            assert float(d.max()) <= 6.5e-3, k
            worst = max(worst, float(d.max()))
            moved += int((d > 2e-5).sum())
        total = sum(v.numel() for v in state.values())
        assert moved <= 0.002 * total, (moved, total, worst)
    else:
        # bf16: the bounds of test_config5_edge_law_matches_autocast_oracle — logits at bf16 resolution of the oracle under CPU
        # autocast, the loss tracks the oracle's and goes down, and the rows really were STORED as bfloat16
        for step, ((lm, om), (lo, oo)) in enumerate(zip(got, ref)):
            scale = float(oo.abs().max())
            # step 0: bf16 resolution.  Later steps: the two sides round different bf16 elements, and Adam's first updates are
            # sign-like (lr * g / |g|), so a gradient element at rounding level moves its parameter the other way: the
            # logits drift apart by more than one rounding (measured 6.6e-2 of scale at step 1 on cfg 2) while the loss tracks
            assert float((om - oo).abs().max()) < (5e-2 if step == 0 else 2e-1) * scale, step
            assert abs(lm - lo) < 5e-2 * abs(ref[0][0]), (step, lm, lo)
        if name == "cfg2_whole":
            assert got[-1][0] < got[0][0]
        bf = torch.bfloat16
        assert ("embed_conv_in", bf) in seen or ("linear", bf) in seen
        assert sum(1 for f, d in seen if f == "linear" and d == bf) >= 3 * 2        # conv_out's dense part and P|Q, every step
        assert all(v.dtype == torch.float32 for v in state.values())
    from accelerate.state import AcceleratorState
    AcceleratorState._reset_state(True)


@pytest.mark.parametrize("flags", [dict(), dict(skip_connections=True), dict(base_model=True), dict(union_edge_weights=True)],
                         ids=["default", "skip", "base", "union"])
def test_deferred_route_is_loss_and_logits_bit_for_bit(flags):
    """model(g) -> torch BCEWithLogitsLoss -> loss.backward() runs the kernels of model.loss_and_logits(g, y, pw) +
    backward(unit_grad): loss, logits and every gradient are bitwise equal (the upstream gradient 1.0 is found on the device
    by pangnn_scale_unless_one_f32, which then touches nothing)."""
    import pangnn_amd
    from pangnn_amd import functional as PF
    g = whole_graph_from_golden("cfg3_5genomes")
    if flags.get("union_edge_weights"):
        g.edge_attr = g.union_edge_attr
    gd = copy_graph(g, "cuda")
    torch.manual_seed(1)
    model = pangnn_amd.AlternateGCN("cuda", None, False, dims=[64, 128], **flags)
    pw_host = torch.tensor(float((g.y == 0).sum() / g.y.sum()))
    loss, logits = model.loss_and_logits(gd, gd.y, pw_host.cuda())
    loss.backward(PF.unit_grad(loss.device))
    want = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    model.zero_grad()
    out = model(gd)
    assert type(out) is pangnn_amd.DeferredLogits
    loss2 = torch.nn.BCEWithLogitsLoss(pos_weight=pw_host)(out, gd.y)
    loss2.backward()
    assert torch.equal(loss2.detach(), loss.detach()) and torch.equal(out.detach(), logits)
    got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert got.keys() == want.keys()
    for k in want:
        assert torch.equal(got[k], want[k]), k


def test_scaled_loss_scales_every_gradient_and_second_backward_is_refused():
    """`(loss * s).backward()` (a GradScaler's scale, gradient accumulation): the stored gradients are multiplied in place by
    the device scalar; backward through the same fused loss twice raises instead of scaling twice"""
    import pangnn_amd
    g = copy_graph(whole_graph_from_golden("cfg1_2genomes"), "cuda")
    torch.manual_seed(2)
    model = pangnn_amd.AlternateGCN("cuda", None, False, dims=[64, 128], skip_connections=True)
    crit = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor(4.0))
    crit(model(g), g.y).backward()
    base = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    model.zero_grad()
    loss = crit(model(g), g.y)
    (loss * 1024.0).backward(retain_graph=True)
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert torch.allclose(p.grad, base[k] * 1024.0, rtol=1e-6, atol=0), k
    with pytest.raises(RuntimeError, match="second time"):
        (loss * 2.0).backward()


def test_scale_unless_one_entry_point():
    """pangnn_scale_unless_one_f32 through the C ABI: exact 1.0 leaves every buffer untouched (bitwise), any other scalar
    multiplies each in place, odd lengths / unaligned starts included"""
    import ctypes as C
    from pangnn_amd import _lib
    lib = _lib.load()
    torch.manual_seed(0)
    base = torch.randn(1 << 20, device="cuda")
    bufs = [base[:1000003].clone(), base[1:130].clone()[1:], base[:0].clone(), base[:64 * 128].clone().view(64, 128), base[:1].clone()]
    live = [b for b in bufs if b.numel()]
    for s in (1.0, 0.5, -3.0):
        work = [b.clone() for b in live]
        ptrs = (C.c_void_p * len(work))(*[w.data_ptr() for w in work])
        counts = (C.c_int64 * len(work))(*[w.numel() for w in work])
        sc = torch.tensor(s, device="cuda")
        _lib.check(lib.pangnn_scale_unless_one_f32(ptrs, counts, len(work), sc.data_ptr(), _lib.stream_ptr()))
        torch.cuda.synchronize()
        for w, b in zip(work, live):
            assert torch.equal(w, b * s)
    assert lib.pangnn_scale_unless_one_f32(None, None, 9, None, None) == -1          # PANGNN_E_BADARG: more than 8 tensors


def test_forward_returns_plain_tensors_where_the_handle_does_not_apply():
    import pangnn_amd
    g = copy_graph(whole_graph_from_golden("cfg1_2genomes"), "cuda")
    model = pangnn_amd.AlternateGCN("cuda", None, False, dims=[64, 128])
    model.eval()
    assert type(model(g)) is torch.Tensor                                            # validation loop, pangnn.py:243-249
    model.train()
    with torch.no_grad():
        assert type(model(g)) is torch.Tensor
    assert type(pangnn_amd.AlternateGCN("cuda", None, False, dims=[64, 128], decoder="cosine")(g)) is torch.Tensor
    assert type(pangnn_amd.AlternateGCN("cuda", None, False, dims=[32, 64])(g)) is torch.Tensor      # node_dim != 64
    assert type(pangnn_amd.AlternateGCN("cuda", None, False, dims=[64, 128], deferred_logits=False)(g)) is torch.Tensor
    os.environ["PANGNN_DEFERRED_LOGITS"] = "0"
    try:
        assert type(pangnn_amd.AlternateGCN("cuda", None, False, dims=[64, 128])(g)) is torch.Tensor
    finally:
        del os.environ["PANGNN_DEFERRED_LOGITS"]


def test_padded_fixed_shape_batch_reaches_the_fused_pass_through_the_handle():
    """train.ReplayedFreshStep's padded buffers carry `live_edges`: forward + torch's criterion is the fused pass there too,
    and a use that would need the inference kernel on a padded batch is refused instead of scoring the padding"""
    import pangnn_amd
    from pangnn_amd.subgraphs import SubGraphDataset
    subs = sub_graphs_from_golden("cfg1_2genomes", count=64)
    ds = SubGraphDataset.from_data_list(subs, device="cuda")
    spec = ds.padded_spec(32)
    buf = ds.padded_buffers(spec)
    ds.set_graph_ids(buf, list(range(32)))
    ds.collate_padded(buf)
    torch.manual_seed(0)
    model = pangnn_amd.AlternateGCN("cuda", None, False, dims=[64, 128])
    pw = torch.tensor(3.0, device="cuda")
    loss, logits = model.loss_and_logits(buf, buf.y, pw)
    out = model(buf)
    loss2 = torch.nn.BCEWithLogitsLoss(pos_weight=pw)(out, buf.y)
    assert torch.equal(loss2.detach(), loss.detach()) and torch.equal(out.detach(), logits)
    with pytest.raises(NotImplementedError):
        torch.sigmoid(model(buf))


def test_reference_loop_under_accelerate_fp16_tracks_the_fp32_oracle_at_f16_resolution():
    """`--mixed_precision fp16` (/root/reference/src/setup.py:50): accelerate wraps `forward` in float16 autocast and drives
    `accelerator.backward(loss)` through a GradScaler (loss x 65 536).  The encoder's Linear outputs / propagated rows are stored
    as float16 (PANGNN_DTYPE_F16, round 5 — tests/test_f16_rows.py), every product and sum and the decoder's P | Q tables stay
    fp32, so the loop tracks the FP32 oracle at float16 resolution — nearer than the reference's own fp16 arithmetic would —
    and the scaled loss exercises the in-place gradient scaling behind the fused decoder pass (pangnn_scale_unless_one_f32 with a
    scalar that is NOT 1), unscaled again by the scaler before Adam; float16 gradient rows carry the 65 536 x scale, as in the
    reference."""
    import pangnn_amd
    from pangnn_amd import DeferredLogits
    g = whole_graph_from_golden("cfg2_sim_1000x5")
    class_balance = float((g.y == 0).sum() / g.y.sum())
    torch.manual_seed(0)
    oracle = go.AlternateGCNOracle(dims=(64, 64))
    init = {k: v.clone() for k, v in oracle.state_dict().items()}
    opt_o = torch.optim.Adam(oracle.parameters(), lr=0.001)
    ref = []
    for _ in range(3):
        lo, out_o = go.train_step(oracle, opt_o, g, g.y, torch.tensor(class_balance))
        ref.append((float(lo), out_o))
    accelerator = _accelerator("fp16")
    assert accelerator.scaler is not None
    model = pangnn_amd.AlternateGCN(device=accelerator.device, dataset=None, categorical_nodes=False, dims=[64, 64])
    model.load_state_dict(init)
    optimizer = torch.optim.Adam(model.parameters(), lr=0.001)
    criterion = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor(class_balance))
    model, optimizer = accelerator.prepare(model, optimizer)
    batch = copy_graph(g, accelerator.device)
    for step in range(3):
        model.train()
        optimizer.zero_grad()
        output = model(batch)
        assert type(output) is DeferredLogits
        loss = criterion(output, batch.y)
        accelerator.backward(loss)                      # scaler.scale(loss).backward(): upstream gradient = the scale
        optimizer.step()                                # scaler.step: unscale, inf check, Adam; scaler.update
        assert output.route == "fused" and not optimizer.step_was_skipped
        scale = float(ref[step][1].abs().max())
        err = float((output.detach().cpu() - ref[step][1]).abs().max())
        print(f"fp16 loop step {step}: max |logit - fp32 oracle| = {err:.3e} (scale {scale:.3e}), "
              f"loss {loss.item():.6f} vs {ref[step][0]:.6f}")
        assert err < (4e-3 if step == 0 else 2e-2) * scale, (step, err, scale)          # float16 rows: 2^-11 per element
        assert abs(loss.item() - ref[step][0]) <= (2e-3 if step == 0 else 1e-2) * max(1.0, abs(ref[step][0]))
    state = accelerator.unwrap_model(model).state_dict()
    assert all(v.dtype == torch.float32 for v in state.values())
    diffs = torch.cat([(v.cpu() - oracle.state_dict()[k]).abs().flatten() for k, v in state.items()])
    worst, off = float(diffs.max()), float((diffs > 1e-4).float().mean())
    print(f"fp16 loop: max |parameter - fp32 oracle| after 3 Adam steps = {worst:.3e}; {100 * off:.2f} % of the entries off by > 1e-4")
    # Adam's normalised step moves an entry by ~lr per step whatever the gradient's size: an entry whose tiny gradient changes sign
    # under float16 rounding ends up to 2 lr x 3 steps away — a few entries, bounded in number and in distance
    assert worst < 6.5e-3 and off < 0.05
    from accelerate.state import AcceleratorState
    AcceleratorState._reset_state(True)
