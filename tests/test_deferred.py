"""The handle protocol of `pangnn_amd.deferred.DeferredLogits` on CPU tensors (no kernel behind it: the two thunks are plain
torch here) — which uses resolve the handle through the fused loss, which materialise it, what stays free — and the
reference loop's call shape (pangnn.py:25,98,122,194-222) under a real `accelerate.Accelerator` around a toy module that
hands the handle out.  The GPU legs (the model's own thunks, kernels, oracle parity) are tests/test_accelerate_loop.py."""
import pytest
import torch
import torch.nn.functional as F

from pangnn_amd.data import Batch, Data, DataLoader
from pangnn_amd.deferred import BCEWithLogitsLoss, DeferredLogits


class Toy(torch.nn.Module):
    """logits = 2 w: hands out a DeferredLogits in training mode like AlternateGCN.forward, counting which thunk ran"""

    def __init__(self, n=10):
        super().__init__()
        self.w = torch.nn.Parameter(torch.linspace(-1, 1, n))
        self.calls = []

    def forward(self, _batch=None):
        if not (self.training and torch.is_grad_enabled()):
            return 2.0 * self.w

        def materialize():
            self.calls.append("materialize")
            return 2.0 * self.w

        def fused(y, pw):
            self.calls.append("fused")
            x = 2.0 * self.w
            return F.binary_cross_entropy_with_logits(x, y, pos_weight=pw), x.detach()
        return DeferredLogits(self.w.shape[0], self.w.device, materialize, fused)


def _labels(n=10):
    return (torch.arange(n) % 3 == 0).float()


def test_metadata_is_free_and_looks_like_the_logits():
    m = Toy()
    h = m()
    assert isinstance(h, torch.Tensor) and type(h) is DeferredLogits and h.pending
    assert h.shape == (10,) and h.dtype == torch.float32 and h.device.type == "cpu" and h.requires_grad
    assert len(h) == 10 and h.dim() == 1 and h.numel() == 10 and h.size(0) == 10 and h.ndim == 1 and not h.is_cuda
    assert "pending" in repr(h)
    assert m.calls == [] and h.pending                     # nothing ran


def test_torch_bce_with_logits_loss_is_resolved_by_the_fused_pass():
    """pangnn.py:98,203: torch.nn.BCEWithLogitsLoss(pos_weight = CPU scalar)(output, labels) — not a line changed"""
    m, y = Toy(), _labels()
    crit = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor(3.0))
    out = m()
    loss = crit(out, y)
    assert m.calls == ["fused"] and out.route == "fused" and not out.pending
    ref = F.binary_cross_entropy_with_logits(2.0 * m.w, y, pos_weight=torch.tensor(3.0))
    assert torch.equal(loss, ref)
    loss.backward()
    g = m.w.grad.clone()
    m.w.grad = None
    ref.backward()
    assert torch.equal(g, m.w.grad)
    # pangnn.py:219-221: loss.item(), torch.sigmoid(output.detach()) — the fused pass's logits, nothing else runs
    p = torch.sigmoid(out.detach())
    assert m.calls == ["fused"] and torch.equal(p, torch.sigmoid(2.0 * m.w.detach()))
    with torch.no_grad():
        assert torch.equal((out >= 0).int(), (2.0 * m.w >= 0).int()) and m.calls == ["fused"]


def test_package_criterion_and_functional_form_take_the_fused_pass_too():
    from pangnn_amd.train import criterion
    m, y = Toy(), _labels()
    out = m()
    F.binary_cross_entropy_with_logits(out, y)
    assert m.calls == ["fused"]
    out = m()
    BCEWithLogitsLoss(pos_weight=torch.tensor([2.0]))(out, y)
    assert m.calls == ["fused", "fused"]
    out = m()
    criterion(out, y, torch.tensor(2.0))
    assert m.calls == ["fused"] * 3


@pytest.mark.parametrize("use", ["sigmoid", "weighted", "sum_reduction", "arith", "other_loss", "no_grad_bce", "vector_pw"])
def test_any_other_first_use_materialises_and_keeps_the_gradient(use):
    m, y = Toy(), _labels()
    out = m()
    x = 2.0 * m.w
    if use == "sigmoid":
        got, ref = torch.sigmoid(out).sum(), torch.sigmoid(x).sum()
    elif use == "weighted":
        w = torch.linspace(0.5, 1.5, 10)
        got, ref = F.binary_cross_entropy_with_logits(out, y, weight=w), F.binary_cross_entropy_with_logits(x, y, weight=w)
    elif use == "sum_reduction":
        got = torch.nn.BCEWithLogitsLoss(reduction="sum")(out, y)
        ref = F.binary_cross_entropy_with_logits(x, y, reduction="sum")
    elif use == "arith":
        got, ref = (out * 3 - 1).pow(2).mean(), (x * 3 - 1).pow(2).mean()
    elif use == "other_loss":
        got, ref = F.mse_loss(out, y), F.mse_loss(x, y)
    elif use == "vector_pw":
        pw = torch.linspace(1, 2, 10)
        got, ref = F.binary_cross_entropy_with_logits(out, y, pos_weight=pw), F.binary_cross_entropy_with_logits(x, y, pos_weight=pw)
    else:
        with torch.no_grad():                               # validation-style use of a training-mode output
            got = F.binary_cross_entropy_with_logits(out, y)
        assert m.calls == ["materialize"] and not got.requires_grad
        assert torch.equal(got, F.binary_cross_entropy_with_logits(x.detach(), y))
        got, ref = out.sum(), x.sum()                       # ... and the graph is still there afterwards
    assert m.calls == ["materialize"] and out.route == "materialized"
    assert torch.equal(got, ref)
    got.backward()
    g = m.w.grad.clone()
    m.w.grad = None
    ref.backward()
    assert torch.equal(g, m.w.grad)
    assert m.calls == ["materialize"]                       # materialised once, reused


def test_differentiable_use_after_the_fused_loss_materialises_instead_of_losing_the_gradient():
    m, y = Toy(), _labels()
    out = m()
    loss = torch.nn.BCEWithLogitsLoss()(out, y)
    extra = (out ** 2).mean()                               # a second, differentiable use: must not see detached logits
    assert m.calls == ["fused", "materialize"] and extra.requires_grad
    (loss + extra).backward()
    g = m.w.grad.clone()
    m.w.grad = None
    (F.binary_cross_entropy_with_logits(2.0 * m.w, y) + ((2.0 * m.w) ** 2).mean()).backward()
    assert torch.equal(g, m.w.grad)


def test_a_second_loss_on_a_resolved_handle_is_torchs_own():
    m, y = Toy(), _labels()
    out = m()
    crit = torch.nn.BCEWithLogitsLoss()
    a = crit(out, y)
    b = crit(out, y)                                        # resolved already: materialise + torch's BCE, same value
    assert m.calls == ["fused", "materialize"] and torch.equal(a, b)


def test_eval_mode_and_no_grad_return_plain_tensors():
    m = Toy()
    m.eval()
    assert type(m()) is torch.Tensor
    m.train()
    with torch.no_grad():
        assert type(m()) is torch.Tensor


def test_dispatcher_level_use_is_refused_loudly():
    h = Toy()()
    with pytest.raises(RuntimeError, match="materialize"):
        with torch._C.DisableTorchFunctionSubclass():
            h + 1                                           # bypasses __torch_function__: would cut the gradient silently


def test_accelerate_output_conversion_passes_the_handle_through():
    from accelerate.utils.operations import convert_to_fp32
    m = Toy()
    h = m()
    assert convert_to_fp32(h) is h and convert_to_fp32({"out": (h,)})["out"][0] is h and m.calls == []


@pytest.mark.parametrize("mixed", ["no", "bf16"])
def test_reference_loop_shape_under_accelerate_on_cpu(mixed):
    """pangnn.py:25,88,98,122,152-155,194-222 with a toy model: Accelerator, prepare(model, optimizer, loader), model(batch),
    torch's BCEWithLogitsLoss with a host pos_weight, accelerator.backward, optimizer.step, loss.item(), sigmoid of
    output.detach(), unwrap_model().state_dict() — the fused thunk answers every training step, the values are torch's."""
    from accelerate import Accelerator
    from accelerate.state import AcceleratorState
    AcceleratorState._reset_state(True)
    accelerator = Accelerator(cpu=True, mixed_precision=mixed)
    torch.manual_seed(0)
    model, ref = Toy(), Toy()
    optimizer, ref_opt = torch.optim.Adam(model.parameters(), lr=0.001), torch.optim.Adam(ref.parameters(), lr=0.001)
    criterion = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor(2.5))
    graphs = [Data(torch.ones(3, 1), torch.tensor([[0, 1], [1, 2]]), torch.ones(2), _labels(), neighbour_edge_index=torch.tensor([[0], [1]]))
              for _ in range(6)]
    loader = DataLoader(graphs, batch_size=2, shuffle=True)
    model, optimizer, loader = accelerator.prepare(model, optimizer, loader)
    steps = 0
    for batch in loader:
        assert isinstance(batch, Batch) and batch.num_graphs == 2
        model.train()
        labels = _labels()
        optimizer.zero_grad()
        output = model(batch)
        assert type(output) is DeferredLogits               # also through accelerate's autocast / fp32-conversion wrapper
        loss = criterion(output, labels)
        accelerator.backward(loss)
        optimizer.step()
        ref_opt.zero_grad()
        ref.eval()
        rl = F.binary_cross_entropy_with_logits(ref(None), labels, pos_weight=torch.tensor(2.5))
        rl.backward()
        ref_opt.step()
        assert abs(loss.item() - rl.item()) < 1e-6
        probabilities = torch.sigmoid(output.detach())
        assert probabilities.shape == (10,) and output.route == "fused"
        steps += 1
    inner = accelerator.unwrap_model(model)
    assert steps == 3 and inner.calls == ["fused"] * 3
    assert torch.allclose(inner.state_dict()["w"], ref.state_dict()["w"], atol=1e-6)
    AcceleratorState._reset_state(True)
