"""`DeferredLogits` — what `AlternateGCN.forward` returns in training mode, so that the reference's own loop

    output = model(batch)                       # pangnn.py:200
    loss = criterion(output, labels)            # pangnn.py:203   torch.nn.BCEWithLogitsLoss(pos_weight=...)  (pangnn.py:98)
    accelerator.backward(loss)                  # pangnn.py:207
    ... torch.sigmoid(output.detach()) ...      # pangnn.py:220

reaches the ONE-PASS training decoder (logits + loss + every gradient in one sweep over the edges, csrc/decoder16.hip S + T)
without a line of that loop changing.  When `forward` runs the labels are not known yet, so the per-edge decoder is not
launched there: `forward` runs the encoder and the node-level half of `mlp[0]` (under accelerate's autocast wrapper, which
only covers `forward`) and returns this handle — a `torch.Tensor` subclass with the logits' shape / dtype / device and no
storage.  What happens next decides which kernels run:

  * `torch.nn.functional.binary_cross_entropy_with_logits(handle, labels, pos_weight=..., reduction='mean')` — what
    `torch.nn.BCEWithLogitsLoss.forward` calls — is intercepted through `__torch_function__` and resolved by the fused pass;
    the handle then holds that pass's logits, so the loop's later `output.detach()` costs nothing.  The same for
    `pangnn_amd.train.criterion` and `pangnn_amd.BCEWithLogitsLoss`.
  * ANY other first use (`torch.sigmoid(output)`, `output.cpu()`, arithmetic, another loss, a weighted / unreduced BCE)
    materialises the logits through the inference decoder kernel with its ordinary autograd node (the literal
    forward → criterion → backward route), and from then on the handle is that tensor.

So the handle is never wrong, only faster when the loop has the reference's shape.  Metadata (`shape`, `dtype`, `device`,
`requires_grad`, `len()`, `size()`, `dim()`, `numel()`) is answered without launching anything; accelerate's
`convert_outputs_to_fp32` (it asks for `.dtype`) passes the handle through untouched.

The handle itself knows nothing about graphs or kernels: it is built from two thunks (`materialize()` -> differentiable
logits, `fused_loss(labels, pos_weight)` -> (loss, detached logits)), which is also how tests/test_deferred.py exercises the
protocol on CPU tensors.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn.functional as F
from torch.utils._pytree import tree_map

_DEVICE_SCALARS = {}


def scalar_on(t: Optional[torch.Tensor], device) -> Optional[torch.Tensor]:
    """`t` on `device`.  The reference builds its criterion with a CPU scalar — `torch.tensor(hparams['class_balance'])`
    (pangnn.py:98), legal for torch's elementwise BCE, not for a kernel argument — so the device copy is made once per
    (tensor, version) and reused; a tensor already on the device passes through."""
    if t is None or t.device == torch.device(device):
        return t
    key = (id(t), t._version, str(device))
    hit = _DEVICE_SCALARS.get(key)
    if hit is None or hit[0] is not t:
        if len(_DEVICE_SCALARS) > 64:
            _DEVICE_SCALARS.clear()
        hit = _DEVICE_SCALARS[key] = (t, t.detach().to(device=device, dtype=torch.float32))
    return hit[1]


class DeferredLogits(torch.Tensor):
    @staticmethod
    def __new__(cls, num_edges: int, device, materialize: Callable, fused_loss: Optional[Callable]):
        return torch.Tensor._make_wrapper_subclass(cls, (int(num_edges),), dtype=torch.float32, device=device,
                                                   requires_grad=True)

    def __init__(self, num_edges: int, device, materialize: Callable, fused_loss: Optional[Callable]):
        self._materialize = materialize      # () -> logits [E] with their autograd node (inference decoder kernel)
        self._fused_loss = fused_loss        # (labels, pos_weight) -> (loss with its autograd node, detached logits)
        self._diff = None                    # the materialised, differentiable logits
        self._values = None                  # the fused pass's logits (no graph: the fused loss owns the gradients)
        self.route = None                    # "fused" / "materialized": which kernels answered (tests, bench)

    # ------------------------------------------------------------------ answered without launching anything
    @property
    def pending(self) -> bool:
        return self._diff is None and self._values is None

    def __repr__(self, *, tensor_contents=None):
        state = "pending" if self.pending else self.route
        with torch._C.DisableTorchFunctionSubclass():
            return f"DeferredLogits(num_edges={self.size(0)}, device={self.device}, {state})"

    def __len__(self):
        with torch._C.DisableTorchFunctionSubclass():
            return self.size(0)

    # ------------------------------------------------------------------ resolution
    def materialize(self) -> torch.Tensor:
        """the differentiable logits (inference decoder kernel + its autograd node), computed once"""
        if self._diff is None:
            with torch.enable_grad():
                self._diff = self._materialize()
            if self.route is None:
                self.route = "materialized"
        return self._diff

    def _plain(self, differentiable: bool) -> torch.Tensor:
        if not differentiable:
            if self._values is not None:
                return self._values
            return self.materialize().detach()
        return self.materialize()

    def fused_bce(self, target, pos_weight=None):
        """mean BCEWithLogits(pos_weight) of these logits as ONE decoder pass; None when that does not apply (already
        resolved, no fused decoder behind the handle, no gradient wanted, or labels of another shape / device)"""
        if not self.pending or self._fused_loss is None or not torch.is_grad_enabled():
            return None
        with torch._C.DisableTorchFunctionSubclass():
            n, dev = self.size(0), self.device
        if not torch.is_tensor(target) or isinstance(target, DeferredLogits) or target.device != dev \
                or target.dim() != 1 or target.shape[0] != n:
            return None
        if pos_weight is not None and (not torch.is_tensor(pos_weight) or pos_weight.numel() != 1):
            return None
        loss, logits = self._fused_loss(target, scalar_on(pos_weight, dev))
        self._values, self.route = logits.detach(), "fused"
        return loss

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # every Python-visible use is unwrapped in __torch_function__ (above autograd, where the materialised logits' graph
        # can still be attached); something that reaches the dispatcher with the handle itself bypassed that, and running
        # it here — below autograd — would silently cut the gradient
        raise RuntimeError(f"pangnn_amd.DeferredLogits reached the dispatcher in {func} without passing __torch_function__; "
                           f"call `.materialize()` on the model's output first (or build the model with deferred_logits=False)")

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _METADATA:
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        if func is F.binary_cross_entropy_with_logits:
            loss = _bce_through_fused_pass(*args, **kwargs)
            if loss is not None:
                return loss
        differentiable = torch.is_grad_enabled() and func not in _DETACHING

        def plain(a):
            return a._plain(differentiable) if isinstance(a, DeferredLogits) else a

        with torch._C.DisableTorchFunctionSubclass():
            return func(*tree_map(plain, args), **tree_map(plain, kwargs))


def _bce_through_fused_pass(input, target, weight=None, size_average=None, reduce=None, reduction="mean", pos_weight=None):
    if not isinstance(input, DeferredLogits) or weight is not None or size_average is not None or reduce is not None \
            or reduction != "mean":
        return None
    return input.fused_bce(target, pos_weight)


_T = torch.Tensor
_METADATA = {
    _T.shape.__get__, _T.dtype.__get__, _T.device.__get__, _T.requires_grad.__get__, _T.ndim.__get__, _T.is_cuda.__get__,
    _T.is_cpu.__get__, _T.layout.__get__, _T.is_sparse.__get__, _T.is_quantized.__get__, _T.is_meta.__get__,
    _T.grad_fn.__get__, _T.is_leaf.__get__, _T.names.__get__,
    _T.size, _T.dim, _T.numel, _T.nelement, _T.is_floating_point, _T.is_complex, _T.get_device, _T.element_size,
    _T.__hash__, _T.is_contiguous, _T.stride, _T.storage_offset,
}
_DETACHING = {_T.detach, _T.data.__get__, _T.item, _T.tolist, _T.numpy, _T.__array__}


class BCEWithLogitsLoss(torch.nn.BCEWithLogitsLoss):
    """`torch.nn.BCEWithLogitsLoss` (pangnn.py:98) whose `pos_weight` may sit on the host like the reference's: on
    `DeferredLogits` it is the fused decoder pass (as torch's own class is, through `__torch_function__`); on plain device
    logits with mean reduction and no per-element weight it is the HIP loss kernel (loss + dL/dlogits in one pass,
    `pangnn::bce_with_logits`) instead of torch's three elementwise passes; anything else is torch's."""

    def forward(self, input, target):
        if isinstance(input, DeferredLogits):
            return F.binary_cross_entropy_with_logits(input, target, self.weight, pos_weight=self.pos_weight,
                                                      reduction=self.reduction)
        if self.weight is None and self.reduction == "mean" and input.is_cuda and input.dim() == 1 \
                and (self.pos_weight is None or self.pos_weight.numel() == 1):
            from . import functional as PF
            return PF.bce_with_logits(input, target.to(input.dtype), scalar_on(self.pos_weight, input.device))
        return super().forward(input, target)
