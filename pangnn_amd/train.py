"""The link-prediction train step around the model (pangnn.py:88-98,194-216): zero_grad ->
forward -> BCEWithLogits(pos_weight) -> backward -> Adam.  Host side is plain PyTorch; every
gather/scatter inside `model` runs in libpangnn_hip.so."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import functional as PF


def criterion(logits, labels, pos_weight):
    """torch.nn.BCEWithLogitsLoss(pos_weight=class_balance) (pangnn.py:98) — loss and dL/dlogits from one
    HIP pass on the GPU"""
    if logits.is_cuda:
        return PF.bce_with_logits(logits, labels, pos_weight)
    return F.binary_cross_entropy_with_logits(logits, labels, pos_weight=pos_weight)


def make_optimizer(model, lr: float = 1e-3):
    return torch.optim.Adam(model.parameters(), lr=lr)      # pangnn.py:88


def train_step(model, optimizer, graph, labels, pos_weight):
    """One step; returns (loss, logits) as device tensors without synchronising."""
    optimizer.zero_grad(set_to_none=True)
    out = model(graph)
    loss = criterion(out, labels, pos_weight)                                       # pangnn.py:98,203
    loss.backward()
    optimizer.step()
    return loss.detach(), out.detach()


@torch.no_grad()
def eval_step(model, graph, labels, pos_weight):
    out = model(graph)
    return criterion(out, labels, pos_weight), out
