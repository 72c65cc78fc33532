"""The link-prediction train step around the model (pangnn.py:88-98,194-216): zero_grad ->
forward -> BCEWithLogits(pos_weight) -> backward -> Adam.  Host side is plain PyTorch; every
gather/scatter inside `model` runs in libpangnn_hip.so."""
from __future__ import annotations

import torch

from . import functional as PF


def criterion(logits, labels, pos_weight):
    """torch.nn.BCEWithLogitsLoss(pos_weight=class_balance) (pangnn.py:98) — loss and dL/dlogits from one
    HIP pass (device tensors only: there is no CPU path)"""
    return PF.bce_with_logits(logits, labels, pos_weight)


def make_optimizer(model, lr: float = 1e-3, capturable: bool = False, fused: bool = True):
    """torch.optim.Adam(lr) of pangnn.py:88.  `fused=True`: PyTorch's single-launch Adam over all parameter tensors
    (the same update rule; ~8 launches fewer per step than the default foreach form, which is what a launch-bound
    mini-batch step notices)."""
    params = list(model.parameters())
    fused = bool(fused) and all(p.is_cuda and p.dtype == torch.float32 for p in params)
    return torch.optim.Adam(params, lr=lr, capturable=capturable, fused=fused)


def train_step(model, optimizer, graph, labels, pos_weight):
    """One step; returns (loss, logits) as device tensors without synchronising."""
    for group in optimizer.param_groups:              # optimizer.zero_grad(set_to_none=True) without its dynamo-disable
        for p in group["params"]:                     # wrapper (~20 us of a launch-bound mini-batch step)
            p.grad = None
    if hasattr(model, "loss_and_logits"):
        loss, out = model.loss_and_logits(graph, labels, pos_weight)                # fused forward + criterion
    else:
        out = model(graph)
        loss = criterion(out, labels, pos_weight)                                   # pangnn.py:98,203
    loss.backward(PF.unit_grad(loss.device))                                        # = loss.backward(), see unit_grad
    optimizer.step()
    return loss.detach(), out.detach()


@torch.no_grad()
def eval_step(model, graph, labels, pos_weight):
    out = model(graph)
    return criterion(out, labels, pos_weight), out


@torch.no_grad()
def evaluate(model, batches, pos_weight, threshold: float = 0.5) -> dict:
    """The validation pass of pangnn.py:241-290: mean loss over the batches, confusion counts at
    `threshold`, precision / recall / f1 / accuracy, ROC-AUC and PR-AUC.  Everything accumulates on the
    GPU; the only host synchronisation is the final read-out."""
    from .metrics import BinaryAUROC, BinaryAveragePrecision, BinaryConfusionMatrix, summary_from_confusion
    conf, auroc, ap = None, BinaryAUROC(), BinaryAveragePrecision()
    loss_sum, n_batches = None, 0
    was_training = model.training
    model.eval()
    for batch in batches:
        out = model(batch)
        loss = criterion(out, batch.y, pos_weight)
        loss_sum = loss if loss_sum is None else loss_sum + loss
        n_batches += 1
        if conf is None:
            conf = BinaryConfusionMatrix(threshold, device=out.device)
        conf.update_from_logits(out, batch.y)
        prob = torch.sigmoid(out)
        auroc.update(prob, batch.y)
        ap.update(prob, batch.y)
    model.train(was_training)
    if conf is None:
        raise ValueError("evaluate() needs at least one batch")
    res = summary_from_confusion(conf.compute())
    res.update(loss=float(loss_sum) / n_batches, roc_auc=float(auroc.compute()), pr_auc=float(ap.compute()))
    return res


class GraphedTrainStep:
    """The whole train step (zero_grad -> forward -> loss -> backward -> Adam) of ONE fixed batch captured
    into a HIP graph and replayed.  For the reference's regime (32 small sub-graphs per batch,
    pangnn.py:152-216) a step is ~70 launches of a few microseconds of GPU work each, i.e. bound by the
    host launch path; replaying a captured graph removes it.  One instance per distinct batch (shapes are
    baked in); parameters and optimizer state are shared, so instances for different batches can be
    replayed in any order, exactly like iterating a DataLoader.

    The batch's structures / normalisations are built (and cached on the batch) by the warm-up steps
    that precede capture, so the captured region contains kernel launches only."""

    def __init__(self, model, optimizer, graph, labels, pos_weight, warmup: int = 2):
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                raise ValueError("GraphedTrainStep needs torch.optim.Adam(..., capturable=True)")
        self.model, self.optimizer = model, optimizer
        self.graph, self.labels, self.pos_weight = graph, labels, pos_weight
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(s)
        self.cuda_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.cuda_graph):
            self.loss, self.logits = self._step()

    def _step(self):
        self.optimizer.zero_grad(set_to_none=True)
        if hasattr(self.model, "loss_and_logits"):
            loss, out = self.model.loss_and_logits(self.graph, self.labels, self.pos_weight)
        else:
            out = self.model(self.graph)
            loss = criterion(out, self.labels, self.pos_weight)
        loss.backward(PF.unit_grad(loss.device))
        self.optimizer.step()
        return loss.detach(), out.detach()

    def __call__(self):
        self.cuda_graph.replay()
        return self.loss, self.logits
