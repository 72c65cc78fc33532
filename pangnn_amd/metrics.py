"""Evaluation metrics of the link predictor, accumulated on the GPU.

The reference keeps three torchmetrics objects (pangnn.py:104-108,218-222,255-262,268-285):
`BinaryConfusionMatrix` (thresholded predictions), `BinaryAUROC` and `BinaryAveragePrecision` (exact,
`thresholds=None`: every distinct score is a threshold).  torchmetrics is a third-party package that is
not in the reference tree; the classes here keep its `update / compute / reset` protocol and the
definitions it documents, and tests pin them against scikit-learn (`confusion_matrix`, `roc_auc_score`,
`average_precision_score`), which implements the same definitions.

  * confusion counts: one HIP pass over the logits (sigmoid + threshold + 4 integer counters,
    `pangnn_confusion_update_f32`), no [E] probability / prediction tensors, no host sync per batch.
  * AUROC / average precision: scores and labels of all batches stay on the device; `compute()` is one
    descending sort + prefix sums in int64/float64 (ties share one threshold, as in both libraries).
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch

from . import _lib


class BinaryConfusionMatrix:
    """[[tn, fp], [fn, tp]] (row = true label, column = prediction), int64 on the device."""

    def __init__(self, threshold: float = 0.5, device=None):
        self.threshold = float(threshold)
        self.counts = torch.zeros(4, dtype=torch.int64, device=device if device is not None else "cuda")

    def _launch(self, scores, target, threshold, apply_sigmoid):
        lib = _lib.load()
        _lib.require_device(scores, target)
        s = scores.detach().reshape(-1).to(torch.float32).contiguous()
        t = target.detach().reshape(-1).to(torch.float32).contiguous()
        if s.shape != t.shape:
            raise ValueError(f"{s.shape[0]} scores for {t.shape[0]} labels")
        if self.counts.device != s.device:
            self.counts = self.counts.to(s.device)
        with torch.cuda.device(s.device):
            _lib.check(lib.pangnn_confusion_update_f32(s.data_ptr(), t.data_ptr(), s.shape[0], float(threshold),
                                                       int(apply_sigmoid), self.counts.data_ptr(),
                                                       _lib.stream_ptr()), "pangnn_confusion_update_f32")

    def update_from_logits(self, logits, target, threshold: Optional[float] = None):
        """sigmoid(logits) >= threshold, counted in one pass (pangnn.py:219-222 without the temporaries)"""
        self._launch(logits, target, self.threshold if threshold is None else threshold, True)

    def update(self, preds, target):
        """torchmetrics signature: `preds` are 0/1 predictions (what pangnn.py:221 passes) or probabilities
        (thresholded at `self.threshold`)"""
        if preds.dtype.is_floating_point:
            self._launch(preds, target, self.threshold, False)
        else:
            self._launch(preds.to(torch.float32), target, 0.5, False)

    def compute(self) -> torch.Tensor:
        return self.counts.view(2, 2).clone()

    def reset(self):
        self.counts.zero_()


def summary_from_confusion(conf) -> dict:
    """pangnn.py:268-285: precision / recall / f1 / accuracy with the reference's 1e-10 guards"""
    tn, fp, fn, tp = (float(v) for v in conf.reshape(-1).tolist())
    precision = tp / (tp + fp + 1e-10)
    recall = tp / (tp + fn + 1e-10)
    f1 = 2 * (precision * recall) / (precision + recall + 1e-10)
    total = tp + tn + fp + fn
    return dict(tn=int(tn), fp=int(fp), fn=int(fn), tp=int(tp), precision=precision, recall=recall, f1=f1,
                accuracy=(tp + tn) / total if total else 0.0)


class _RankingMetric:
    """keeps (score, label) of every update on the device, like torchmetrics with thresholds=None.
    `share_curve_with=other`: the two metrics are going to be fed the same tensors (ROC-AUC and PR-AUC of one validation pass,
    pangnn.py:255-285) — whichever computes second reuses the sorted curve of the first instead of sorting the 7.5e7 scores of
    config 4 again (the sort is most of the pass).  The curve lives on the metric objects and goes with them."""

    def __init__(self, share_curve_with=None):
        self.scores, self.labels, self._src, self._curve_kept = [], [], [], None
        self._partner = None if share_curve_with is None else weakref.ref(share_curve_with)     # weak both ways: no cycle
        if share_curve_with is not None:
            share_curve_with._partner = weakref.ref(self)

    def update(self, preds, target):
        self.scores.append(preds.detach().reshape(-1).to(torch.float32))
        self.labels.append((target.detach().reshape(-1) > 0.5))
        self._src.append((preds, target, preds._version, target._version))
        self._curve_kept = None

    def reset(self):
        self.scores, self.labels, self._src, self._curve_kept = [], [], [], None

    def _same_data(self, other) -> bool:
        return len(self._src) == len(other._src) and all(
            a[0] is b[0] and a[1] is b[1] and a[2] == b[2] and a[3] == b[3] for a, b in zip(self._src, other._src))

    def _curve(self):
        """(tps, fps) at every distinct threshold, scores descending; int64"""
        if self._curve_kept is None:
            p = None if self._partner is None else self._partner()
            if p is not None and p._curve_kept is not None and self._same_data(p):
                self._curve_kept = p._curve_kept
            else:
                self._curve_kept = self._curve_of(torch.cat(self.scores), torch.cat(self.labels))
        return self._curve_kept

    @staticmethod
    def _curve_of(s, y):
        s, order = torch.sort(s, descending=True, stable=True)
        y = y[order].to(torch.int64)
        n = s.shape[0]
        last = torch.ones(n, dtype=torch.bool, device=s.device)
        last[:-1] = s[1:] != s[:-1]                       # last element of each run of equal scores
        idx = torch.nonzero(last).view(-1)
        tps = torch.cumsum(y, 0)[idx]
        fps = idx + 1 - tps
        return tps, fps


class BinaryAUROC(_RankingMetric):
    """area under the ROC curve by the trapezoidal rule over all distinct thresholds; 0 if one class is absent
    (torchmetrics returns 0 with a warning there)"""

    def compute(self) -> torch.Tensor:
        if not self.scores or sum(t.numel() for t in self.scores) == 0:
            return torch.zeros((), dtype=torch.float32)
        tps, fps = self._curve()
        p, n = tps[-1], fps[-1]
        if int(p) == 0 or int(n) == 0:
            return torch.zeros((), dtype=torch.float32, device=tps.device)
        z = torch.zeros(1, dtype=torch.float64, device=tps.device)
        tpr = torch.cat([z, tps.double() / p.double()])
        fpr = torch.cat([z, fps.double() / n.double()])
        return torch.trapezoid(tpr, fpr).to(torch.float32)


class BinaryAveragePrecision(_RankingMetric):
    """AP = sum_k (R_k - R_{k-1}) P_k over distinct thresholds (no interpolation)"""

    def compute(self) -> torch.Tensor:
        if not self.scores or sum(t.numel() for t in self.scores) == 0:
            return torch.zeros((), dtype=torch.float32)
        tps, fps = self._curve()
        p = tps[-1]
        if int(p) == 0:
            return torch.zeros((), dtype=torch.float32, device=tps.device)
        precision = tps.double() / (tps + fps).double()
        recall = tps.double() / p.double()
        prev = torch.cat([torch.zeros(1, dtype=torch.float64, device=tps.device), recall[:-1]])
        return ((recall - prev) * precision).sum().to(torch.float32)
