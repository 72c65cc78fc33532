"""Evaluation metrics (pangnn.py:218-222,255-285) against scikit-learn, which implements the same
definitions as the torchmetrics classes the reference uses."""
import numpy as np
import pytest
import torch
from sklearn.metrics import average_precision_score, confusion_matrix, roc_auc_score

from pangnn_amd.metrics import (BinaryAUROC, BinaryAveragePrecision, BinaryConfusionMatrix,
                                summary_from_confusion)


def _scores(n, seed, ties):
    rng = np.random.default_rng(seed)
    y = (rng.random(n) < 0.15).astype(np.float32)
    s = rng.normal(size=n).astype(np.float32) + 1.2 * y
    if ties:
        s = np.round(s, 1)                         # many equal scores: ties must share one threshold
    return s, y


@pytest.mark.parametrize("n,ties", [(2, False), (50, True), (5000, False), (5000, True)])
def test_ranking_metrics_match_sklearn_on_host_tensors(n, ties):
    s, y = _scores(n, n + ties, ties)
    y[0], y[1] = 0.0, 1.0
    p = 1.0 / (1.0 + np.exp(-s))
    auroc, ap = BinaryAUROC(), BinaryAveragePrecision()
    for lo in range(0, n, 1700):                   # several updates, like one per validation batch
        auroc.update(torch.from_numpy(p[lo:lo + 1700]), torch.from_numpy(y[lo:lo + 1700]))
        ap.update(torch.from_numpy(p[lo:lo + 1700]), torch.from_numpy(y[lo:lo + 1700]))
    assert abs(float(auroc.compute()) - roc_auc_score(y, p)) < 1e-6
    assert abs(float(ap.compute()) - average_precision_score(y, p)) < 1e-6
    auroc.reset()
    assert float(auroc.compute()) == 0.0


def test_two_ranking_metrics_fed_the_same_tensors_share_one_sorted_curve(monkeypatch):
    """ROC-AUC and PR-AUC of one validation pass get the same (probabilities, labels) tensors: with share_curve_with the second
    compute() reuses the first one's curve (one sort) — and does NOT for other tensors, after an in-place change, or unpaired"""
    from pangnn_amd import metrics as M
    s, y = _scores(3000, 7, True)
    y[0], y[1] = 0.0, 1.0
    p = torch.from_numpy(1.0 / (1.0 + np.exp(-s)))
    t = torch.from_numpy(y)
    calls = []
    real = M._RankingMetric._curve_of
    monkeypatch.setattr(M._RankingMetric, "_curve_of", staticmethod(lambda a, b: (calls.append(1), real(a, b))[1]))
    auroc = BinaryAUROC()
    ap = BinaryAveragePrecision(share_curve_with=auroc)
    auroc.update(p, t)
    ap.update(p, t)
    assert abs(float(ap.compute()) - average_precision_score(y, p.numpy())) < 1e-6          # either order
    assert abs(float(auroc.compute()) - roc_auc_score(y, p.numpy())) < 1e-6
    assert len(calls) == 1
    lone = BinaryAveragePrecision()                       # unpaired: its own sort
    lone.update(p, t)
    lone.compute()
    assert len(calls) == 2
    a2 = BinaryAUROC()
    b2 = BinaryAveragePrecision(share_curve_with=a2)
    a2.update(p, t)
    b2.update(p.clone(), t)                               # equal values, another tensor: computed afresh
    a2.compute(), b2.compute()
    assert len(calls) == 4
    a3 = BinaryAUROC()
    b3 = BinaryAveragePrecision(share_curve_with=a3)
    p3 = p.clone()
    a3.update(p3, t)
    a3.compute()
    p3.mul_(0.5)                                          # the same tensor changed in place before the partner saw it
    b3.update(p3, t)
    b3.compute()
    assert len(calls) == 6
    a3.reset()
    assert a3._curve_kept is None


def test_ranking_metrics_degenerate_classes():
    a, p = BinaryAUROC(), BinaryAveragePrecision()
    a.update(torch.tensor([0.2, 0.7]), torch.tensor([0.0, 0.0]))
    p.update(torch.tensor([0.2, 0.7]), torch.tensor([0.0, 0.0]))
    assert float(a.compute()) == 0.0 and float(p.compute()) == 0.0


def test_summary_uses_the_reference_guards():
    r = summary_from_confusion(torch.tensor([[5, 0], [0, 0]]))
    assert r["precision"] == 0.0 and r["recall"] == 0.0 and r["f1"] == 0.0 and r["accuracy"] == 1.0
    r = summary_from_confusion(torch.tensor([[50, 10], [5, 35]]))
    assert abs(r["precision"] - 35 / 45) < 1e-9 and abs(r["recall"] - 35 / 40) < 1e-9
    assert abs(r["accuracy"] - 0.85) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 63, 1000, 3000017])
@pytest.mark.parametrize("th", [0.5, 0.9])
def test_confusion_counts_on_device(n, th):
    dev = torch.device("cuda:0")
    s, y = _scores(n, n, False)
    x = torch.from_numpy(s).to(dev)
    yt = torch.from_numpy(y).to(dev)
    cm = BinaryConfusionMatrix(th, device=dev)
    half = n // 2
    cm.update_from_logits(x[:half], yt[:half])
    cm.update_from_logits(x[half:], yt[half:])
    prob = (1.0 / (1.0 + np.exp(-s.astype(np.float32)))).astype(np.float32)
    pred = (prob >= np.float32(th)).astype(np.int64)
    want = confusion_matrix(y.astype(np.int64), pred, labels=[0, 1])
    got = cm.compute().cpu().numpy()
    # a score within one rounding of the threshold may fall on either side (expf vs numpy's exp)
    near = int((np.abs(prob - th) < 1e-6).sum())
    assert np.abs(got - want).sum() <= 2 * near
    assert got.sum() == n
    # torchmetrics call forms: integer predictions, and probabilities thresholded by the metric
    cm2 = BinaryConfusionMatrix(th, device=dev)
    cm2.update(torch.from_numpy(pred).to(dev).int(), yt.int())
    assert np.array_equal(cm2.compute().cpu().numpy(), want)
    cm3 = BinaryConfusionMatrix(th, device=dev)
    cm3.update(torch.from_numpy(prob).to(dev), yt)
    assert np.array_equal(cm3.compute().cpu().numpy(), want)
    cm3.reset()
    assert int(cm3.compute().sum()) == 0


@pytest.mark.gpu
def test_evaluate_matches_sklearn_on_model_outputs():
    import pangnn_amd
    from pangnn_amd.train import evaluate
    from conftest import sub_graphs_from_golden
    dev = torch.device("cuda:0")
    graphs = [pangnn_amd.Data(s.x, s.edge_index, s.edge_attr, s.y, neighbour_edge_index=s.neighbour_edge_index)
              for s in sub_graphs_from_golden("cfg2_sim_1000x5")[:64]]
    batches = [pangnn_amd.Batch.from_data_list(graphs[i:i + 32]).to(dev) for i in (0, 32)]
    torch.manual_seed(3)
    model = pangnn_amd.AlternateGCN(dev, None, False, dims=[64, 64])
    pw = torch.tensor(3.0, device=dev)
    res = evaluate(model, batches, pw, threshold=0.5)
    with torch.no_grad():
        outs = [model(b) for b in batches]
    logit = torch.cat(outs).cpu().numpy()
    y = torch.cat([b.y for b in batches]).cpu().numpy()
    prob = 1.0 / (1.0 + np.exp(-logit))
    assert abs(res["roc_auc"] - roc_auc_score(y, prob)) < 1e-5
    assert abs(res["pr_auc"] - average_precision_score(y, prob)) < 1e-5
    want = confusion_matrix(y.astype(np.int64), (prob >= 0.5).astype(np.int64), labels=[0, 1])
    assert abs(res["tp"] - want[1, 1]) <= 1 and abs(res["tn"] - want[0, 0]) <= 1
    assert res["tp"] + res["tn"] + res["fp"] + res["fn"] == y.shape[0]
    ref_loss = np.mean([float(torch.nn.functional.binary_cross_entropy_with_logits(o, b.y, pos_weight=pw))
                        for o, b in zip(outs, batches)])
    assert abs(res["loss"] - ref_loss) < 1e-5
