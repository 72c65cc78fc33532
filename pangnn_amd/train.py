"""The link-prediction train step around the model (pangnn.py:88-98,194-216): zero_grad ->
forward -> BCEWithLogits(pos_weight) -> backward -> Adam.  Host side is plain PyTorch; every
gather/scatter inside `model` runs in libpangnn_hip.so."""
from __future__ import annotations

import torch

from . import functional as PF


def criterion(logits, labels, pos_weight):
    """torch.nn.BCEWithLogitsLoss(pos_weight=class_balance) (pangnn.py:98) — loss and dL/dlogits from one
    HIP pass (device tensors only: there is no CPU path)"""
    from .deferred import DeferredLogits
    if isinstance(logits, DeferredLogits):     # training-mode model(graph): resolved by the one-pass decoder when it still can be
        loss = logits.fused_bce(labels, pos_weight)
        if loss is not None:
            return loss
        logits = logits.materialize()
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
    conf, auroc = None, BinaryAUROC()
    ap = BinaryAveragePrecision(share_curve_with=auroc)      # fed the same tensors: one sort serves both
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


class ReplayedFreshStep:
    """The reference's real regime — `DataLoader(batch_size=32, shuffle=True)`: every step sees a NEW batch
    (pangnn.py:152-153,180-216) — as ONE captured HIP graph that serves every mini-batch of a `SubGraphDataset`.

    What a fresh batch needs (collation, both CSR orders and run-sum plans of both edge lists, the degree normalisations, the
    first layer's node vectors r / s, then forward / loss / backward / Adam) is ~65 launches of a few microseconds each and
    therefore bound by the host's launch path (0.7-0.86 ms per step).  Here all of it is captured once:
      * the batch lives in buffers of FIXED shapes sized to the data set's maxima (`SubGraphDataset.padded_spec`), collated
        on the device from a device-resident list of sub-graph ids by `pangnn_collate_subgraphs_padded`, with an inert padded
        tail (self loops spread over the padded nodes, which are never real; the decoder kernels take the number of real edges from device memory:
        padded edges get dL/dlogit = 0 and the loss is the mean over the real ones);
      * so every kernel of the step runs on one shape, none of them needs a host-known size of the batch, and nothing reads
        back: the step is capturable, and a replay collates and trains on whatever ids the list holds at that moment.
    Per step the host does two things: one tiny launch that writes the next batch's ids (`pangnn_set_i64`, values carried in
    the kernel arguments) and one graph launch.  `capture=False` runs the very same padded step eagerly (the reference
    point of the bit-equality test).  Results equal the unpadded fresh step (`train_step` on `ds.batch(...)`) up to fp32
    re-association: padded rows are zero rows of every sum, but the per-workgroup partial sums of the node-level kernels are
    grouped by the padded row count."""

    def __init__(self, model, optimizer, ds, pos_weight, batch_size: int = 32, graphs=None, capture: bool = True,
                 warmup: int = 2, slack: float = 1.4):
        """`slack`: the everyday buffers hold `slack` x the mean batch (SubGraphDataset.padded_spec); a batch that exceeds them
        runs through a second, worst-case set of buffers (and its own captured graph), built on first use.  None: worst-case
        buffers only."""
        if capture:
            for g in optimizer.param_groups:
                if not g.get("capturable", False):
                    raise ValueError("ReplayedFreshStep(capture=True) needs torch.optim.Adam(..., capturable=True)")
        if not hasattr(model, "loss_and_logits"):
            raise ValueError("ReplayedFreshStep needs a model with the fused loss_and_logits path")
        self.model, self.optimizer, self.ds, self.pos_weight = model, optimizer, ds, pos_weight
        self.capture, self.warmup = bool(capture), int(warmup)
        self._first = list(range(min(batch_size, ds.num_graphs))) if graphs is None else [int(i) for i in list(graphs)[:batch_size]]
        worst = ds.padded_spec(batch_size, graphs)
        tight = worst if slack is None else ds.padded_spec(batch_size, graphs, slack)
        self.spec, self.spec_worst = tight, worst
        self._slots = {}                                   # spec -> (buffers, captured graph | None, loss, logits)
        self._slot(tight)

    def _slot(self, spec):
        """the buffers (and, with capture, the captured graph) of the padded shapes `spec`, built on first use"""
        hit = self._slots.get(spec)
        if hit is not None:
            return hit
        ds, model, optimizer = self.ds, self.model, self.optimizer
        buf = ds.padded_buffers(spec)
        first = [i for i in self._first]
        while len(first) > 1 and not ds.fits(spec, first):
            first.pop()
        ds.set_graph_ids(buf, first)
        if not self.capture:
            self._slots[spec] = (buf, None, None, None)
            return self._slots[spec]
        # the warm-up steps run on a side stream and train on `first`: parameters and optimizer state are restored
        # afterwards (values copied back into the very tensors the captured graph updates), so that building a slot is
        # not a training step
        import copy
        saved_p = [p.detach().clone() for p in model.parameters()]
        opt_state = copy.deepcopy(optimizer.state_dict())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(max(self.warmup, 1)):
                self._step(buf)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss, logits = self._step(buf)
        torch.cuda.synchronize()
        with torch.no_grad():
            for p, q in zip(model.parameters(), saved_p):
                p.copy_(q)
            new_state = optimizer.state_dict()["state"]
            for k, st_ in new_state.items():
                old = opt_state["state"].get(k)
                for name, v in st_.items():
                    if torch.is_tensor(v):
                        if old is not None and torch.is_tensor(old.get(name)):
                            v.copy_(old[name])
                        else:
                            v.zero_()                      # a fresh optimizer: forget what the warm-up accumulated
        self._slots[spec] = (buf, graph, loss, logits)
        return self._slots[spec]

    def _step(self, buf):
        """one padded fresh step: everything a new batch needs, no host read-back"""
        self.ds.collate_padded(buf)       # (also replaces / drops the structures cached on the buffers' addresses)
        self.optimizer.zero_grad(set_to_none=True)
        loss, out = self.model.loss_and_logits(buf, buf.y, self.pos_weight)
        loss.backward(PF.unit_grad(loss.device))
        self.optimizer.step()
        return loss.detach(), out.detach()

    def __call__(self, graph_ids):
        """train on the disjoint union of the sub-graphs `graph_ids` (host ints, any order, at most batch_size);
        returns (loss, logits of the batch's real edges) as device tensors without synchronising.  With capture the two
        are VIEWS of the captured graph's static outputs: the next call overwrites them — clone what must outlive it."""
        spec = self.spec if self.ds.fits(self.spec, graph_ids) else self.spec_worst
        buf, graph, loss, logits = self._slot(spec)
        edges = self.ds.set_graph_ids(buf, graph_ids)[1]
        if graph is not None:
            graph.replay()
            return loss, logits[:edges]
        loss, out = self._step(buf)
        return loss, out[:edges]
