"""`AlternateGCN` — drop-in for /root/reference/src/gnn.py:84-207 on MI355X.

Same constructor signature (`device, dataset, categorical_nodes, dims=[node_dim, hidden_dim]`), same
`forward(graph) -> logits [E]`, same `state_dict` keys and shapes:

  embedding.{weight[D,1],bias[D]}   conv_in.{bias[H],lin.weight[H,D]}   conv_hidden.{bias[H],lin.weight[H,H]}
  conv_out.{bias[D],lin.weight[D,H]}   linear_out.{weight[D,H],bias[D]}
  mlp.0.{weight[D,2D(+1)],bias[D]}   mlp.2.{weight[D,D],bias[D]}   mlp.4.{weight[1,D],bias[1]}

The reference reads its topology flags from a module-global argparse namespace
(gnn.py:5,111,128,132,143,171-180); here they are constructor kwargs, and `args=` accepts that same
namespace (anything with the attributes) so the reference's call site works unchanged.

Decoder: `mlp.0` applied to cat(z[src], z[dst] [, w]) is computed in the re-associated form
P[src] + Q[dst] (+ w*c) with P = z W_a^T, Q = z W_b^T + b — identical algebra, E*2D*D fewer
multiply-adds and no [E, 2D] intermediate (pangnn_edge_pair_add_f32).  `fused_decoder=False`
selects the literal gather-concat-Linear form (pangnn_edge_gather_concat_f32).

First layer: `conv_in(embedding(x))` with the scalar node feature is one operator evaluated by linearity
(functional._EmbedConvIn): A_hat (x w^T + 1 b^T) W^T + b_in = r a^T + s c^T + b_in with the node vectors
r = A_hat x, s = A_hat 1 computed once per graph by the propagate kernel and a = W w, c = W b per step — one
[N, H] write forward, three weighted column sums of dL/dout backward, no per-step propagate or dense product.
`fuse_embedding="propagate"` = round 2's form (per-step propagate of h0, embedding gradients by linearity),
`fuse_embedding=False` the layer-by-layer form (what the reference executes); all three are tested against
each other and the oracle.

Activations: every `h = ELU(layer(...))` of the encoder is consumed by exactly one dense layer, so the ELU runs
inside that layer's kernels (`functional.linear(..., in_act=1)`); `fold_activation=False` applies it as its own
op.  `encode()` always returns activated embeddings.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from . import functional as PF
from .convolution import GCNConv
from .graph import structure_of

_FLAG_DEFAULTS = dict(union_edge_weights=False, base_model=False, skip_connections=False,
                      decoder="mlp", neighbours=1)


class AlternateGCN(nn.Module):
    def __init__(self, device=None, dataset=None, categorical_nodes: bool = False, dims=(64, 128),
                 args=None, num_nodes: Optional[int] = None, fused_decoder: bool = True,
                 fuse_embedding: bool = True, fold_activation: bool = True, fuse_first_dense: Optional[bool] = None,
                 deferred_logits: Optional[bool] = None, **flags):
        super().__init__()
        self.device = device
        cfg = dict(_FLAG_DEFAULTS)
        if args is not None:
            for k in cfg:
                if hasattr(args, k):
                    cfg[k] = getattr(args, k)
        unknown = set(flags) - set(cfg)
        if unknown:
            raise TypeError(f"unknown flags: {sorted(unknown)}")
        cfg.update(flags)
        self.flags = SimpleNamespace(**cfg)
        self.fused_decoder = fused_decoder
        self.fuse_embedding = fuse_embedding
        self.fold_activation = fold_activation
        # None: on unless PANGNN_FUSE_FIRST_DENSE=0 (same-box A/B of bench.py)
        self.fuse_first_dense = (os.environ.get("PANGNN_FUSE_FIRST_DENSE", "1") != "0") if fuse_first_dense is None \
            else bool(fuse_first_dense)
        # None: on unless PANGNN_DEFERRED_LOGITS=0 — training-mode forward() returns a DeferredLogits handle (deferred.py)
        self.deferred_logits = (os.environ.get("PANGNN_DEFERRED_LOGITS", "1") != "0") if deferred_logits is None \
            else bool(deferred_logits)
        node_embedding_dim, hidden_dim = dims

        if categorical_nodes:
            # The reference's categorical path cannot run (gnn.py:93 takes len() of a list of graphs,
            # and x is float ones, dataset.py:369).  Build-defined semantics:
            # x holds integer gene positions, one embedding row per node.
            if num_nodes is None:
                num_nodes = len(dataset.x) if hasattr(dataset, "x") else None
            if num_nodes is None:
                raise ValueError("categorical_nodes=True needs num_nodes=")
            self.embedding = nn.Embedding(int(num_nodes), node_embedding_dim)
        else:
            self.embedding = nn.Linear(1, node_embedding_dim)
        self.categorical_nodes = categorical_nodes

        self.conv_in = GCNConv(node_embedding_dim, hidden_dim, add_self_loops=False)
        self.conv_hidden = GCNConv(hidden_dim, hidden_dim, add_self_loops=False)
        self.conv_out = GCNConv(hidden_dim, node_embedding_dim, add_self_loops=False)
        self.linear_out = nn.Linear(hidden_dim, node_embedding_dim)
        self.activation_fct = nn.ELU()
        d = node_embedding_dim
        self.mlp = nn.Sequential(
            nn.Linear(2 * d + (1 if self.flags.skip_connections else 0), d), nn.ReLU(),
            nn.Linear(d, d), nn.ReLU(), nn.Linear(d, 1))
        self.epoch = 0
        if device is not None:
            self.to(device)

    # ---------------------------------------------------------------------------------
    def _embed_conv_in(self, graph, ei, name):
        """act-less `conv_in(embedding(x))` (gnn.py:125-131,143-146,156-158)"""
        x, conv = graph.x, self.conv_in
        _lib.require_device(x)
        if self.categorical_nodes:
            return conv(self.embedding(x.long().view(-1)), ei, graph.edge_attr, graph=graph, name=name)
        if self.fuse_embedding and (self.fuse_embedding != "propagate" or conv.in_channels < conv.out_channels):
            st = structure_of(ei, x.shape[0], holder=graph, name=name)
            w = graph.edge_attr
            if w is not None and w.shape[0] != st.num_edges:
                raise ValueError(f"edge_weight has {w.shape[0]} entries for {st.num_edges} edges")
            out_dtype = PF.autocast_rows_dtype(x)          # the autocast Linear's output type (bf16 / fp16 mixed precision)
            if self.fuse_embedding != "propagate":
                # the whole layer by linearity: r a^T + s c^T + b_in, r = A_hat x and s = A_hat 1 cached per graph
                # (functional._EmbedConvIn) — one [N, H] write per step, one pass over its gradient in backward
                return PF.embed_conv_in(x, self.embedding.weight, self.embedding.bias, conv.lin.weight, conv.bias, st,
                                        st.gcn_norm(w), out_dtype)
            # round 2's form: embedding + propagate as one operator whose backward yields the embedding's two parameter
            # gradients without the transposed propagate (functional._EmbedPropagate), then the dense layer
            agg = PF.embed_propagate(x, self.embedding.weight, self.embedding.bias, st, st.gcn_norm(w), tag=name)
            return PF.linear(agg, conv.lin.weight, conv.bias, 0, out_dtype)
        # Linear(1, D) on a [N,1] column is an outer product; as a GEMM its weight gradient is a
        # 64 x 1 x N problem that the BLAS library runs at < 0.1 TB/s
        h = x.float().view(-1, 1) * self.embedding.weight.view(1, -1) + self.embedding.bias
        return conv(h, ei, graph.edge_attr, graph=graph, name=name)

    def _embed_conv_in_then_dense(self, graph, ei, name, w_out, bias_out):
        """linear(ELU(conv_in(embedding(x))), w_out, bias_out) with the [N, H] rows of the first layer generated inside the
        dense layer's kernels (functional._EmbedConvInLinear) — or None where that operator does not apply: categorical
        nodes, fuse_embedding / fuse_first_dense / fold_activation off, widths its kernels do not cover, bf16 / fp16 autocast
        (the reference's autocast stores those rows in the autocast type; the generated rows are fp32)."""
        x, conv = graph.x, self.conv_in
        if self.categorical_nodes or not self.fuse_first_dense or not self._fold_elu() or not self.fuse_embedding \
                or self.fuse_embedding == "propagate":
            return None
        if w_out.shape[1] != conv.out_channels or not PF.embed_linear_supported(conv.out_channels, w_out.shape[0]):
            return None
        _lib.require_device(x)
        if PF.autocast_rows_dtype(x) is not None:
            return None
        st = structure_of(ei, x.shape[0], holder=graph, name=name)
        w = graph.edge_attr
        if w is not None and w.shape[0] != st.num_edges:
            raise ValueError(f"edge_weight has {w.shape[0]} entries for {st.num_edges} edges")
        return PF.embed_conv_in_linear(x, self.embedding.weight, self.embedding.bias, conv.lin.weight, conv.bias, w_out,
                                       bias_out, st, st.gcn_norm(w))

    def _fold_elu(self) -> bool:
        """the encoder's activation (gnn.py:108: ELU) can be folded into the dense layer that follows it"""
        a = self.activation_fct
        return self.fold_activation and isinstance(a, nn.ELU) and float(a.alpha) == 1.0

    def _encode_pre(self, graph):
        """(z or its pre-activation, pending): every `h = act(layer(...))` of gnn.py:125-166 is consumed by exactly
        one dense layer (GCNConv.lin of the next conv, linear_out, or the decoder's first layer), so with ELU the
        activation runs inside that layer's kernels (functional.linear, in_act) and only the LAST one can be
        left pending for the caller."""
        fl = self.flags
        act, fold = self.activation_fct, self._fold_elu()
        if fl.union_edge_weights:                                              # gnn.py:128-139
            ei = graph.union_edge_index
            hid = self.conv_hidden
            y = self._embed_conv_in_then_dense(graph, ei, "union", hid.lin.weight, None) \
                if hid.in_channels >= hid.out_channels else None
            h = self._embed_conv_in(graph, ei, "union") if y is None else None
            for k in range(max(fl.neighbours - 2, 1)):
                if k == 0 and y is not None:           # the first hidden layer's dense part came fused with conv_in
                    h = hid(y, ei, graph.edge_attr, graph=graph, name="union", dense_done=True)
                    continue
                h = hid(h, ei, graph.edge_attr, graph=graph, name="union", in_elu=True) if fold \
                    else hid(act(h), ei, graph.edge_attr, graph=graph, name="union")
            h = self.conv_out(h, ei, graph=graph, name="union", in_elu=True) if fold \
                else self.conv_out(act(h), ei, graph=graph, name="union")
        elif fl.base_model:                                                    # gnn.py:143-150
            h = self._embed_conv_in_then_dense(graph, graph.edge_index, "sim", self.linear_out.weight, self.linear_out.bias)
            if h is None:
                h = self._embed_conv_in(graph, graph.edge_index, "sim")
                h = PF.linear(h, self.linear_out.weight, self.linear_out.bias, 1) if fold \
                    else PF.linear(act(h), self.linear_out.weight, self.linear_out.bias)
        else:                                                                  # gnn.py:153-166
            out = self.conv_out
            y = self._embed_conv_in_then_dense(graph, graph.edge_index, "sim", out.lin.weight, None) \
                if out.in_channels >= out.out_channels else None
            if y is not None:                          # conv_out's dense part came fused with conv_in: propagate + bias left
                h = out(y, graph.neighbour_edge_index, graph=graph, name="nb", dense_done=True)
            else:
                h = self._embed_conv_in(graph, graph.edge_index, "sim")
                h = out(h, graph.neighbour_edge_index, graph=graph, name="nb", in_elu=True) if fold \
                    else out(act(h), graph.neighbour_edge_index, graph=graph, name="nb")
        return h, True

    def encode(self, graph) -> torch.Tensor:
        h, pending = self._encode_pre(graph)
        return self.activation_fct(h) if pending else h

    def _decoder_inputs(self, z, graph, in_act: int = 0):
        """(pq [N,2D], structure, extra, cvec) of the re-associated first decoder layer"""
        fl = self.flags
        ei = graph.edge_index
        st = structure_of(ei, z.shape[0], holder=graph, name="sim")
        d = z.shape[1]
        extra = graph.edge_attr[: ei.shape[1]] if fl.skip_connections else None
        w = self.mlp[0].weight
        w_pq, b_pq, cvec = PF.pq_operands(w, self.mlp[0].bias, d, bool(fl.skip_connections))
        # bf16 / fp16 mixed precision: mlp[0] is an autocast Linear, its node-level halves are stored (and gathered) in that type
        pq_dtype = PF.autocast_rows_dtype(z) if (d == 64 and PF.DECODER_PRECISION == 1) else None
        return PF.linear(z, w_pq, b_pq, in_act, pq_dtype), st, extra, cvec

    def _fused_decoder_operands(self, graph):
        """(pq, structure, extra, cvec) of the one-pass training decoder, or None where that kernel does not apply (another
        decoder, node_dim != 64, no gradient wanted): the encoder (gnn.py:125-166) with its last ELU folded into the
        node-level half of `mlp[0]` (gnn.py:110,173-175)"""
        z, pending = self._encode_pre(graph)
        fused = "mlp" in self.flags.decoder and self.fused_decoder is True and z.shape[1] == 64 and torch.is_grad_enabled()
        if not (fused and pending and self._fold_elu()):
            z, pending = (self.activation_fct(z) if pending else z), False
        if not fused:
            return None, z
        return self._decoder_inputs(z, graph, 1 if pending else 0), z

    def loss_and_logits(self, graph, labels, pos_weight=None):
        """`criterion(model(graph), labels)` (pangnn.py:200-203) as ONE decoder pass when the fused kernel
        applies (mlp decoder, node_dim 64): returns (loss, detached logits).  Falls back to forward +
        criterion otherwise.  `model(graph)` followed by torch's / this package's BCEWithLogitsLoss reaches the same pass
        through the handle `forward` returns in training mode (deferred.DeferredLogits)."""
        from .train import criterion
        ops, z = self._fused_decoder_operands(graph)
        if ops is not None:
            pq, st, extra, cvec = ops
            # a padded fixed-shape batch (SubGraphDataset.padded_buffers) carries the number of its real edges on the device
            return PF.decoder_loss_pq(pq, st, extra, cvec, self.mlp[2].weight, self.mlp[2].bias,
                                      self.mlp[4].weight.view(-1), self.mlp[4].bias, labels, pos_weight,
                                      labels.shape[0], live=getattr(graph, "live_edges", None))
        if getattr(graph, "live_edges", None) is not None:
            raise NotImplementedError("a padded fixed-shape batch needs the fused training decoder (mlp decoder, node_dim 64)")
        out = self._decode(z, graph)
        return criterion(out, labels, pos_weight), out.detach()

    def _defers(self) -> bool:
        """training-mode `forward` hands out a DeferredLogits handle instead of launching the inference decoder: only where the
        one-pass training decoder could resolve it, and never while a tracer / dispatch mode watches (a traced program
        wants plain tensors: `loss_and_logits` is the traceable training entry)"""
        fl = self.flags
        return (self.deferred_logits and self.training and torch.is_grad_enabled() and fl.decoder == "mlp"
                and self.fused_decoder is True and self.mlp[2].in_features == 64 and PF.DECODER_PRECISION == 1
                and not torch.compiler.is_compiling() and not PF.observed())

    def _decode(self, nodes, graph):
        fl = self.flags
        out = None
        if "mlp" in fl.decoder:
            out = self.decode_mlp(nodes, graph)
        if "cosine" in fl.decoder:
            out = self.cosine_sim(nodes, graph.edge_index, graph)
        if "dot" in fl.decoder:
            out = self.decode(nodes, graph.edge_index, graph)
        return out

    def decode_mlp(self, z, graph) -> torch.Tensor:
        fl = self.flags
        ei = graph.edge_index
        st = structure_of(ei, z.shape[0], holder=graph, name="sim")
        d = z.shape[1]
        extra = graph.edge_attr[: ei.shape[1]] if fl.skip_connections else None   # gnn.py:173
        lin0 = self.mlp[0]
        if self.fused_decoder and d % 4 == 0:
            w = lin0.weight
            # one node-level product gives P | Q = z [W_a ; W_b]^T + [0 ; b1]
            w_pq, b_pq, cvec = PF.pq_operands(w, lin0.bias, d, bool(fl.skip_connections))
            fused = d == 64 and self.fused_decoder != "pair_add"
            pq_dtype = PF.autocast_rows_dtype(z) if (fused and PF.DECODER_PRECISION == 1) else None
            pq = PF.linear(z, w_pq, b_pq, 0, pq_dtype)
            if fused:
                # whole per-edge MLP in one HIP kernel (f32 MFMA), no [E, D] tensor in HBM on the way
                return PF.decoder_mlp_pq(pq, st, extra, cvec, self.mlp[2].weight, self.mlp[2].bias,
                                         self.mlp[4].weight.view(-1), self.mlp[4].bias)
            h = PF.edge_pair_add(pq[:, :d].contiguous(), pq[:, d:].contiguous(), st, extra, cvec)
        else:
            h = lin0(PF.edge_gather_concat(z, st, extra))
        for layer in list(self.mlp)[1:]:
            # [E, D] intermediates are the largest tensors of the step: rectify them in place
            h = torch.relu_(h) if isinstance(layer, nn.ReLU) else layer(h)
        return h.squeeze(-1)

    def forward(self, graph) -> torch.Tensor:
        """logits [E] (gnn.py:121-200).  In training mode with the mlp decoder the result is a `DeferredLogits` handle: the
        encoder and the node-level half of `mlp[0]` run here (inside accelerate's autocast wrapper of `forward`), the per-edge
        decoder when the handle is used — as ONE pass with the loss and every gradient if that use is
        `BCEWithLogitsLoss(pos_weight)(output, labels)` (pangnn.py:98,203), through the inference kernel on any other use."""
        if self._defers():
            ops, z = self._fused_decoder_operands(graph)
            if ops is not None:
                return self._deferred(graph, *ops)
            return self._decode(z, graph)
        return self._decode(self.encode(graph), graph)

    def _deferred(self, graph, pq, st, extra, cvec):
        from .deferred import DeferredLogits
        w2, b2, w3, b3 = self.mlp[2].weight, self.mlp[2].bias, self.mlp[4].weight.view(-1), self.mlp[4].bias
        live = getattr(graph, "live_edges", None)

        def materialize():
            if live is not None:
                raise NotImplementedError("a padded fixed-shape batch needs the fused training decoder: call "
                                          "BCEWithLogitsLoss(pos_weight)(output, labels) on the model's output first")
            return PF.decoder_mlp_pq(pq, st, extra, cvec, w2, b2, w3, b3)

        def fused_loss(labels, pos_weight):
            return PF.decoder_loss_pq(pq, st, extra, cvec, w2, b2, w3, b3, labels, pos_weight, labels.shape[0], live=live)

        return DeferredLogits(st.num_edges, pq.device, materialize, fused_loss)

    def _pairs(self, z, edge_index, graph=None):
        st = structure_of(edge_index, z.shape[0], holder=graph, name="sim")
        both = PF.edge_gather_concat(z, st)
        d = z.shape[1]
        return both[:, :d], both[:, d:]

    def decode(self, z, edge_index, graph=None):
        # gnn.py:202-204 is `z[src] @ z[dst]`, a shape error unless E == D (broken in the
        # reference); build-defined semantics: per-edge dot product.
        a, b = self._pairs(z, edge_index, graph)
        return (a * b).sum(dim=1)

    def cosine_sim(self, z, edge_index, graph=None):
        a, b = self._pairs(z, edge_index, graph)
        return F.cosine_similarity(a, b, dim=1)
