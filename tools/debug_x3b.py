"""Diagnostic (kept as the reproducer of DESIGN.md §4 (iii)): runs the bf16 matrix-pipe training decoder six times
on a 30-edge list (one partial tile) and reports which gradient entries differ from the f32-MFMA mode.  With
decoder.hip built WITHOUT -mllvm -amdgpu-mfma-vgpr-form=1 about four runs in six are wrong."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import random_graph
from pangnn_amd import functional as PF
from pangnn_amd.graph import EdgeStructure
dev = torch.device("cuda")
e = 30
torch.manual_seed(e)
n, d = 257, 64
ei, _ = random_graph(n, e, seed=e, isolated=0.0, hub=min(e, 700))
perm = torch.randperm(e)
ei = ei[:, perm].contiguous()
P, Q = torch.randn(n, d), torch.randn(n, d)
W2, b2, w3, b3 = torch.randn(d, d) / 8, torch.randn(d), torch.randn(d), torch.randn(1)
y = (torch.rand(e) < 0.3).float()
pw = torch.tensor(3.0)
st = EdgeStructure(ei.to(dev), n)
def run(mode):
    PF.DECODER_PRECISION = mode
    leaves = [t.clone().to(dev).requires_grad_(True) for t in (P, Q, W2, b2, w3, b3)]
    # the raw [E,64] dL/dh1 via the non-fused entry point is not exposed; use Q grads per target (by_dst sum)
    loss, logits = PF.decoder_loss(leaves[0], leaves[1], st, None, None, leaves[2], leaves[3], leaves[4], leaves[5],
                                   y.to(dev), pw.to(dev), e)
    loss.backward()
    return [t.grad.cpu() for t in leaves], logits.cpu()
ref, lref = run(0)
for it in range(6):
    g, l = run(1)
    dq = (g[1] - ref[1]).abs()
    bad_rows = (dq.max(1).values > 1e-4 * ref[1].abs().max()).nonzero().view(-1).tolist()
    bad_cols = (dq.max(0).values > 1e-4 * ref[1].abs().max()).nonzero().view(-1).tolist()
    dw = (g[2] - ref[2]).abs()
    bw_rows = (dw.max(1).values > 1e-4 * ref[2].abs().max()).nonzero().view(-1).tolist()
    bw_cols = (dw.max(0).values > 1e-4 * ref[2].abs().max()).nonzero().view(-1).tolist()
    # which edges have those targets
    tg = ei[1].tolist()
    bad_edges = [i for i, t in enumerate(tg) if t in bad_rows]
    print("iter", it, "logit err", float((l - lref).abs().max()), "Q bad rows", len(bad_rows), "edges", bad_edges, "cols", bad_cols[:70], "| W2 bad rows", bw_rows[:70], "cols", bw_cols[:70])
