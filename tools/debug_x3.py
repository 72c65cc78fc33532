import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import random_graph
from pangnn_amd import functional as PF
from pangnn_amd.graph import EdgeStructure
dev = torch.device("cuda")
for e in (31, 30, 32, 63, 95):
    torch.manual_seed(e)
    n, d = 257, 64
    ei, _ = random_graph(n, e, seed=e, isolated=0.0, hub=min(e, 700))
    ei = ei[:, torch.argsort(ei[0] * n + ei[1])]
    P, Q = torch.randn(n, d), torch.randn(n, d)
    W2, b2, w3, b3 = torch.randn(d, d) / 8, torch.randn(d), torch.randn(d), torch.randn(1)
    y = (torch.rand(e) < 0.3).float()
    pw = torch.tensor(3.0)
    lv = [t.clone().double().requires_grad_(True) for t in (P, Q, W2, b2, w3, b3)]
    ref = torch.relu(torch.relu(lv[0][ei[0]] + lv[1][ei[1]]) @ lv[2].t() + lv[3]) @ lv[4] + lv[5]
    torch.nn.functional.binary_cross_entropy_with_logits(ref, y.double(), pos_weight=pw.double()).backward()
    perm = torch.randperm(e)
    for mode in (0, 1):
        PF.DECODER_PRECISION = mode
        for name, eix, yy in (("sorted", ei, y), ("perm", ei[:, perm].contiguous(), y[perm])):
            st = EdgeStructure(eix.to(dev), n)
            leaves = [t.clone().to(dev).requires_grad_(True) for t in (P, Q, W2, b2, w3, b3)]
            loss, logits = PF.decoder_loss(leaves[0], leaves[1], st, None, None, leaves[2], leaves[3], leaves[4], leaves[5],
                                           yy.to(dev), pw.to(dev), e)
            loss.backward()
            errs = [float((a.grad.cpu().double() - b.grad).abs().max() / (b.grad.abs().max() + 1e-30)) for a, b in zip(leaves, lv)]
            print(e, "mode", mode, name, " ".join(f"{x:.1e}" for x in errs))
