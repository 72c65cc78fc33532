"""Diagnostic: the two-wave-per-SIMD training decoder (csrc/decoder16.hip) against an fp64 torch evaluation over
edge counts that exercise partial tiles, several tiles per wave, sorted / unsorted lists, with / without the
skip-connection term, fused-loss and given-gradient entry points.  python tools/check_decoder16.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import random_graph
from pangnn_amd import functional as PF
from pangnn_amd.graph import EdgeStructure

dev = torch.device("cuda")
PF.DECODER_PRECISION = 1
worst = 0.0
for e in (1, 15, 16, 17, 31, 32, 33, 48, 64, 1000, 70001, 300007):
    for srt in (True, False):
        for skip in (False, True):
            torch.manual_seed(e + skip)
            n, d = 97 if e < 100000 else 5003, 64
            ei, w = random_graph(n, e, seed=e, isolated=0.0)
            if srt:
                ei = ei[:, torch.argsort(ei[0] * n + ei[1])]
            P, Q = torch.randn(n, d), torch.randn(n, d)
            W2, b2, w3, b3, cv = torch.randn(d, d) / 8, torch.randn(d), torch.randn(d), torch.randn(1), torch.randn(d)
            extra = (w / 40) if skip else None
            y = (torch.rand(e) < 0.3).float()
            pw = torch.tensor(2.5)
            lv = [t.clone().double().requires_grad_(True) for t in (P, Q, W2, b2, w3, b3, cv)]
            h1 = lv[0][ei[0]] + lv[1][ei[1]]
            if skip:
                h1 = h1 + extra.double().unsqueeze(1) * lv[6]
            ref = torch.relu(torch.relu(h1) @ lv[2].t() + lv[3]) @ lv[4] + lv[5]
            lref = torch.nn.functional.binary_cross_entropy_with_logits(ref, y.double(), pos_weight=pw.double())
            lref.backward()
            st = EdgeStructure(ei.to(dev), n)
            gl = [t.clone().to(dev).requires_grad_(True) for t in (P, Q, W2, b2, w3, b3, cv)]
            loss, logits = PF.decoder_loss(gl[0], gl[1], st, extra.to(dev) if skip else None, gl[6] if skip else None,
                                           gl[2], gl[3], gl[4], gl[5], y.to(dev), pw.to(dev), e)
            loss.backward()
            gl2 = [t.clone().to(dev).requires_grad_(True) for t in (P, Q, W2, b2, w3, b3, cv)]
            out2 = PF.decoder_mlp(gl2[0], gl2[1], st, extra.to(dev) if skip else None, gl2[6] if skip else None,
                                  gl2[2], gl2[3], gl2[4], gl2[5])
            torch.nn.functional.binary_cross_entropy_with_logits(out2, y.to(dev), pos_weight=pw.to(dev)).backward()
            errs = [f"logit {float((logits.cpu().double() - ref).abs().max()):.1e}",
                    f"loss {abs(float(loss) - float(lref)):.1e}"]
            for i, name in enumerate(["P", "Q", "W2", "b2", "w3", "b3", "cvec"]):
                if name == "cvec" and not skip:
                    continue
                rg = lv[i].grad
                sc = float(rg.abs().max()) + 1e-30
                r1 = float((gl[i].grad.cpu().double() - rg).abs().max()) / sc
                r2 = float((gl2[i].grad.cpu().double() - rg).abs().max()) / sc
                worst = max(worst, r1, r2)
                flag = " <<<<" if max(r1, r2) > 1e-4 else ""
                errs.append(f"{name} {r1:.1e}/{r2:.1e}{flag}")
            print(f"E={e} sorted={srt} skip={skip}: " + " ".join(errs), flush=True)
print("worst relative gradient error", worst)
