"""Diagnostic: time the training decoder kernel (pangnn_decoder_mlp_loss_f32) in both matrix-pipe modes on a
cfg-4-like source-sorted edge list.  python tools/time_decoder_modes.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangnn_amd import functional as PF
from pangnn_amd.graph import EdgeStructure
dev = torch.device("cuda")
torch.manual_seed(0)
n, e = 1_000_000, 74_000_000
gen = torch.Generator(device=dev).manual_seed(0)
src = torch.arange(e, device=dev) // 74
dst = (src // 50000 * 50000 + 50000 + torch.randint(0, 50000, (e,), device=dev, generator=gen)).clamp_(max=n - 1)
st = EdgeStructure(torch.stack([src, dst]).contiguous(), n)
pq = torch.randn(n, 128, device=dev, requires_grad=True)
par = [torch.randn(64, 64, device=dev) / 8, torch.randn(64, device=dev), torch.randn(64, device=dev),
       torch.randn(1, device=dev)]
y = (torch.rand(e, device=dev) < 0.03).float()
pw = torch.tensor(30.0, device=dev)
for mode in (0, 1, 0, 1):
    PF.DECODER_PRECISION = mode
    PF.KERNEL_TIMER = {"dec.bwd": []}
    for _ in range(4):
        leaves = [p.clone().requires_grad_(True) for p in par]
        loss, logits = PF.decoder_loss_pq(pq, st, None, None, *leaves, y, pw, e)
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in PF.KERNEL_TIMER["dec.bwd"]][1:]
    print("mode", mode, "kernel ms", sum(ts) / len(ts), "loss", float(loss))
    PF.KERNEL_TIMER = {"dec.fwd": []}
    with torch.no_grad():
        for _ in range(4):
            out = PF.decoder_mlp_pq(pq, st, None, None, *par)
    torch.cuda.synchronize()
    tf = [a.elapsed_time(b) for a, b in PF.KERNEL_TIMER["dec.fwd"]][1:]
    print("mode", mode, "inference kernel ms", sum(tf) / len(tf), "max |logit - training logit|", float((out - logits).abs().max()))
