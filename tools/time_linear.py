"""Times the node-level dense kernels of the cfg-4 step at N = 1e6 through the product API (the library under
PANGNN_HIP_LIB if set) and checks them against an fp64 evaluation on a slice.  usage: python tools/time_linear.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangnn_amd import functional as PF  # noqa: E402
from pangnn_amd.graph import EdgeStructure  # noqa: E402

dev = torch.device("cuda:0")
n = 1_000_000
torch.manual_seed(0)


def timed(fn, reps=12):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("library:", os.environ.get("PANGNN_HIP_LIB", "(in-tree)"))
for k, m in ((64, 128), (128, 64), (64, 64), (128, 128)):
    x = torch.randn(n, k, device=dev)
    w = (torch.randn(m, k, device=dev) / 8).requires_grad_(True)
    b = torch.randn(m, device=dev).requires_grad_(True)
    g = torch.randn(n, m, device=dev)
    xr = x.clone().requires_grad_(True)
    for act in (0, 1):
        t = timed(lambda: PF.linear(x, w, b, act))
        y = PF.linear(x[:4096], w, b, act)
        xd = torch.nn.functional.elu(x[:4096].double()) if act else x[:4096].double()
        err = float((y.double() - (xd @ w.double().T + b.double())).abs().max())
        print(f"fwd<{k},{m}> act={act}: {t:.3f} ms   max |err| vs fp64 {err:.2e}")
    y = PF.linear(xr, w, b, 1)
    t = timed(lambda: torch.autograd.grad(y, (xr, w, b), g, retain_graph=True))
    print(f"bwd<{k},{m}> act=1 (dgrad with gate + wgrad): {t:.3f} ms")
    if (k, m) == (128, 128):        # what the layer cost through the library (torch -> hipBLASLt) until round 5
        F = torch.nn.functional
        t = timed(lambda: F.linear(F.elu(x), w, b))
        print(f"torch fwd<{k},{m}> (elu kernel + addmm): {t:.3f} ms")
        yt = F.linear(F.elu(xr), w, b)
        t = timed(lambda: torch.autograd.grad(yt, (xr, w, b), g, retain_graph=True))
        print(f"torch bwd<{k},{m}>: {t:.3f} ms")

# the generated first layer + conv_out's dense part
ei = torch.stack([torch.arange(n, device=dev), torch.arange(n, device=dev)])
st = EdgeStructure(ei, n)
norm = st.gcn_norm(None)
xt = torch.randn(n, 1, device=dev)
P = lambda *s: (torch.randn(*s, device=dev) * 0.3).requires_grad_(True)   # noqa: E731
we, be, win, bin_, wout = P(64, 1), P(64), P(128, 64), P(128), P(64, 128)
t = timed(lambda: PF.embed_conv_in_linear(xt, we, be, win, bin_, wout, None, st, norm))
print(f"embed_conv_in_linear fwd<128,64>: {t:.3f} ms")
y = PF.embed_conv_in_linear(xt, we, be, win, bin_, wout, None, st, norm)
g = torch.randn(n, 64, device=dev)
t = timed(lambda: torch.autograd.grad(y, (we, be, win, bin_, wout), g, retain_graph=True))
print(f"embed_conv_in_linear bwd<128,64> (wgrad + dgrad sums): {t:.3f} ms")
