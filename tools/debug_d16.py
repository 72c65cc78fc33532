"""Diagnostic for csrc/decoder16.hip (build it with -DPANGNN_D16_DEBUG to get the dL/dh1 dump): per-part errors of the S kernel's by-source run sums and relu-mask flips of
the records against torch.  python tools/debug_d16.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import random_graph
from pangnn_amd import functional as PF, _lib
from pangnn_amd.graph import EdgeStructure
dev = torch.device("cuda")
lib = _lib.load()


def run(e, srt, skip, n=97, seed=None):
    torch.manual_seed(e + skip if seed is None else seed)
    d = 64
    ei, w = random_graph(n, e, seed=e, isolated=0.0)
    if srt:
        ei = ei[:, torch.argsort(ei[0] * n + ei[1])]
    P, Q = torch.randn(n, d), torch.randn(n, d)
    W2, b2, w3, b3, cv = torch.randn(d, d) / 8, torch.randn(d), torch.randn(d), torch.randn(1), torch.randn(d)
    extra = (w / 40) if skip else None
    g = torch.randn(e)
    # fp64 reference of dL/dh1pre per edge
    Pd, Qd = P.double(), Q.double()
    h1p = Pd[ei[0]] + Qd[ei[1]]
    if skip:
        h1p = h1p + extra.double().unsqueeze(1) * cv.double()
    h1 = torch.relu(h1p)
    h2p = h1 @ W2.double().t() + b2.double()
    G = g.double().unsqueeze(1) * w3.double() * (h2p > 0)
    gh1 = (G @ W2.double()) * (h1p > 0)
    st = EdgeStructure(ei.to(dev), n)
    t = lambda x: None if x is None else x.to(dev).contiguous()
    p_, q_, w2_, b2_, w3_, b3_, cv_, ex_, g_ = map(t, (P, Q, W2, b2, w3, b3, cv if skip else None, extra, g))
    rec = torch.zeros(e, 8, dtype=torch.int32, device=dev)
    plan = st.runsum_plan(int(_lib.load().pangnn_decoder_chunk_tiles_for(st.num_edges)))
    parts = None if plan is None else torch.full((plan.n_parts, d), float("nan"), device=dev)
    gw2, gb2, gw3, gb3 = torch.empty_like(w2_), torch.empty_like(b2_), torch.empty_like(w3_), torch.empty_like(b3_)
    gcv = None if not skip else torch.empty_like(cv_)
    import ctypes
    dbg = torch.full((e, 64), float("nan"), device=dev)
    have_dbg = hasattr(lib, "pangnn_debug_set_v")
    if have_dbg:
        lib.pangnn_debug_set_v.argtypes = [ctypes.c_void_p]
        assert lib.pangnn_debug_set_v(dbg.data_ptr()) == 0
    wsb = lib.pangnn_decoder_train_workspace_bytes()
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    _lib.check(lib.pangnn_decoder_train_f32(p_.data_ptr(), 64, q_.data_ptr(), 64, n, st.edge_index.data_ptr(), e, e,
                                            _lib.ptr(ex_), _lib.ptr(cv_), w2_.data_ptr(), b2_.data_ptr(), w3_.data_ptr(),
                                            b3_.data_ptr(), 64, None, None, 0, g_.data_ptr(), None, None, rec.data_ptr(),
                                            _lib.ptr(parts), None if plan is None else plan.part_off.data_ptr(),
                                            gw2.data_ptr(), gw3.data_ptr(), gb3.data_ptr(), _lib.ptr(gcv), None,
                                            ws.data_ptr(), wsb, _lib.stream_ptr()), "train")
    torch.cuda.synchronize()
    msg = f"E={e} sorted={srt} skip={skip}:"
    # mask flips: m1 bits of the records vs fp64 sign
    r = rec.cpu()
    m1_ref = (h1p > 0)
    # bit for k = 32 ks + 8 g' + s in dword g': 16 (s & 1) + 15 - (4 ks + (s >> 1))   (round 4: m1 in the high byte of each half)
    k = torch.arange(64)
    ks, gq, s = k >> 5, (k >> 3) & 3, k & 7
    bit = 16 * (s & 1) + 15 - (4 * ks + (s >> 1))
    words = r[:, :4].long() & 0xffffffff
    m1 = ((words[:, gq] >> bit) & 1).bool()
    flips = (m1 != m1_ref).nonzero()
    msg += f" m1 flips {flips.shape[0]}"
    if flips.shape[0]:
        ee, kk = flips[0].tolist()
        msg += f" (first: edge {ee} k {kk} h1pre {float(h1p[ee, kk]):.3e})"
    ge = r[:, 4].view(torch.float32)
    msg += f" g_e err {float((ge - g).abs().max()):.1e}"
    if plan is not None:
        # expected parts
        src = ei[0]
        flags = (torch.arange(e) % 32) == 0
        flags[1:] |= src[1:] != src[:-1]
        pid = torch.cumsum(flags, 0) - 1
        exp = torch.zeros(plan.n_parts, d, dtype=torch.float64).index_add_(0, pid, gh1)
        got = parts.cpu().double()
        err = (got - exp).abs().max(1).values / (exp.abs().max() + 1e-30)
        bad = (err > 1e-4).nonzero().view(-1)
        nan = torch.isnan(got).any(1).nonzero().view(-1)
        msg += f" parts {plan.n_parts} bad {bad.numel()} nan {nan.numel()}"
        if bad.numel() or nan.numel():
            first_e = torch.searchsorted(pid, torch.cat([bad[:6], nan[:3]]))
            msg += " bad parts " + str(bad[:6].tolist()) + " nan parts " + str(nan[:3].tolist()) + \
                   " start at edge " + str(first_e.tolist()) + " (pos in tile " + str((first_e % 32).tolist()) + ")"
            # run lengths of the first bad part
            b0 = int(torch.cat([bad, nan])[0])
            msg += f" len {int((pid == b0).sum())} got/exp[0] {float(got[b0, 0]):.4e}/{float(exp[b0, 0]):.4e}"
    if have_dbg and (plan is not None or skip):
        dv = dbg.cpu().double()
        errm = (dv - gh1).abs() / (gh1.abs().max() + 1e-30)
        badv = (errm > 1e-5).nonzero()
        msg += f" v bad elements {badv.shape[0]} nan {int(torch.isnan(dv).sum())}"
        if badv.shape[0]:
            es = badv[:, 0].unique()
            msg += f" in {es.numel()} edges: " + str([(int(x), int(x) % 32) for x in es[:8]]) + " cols of first: " + \
                   str(badv[badv[:, 0] == es[0], 1].tolist()[:20])
            ee, kk = badv[0].tolist()
            msg += f" got {float(dv[ee, kk]):.4e} exp {float(gh1[ee, kk]):.4e} h1pre {float(h1p[ee, kk]):.3e}"
    if skip:
        gcv_ref = (extra.double().unsqueeze(1) * gh1).sum(0)
        msg += f" gcv err {float((gcv.cpu().double() - gcv_ref).abs().max() / gcv_ref.abs().max()):.1e}"
    print(msg, flush=True)


for e in (40, 100, 1000, 5000):
    run(e, True, False)
run(1000, True, True)
run(1000, False, True)
for e in (70001, 300007):
    run(e, False, True, n=5003)
    run(e, False, False, n=5003)
