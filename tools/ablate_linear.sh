#!/bin/bash
# Diagnostic only (not product code): builds linear.hip variants and times the node-level kernels at N = 1e6.
# Usage on the GPU box: bash tools/ablate_linear.sh
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/ablate_lin; mkdir -p $OUT
for v in base nont; do
  flags=""
  [ $v = nont ] && flags="-DPANGNN_LIN_NO_NT"
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -shared pangnn_amd/csrc/linear.hip pangnn_amd/csrc/edge_ops.hip -o $OUT/liblin_$v.so
done
python - <<'PY'
import ctypes as C, torch
dev = torch.device('cuda')
n = 1_000_000
P = C.c_void_p
bufs = {k: [torch.randn(n, k, device=dev) for _ in range(3)] for k in (64, 128)}
for v in ["base", "nont"]:
    lib = C.CDLL(f'gpurun_out/ablate_lin/liblin_{v}.so')
    for (k, m) in ((64, 128), (128, 64), (64, 64)):
        w = torch.randn(m, k, device=dev) / 8
        b = torch.randn(m, device=dev)
        ys = [torch.empty(n, m, device=dev) for _ in range(3)]
        def run(i):
            x = bufs[k][i % 3]; y = ys[i % 3]
            rc = lib.pangnn_linear_fwd_f32(P(x.data_ptr()), C.c_int64(k), P(w.data_ptr()), P(b.data_ptr()), P(y.data_ptr()),
                                           C.c_int64(m), C.c_int64(n), C.c_int32(k), C.c_int32(m), None)
            assert rc == 0
        for i in range(3): run(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(12): run(i)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 12
        print(f'{v:8s} fwd<{k},{m}> {ms:.3f} ms  {n*(k+m)*4/ms/1e6:.0f} GB/s  {2*n*k*m/ms/1e9:.1f} TF')
PY
