#!/bin/bash
# Diagnostic only: times the bf16 matrix-pipe training decoder with parts removed (results are wrong by design).
set -e
cd "$(dirname "$0")/.."
cp pangnn_amd/libpangnn_hip.so /tmp/libpangnn_hip.so.keep
for v in base nop3 nostore norunsum all; do
  flags=""
  [ $v = nop3 ] && flags="-DPANGNN_X3_ABL_P3"
  [ $v = nostore ] && flags="-DPANGNN_X3_ABL_STORE"
  [ $v = norunsum ] && flags="-DPANGNN_X3_ABL_RUNSUM"
  [ $v = all ] && flags="-DPANGNN_X3_ABL_P3 -DPANGNN_X3_ABL_STORE -DPANGNN_X3_ABL_RUNSUM"
  (cd pangnn_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 $flags -c decoder.hip -o /tmp/decoder_abl.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC spmm.o edge_ops.o graph_build.o /tmp/decoder_abl.o linear.o -o ../libpangnn_hip.so)
  echo "== $v: $(python tools/time_decoder_modes.py 2>&1 | tail -1)"
done
cp /tmp/libpangnn_hip.so.keep pangnn_amd/libpangnn_hip.so
