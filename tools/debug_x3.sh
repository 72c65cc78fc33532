#!/bin/bash
# Diagnostic: rebuild the library with a flag and run tools/debug_x3.py
set -e
cd "$(dirname "$0")/.."
for v in vgprform base; do
  flags=""
  [ $v = norelu ] && flags="-DPANGNN_NO_ASM_RELU"
  [ $v = vgprform ] && flags="-mllvm -amdgpu-mfma-vgpr-form=1"
  (cd pangnn_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -c decoder.hip -o decoder.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC spmm.o edge_ops.o graph_build.o decoder.o linear.o -o ../libpangnn_hip.so)
  echo "== $v"; python tools/debug_x3b.py 2>&1 | cut -c1-60; python tools/debug_x3b.py 2>&1 | cut -c1-60 | grep -c "rows 1"; python tools/time_decoder_modes.py 2>&1 | tail -1
done
