#!/bin/bash
# Diagnostic builds of libpangnn_hip.so for the SLP-packed-f32 miscompute of the S kernel ("what went
# wrong" 2): decoder16.hip WITH SLP vectorisation (the product builds it with -fno-slp-vectorize) in several variants,
# into build_variants/ (git-ignored, travels to the GPU box).  tools/slp_probe.py compares each variant's S kernel,
# bit for bit, with the product library's on one graph.    usage (repo root): bash tools/slp_probe.sh
set -e
ROOT=$(pwd); SRC=pangnn_amd/csrc; OUT=build_variants; mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result"
OBJS="$SRC/spmm.o $SRC/edge_ops.o $SRC/graph_build.o $SRC/decoder.o $SRC/linear.o"
make -C $SRC >/dev/null
build() {   # name, extra flags
  /opt/rocm/bin/hipcc $FLAGS $2 -c $SRC/decoder16.hip -o $OUT/decoder16_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $OUT/decoder16_$1.o -o $OUT/libpangnn_hip_$1.so
  /opt/rocm/bin/hipcc $FLAGS $2 -S --cuda-device-only $SRC/decoder16.hip -o $OUT/decoder16_$1.s 2>/dev/null
  echo "$1: $(grep -c 'v_pk_[a-z]*_f32' $OUT/decoder16_$1.s) v_pk_*_f32 in the ISA"
}
build noslp   "-fno-slp-vectorize"                                   # = the product object (control: must be bit-identical)
build slp     ""                                                     # SLP on: the build that miscomputed
build slp_nopk "-Xclang -target-feature -Xclang -packed-fp32-ops"    # SLP on, packed-f32 instructions off
build slp_gescalar "-DPANGNN_D16_PROBE_GE_SCALAR"                    # SLP on, g_e read as 4 dwords
build slp_wait0 "-DPANGNN_D16_PROBE_WAIT0"                           # SLP on, LDS operands landed + 16 idle cycles before the products
build slp_noprio "-DPANGNN_D16_PROBE_NOPRIO"                         # SLP on, no s_setprio
build slp_opq_epi "-DPANGNN_D16_PROBE_OPAQUE_EPI"                    # SLP on, the epilogue's products opaque (not packable)
build slp_opq_runsum "-DPANGNN_D16_PROBE_OPAQUE_RUNSUM"              # SLP on, the run sums' partial sums opaque
build slp_opq_gcv "-DPANGNN_D16_PROBE_OPAQUE_GCV"                    # SLP on, the skip-feature gradient accumulation opaque
build slp_onewave "-DPANGNN_D16_PROBE_ONE_WAVE"                      # SLP on, one wave per SIMD (256-thread workgroups)
build noslp_onewave "-fno-slp-vectorize -DPANGNN_D16_PROBE_ONE_WAVE" # control for the tile -> wave mapping of that build
build noslp_dbg "-fno-slp-vectorize -DPANGNN_D16_DEBUG"              # dL/dh1 rows dumped (tools/slp_probe_rows.py): control
build slp_dbg   "-DPANGNN_D16_DEBUG"                                 # dL/dh1 rows dumped, SLP on
