#!/bin/bash
# same-box A/B of decoder-kernel build variants: tools/gpu_ab_variants.sh <out-tag> <variant> [<variant> ...]
# (build_variants/libpangnn_hip_<variant>.so; "intree" = the product library); each variant timed twice, interleaved
set -o pipefail
tag=$1; shift
O=gpurun_out/$tag; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return $rc; }
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = intree ]; then lib=""; else lib=$PWD/build_variants/libpangnn_hip_$v.so; fi
    PANGNN_HIP_LIB=$lib run 300 python tools/time_decoder_ab.py > $O/ab_${v}_$rep.txt 2>&1 || { echo "variant $v failed"; tail -n 5 $O/ab_${v}_$rep.txt; exit 1; }
    grep -h " S:\|loss" $O/ab_${v}_$rep.txt | sed "s/^/[$v $rep] /" | cut -c1-150
  done
done
if [ -n "$TEST_VARIANT" ]; then
  PANGNN_HIP_LIB=$PWD/build_variants/libpangnn_hip_$TEST_VARIANT.so run 600 python -m pytest tests -m gpu -x -q -k "decoder or S_and_T" > $O/tests_$TEST_VARIANT.log 2>&1; echo "tests($TEST_VARIANT) rc=$?"; tail -n 3 $O/tests_$TEST_VARIANT.log
fi
