#!/bin/bash
# round 5, call n2: the 128 x 128 layer timed at N = 1e6, in-tree (f32-MFMA forward, 4 waves) against -DPANGNN_LIN_128_W3 (split-bf16
# forward, 3 waves per workgroup) — the variant library linked with -soname libpangnn_hip.so so that libpangnn_torch.so binds to it too
set -o pipefail
O=gpurun_out/r05n; mkdir -p $O
V=$PWD/build_variants/libpangnn_hip_l128w3.so
for rep in 1 2; do
  timeout -k 10 300 python tools/time_linear.py > $O/time_intree_$rep.txt 2>&1 || { tail -20 $O/time_intree_$rep.txt; exit 1; }
  PANGNN_HIP_LIB=$V timeout -k 10 300 python tools/time_linear.py > $O/time_l128w3_$rep.txt 2>&1 || { tail -20 $O/time_l128w3_$rep.txt; exit 1; }
done
grep -h "128,128" $O/time_intree_1.txt $O/time_intree_2.txt; echo ---; grep -h "128,128" $O/time_l128w3_1.txt $O/time_l128w3_2.txt
PANGNN_HIP_LIB=$V timeout -k 10 600 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "linear or dgrad or union" > $O/tests_l128w3.log 2>&1; echo "tests(l128w3) rc=$?"; tail -3 $O/tests_l128w3.log
