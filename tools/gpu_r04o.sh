#!/bin/bash
# round-4 call o: target rows of the next half tile issued before the first product (-DPANGNN_D16_Q_EARLY) — same-box A/B
set -o pipefail
O=gpurun_out/r04o; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return $rc; }
for k in 1 2; do
  PANGNN_HIP_LIB=$PWD/build_variants/libpangnn_hip_qearly.so run 300 python tools/time_decoder_ab.py > $O/ab_qe$k.txt 2>&1; echo "qe rc=$?"
  run 300 python tools/time_decoder_ab.py > $O/ab_base$k.txt 2>&1; echo "base rc=$?"
done
grep -h "S:\|loss" $O/ab_qe1.txt $O/ab_base1.txt $O/ab_qe2.txt $O/ab_base2.txt | sed 's/\[.*build_variants./[/'
