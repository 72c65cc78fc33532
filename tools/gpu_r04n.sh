#!/bin/bash
# round-4 call n: Q-row prefetch in the S kernel — same-box A/B against the same code without it, then the decoder tests
set -o pipefail
O=gpurun_out/r04n; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return $rc; }
run 300 python tools/time_decoder_ab.py > $O/ab_pf.txt 2>&1; echo "pf rc=$?"
PANGNN_HIP_LIB=$PWD/build_variants/libpangnn_hip_nopf.so run 300 python tools/time_decoder_ab.py > $O/ab_nopf.txt 2>&1; echo "nopf rc=$?"
run 300 python tools/time_decoder_ab.py > $O/ab_pf2.txt 2>&1
PANGNN_HIP_LIB=$PWD/build_variants/libpangnn_hip_nopf.so run 300 python tools/time_decoder_ab.py > $O/ab_nopf2.txt 2>&1
grep -h "S:\|T:\|loss" $O/ab_pf.txt $O/ab_nopf.txt $O/ab_pf2.txt $O/ab_nopf2.txt | sed 's/\[.*build_variants./[/'
true
