#!/bin/bash
# round-4 call p: third product of the S kernel on P1's h1 terms (A = m2 g_e as three terms): decoder tests first, same-box
# A/B against the library before the change (build_variants/libpangnn_hip_r04n.so), then the suite and the bench line
set -o pipefail
O=gpurun_out/r04p; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return $rc; }
run 600 python -m pytest tests -m gpu -x -q -k "decoder or S_and_T" > $O/tests_decoder.log 2>&1; rc=$?; echo "decoder tests rc=$rc"; tail -n 6 $O/tests_decoder.log
[ $rc -ne 0 ] && exit $rc
run 300 python tools/time_decoder_ab.py > $O/ab_new.txt 2>&1; echo "ab new rc=$?"
PANGNN_HIP_LIB=$PWD/build_variants/libpangnn_hip_r04n.so run 300 python tools/time_decoder_ab.py > $O/ab_r04n.txt 2>&1; echo "ab r04n rc=$?"
run 300 python tools/time_decoder_ab.py > $O/ab_new2.txt 2>&1
tail -n 4 $O/ab_r04n.txt $O/ab_new.txt $O/ab_new2.txt
run 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -n 8 $O/tests.log
run 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json
