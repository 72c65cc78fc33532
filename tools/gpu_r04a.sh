#!/bin/bash
# round-4 first GPU call: decoder kernel A/B (round-3 library vs this tree), gradient distances, GPU tests, bench
set -o pipefail
O=gpurun_out/r04a; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return $rc; }
run 300 python tools/time_decoder_ab.py > $O/ab_new.txt 2>&1; echo "ab new rc=$?"
PANGNN_HIP_LIB=$PWD/build_variants/libpangnn_hip_r03.so run 300 python tools/time_decoder_ab.py > $O/ab_r03.txt 2>&1; echo "ab r03 rc=$?"
run 300 python tools/time_decoder_ab.py > $O/ab_new2.txt 2>&1
tail -n 4 $O/ab_r03.txt $O/ab_new.txt $O/ab_new2.txt
run 600 python tools/grad_distances.py $O/grad_fp64.jsonl > $O/grad_distances.txt 2>&1; echo "grad rc=$?"; tail -n 12 $O/grad_distances.txt
run 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -n 15 $O/tests.log
run 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json | cut -c1-600
