#!/bin/bash
# round-4 second GPU call: S trims A/B (r04a library = T changes only), GPU tests, bench, forced-exchange bench over RCCL
set -o pipefail
O=gpurun_out/r04b; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return $rc; }
run 300 python tools/time_decoder_ab.py > $O/ab_new.txt 2>&1; echo "ab new rc=$?"
PANGNN_HIP_LIB=$PWD/build_variants/libpangnn_hip_r04a.so run 300 python tools/time_decoder_ab.py > $O/ab_r04a.txt 2>&1; echo "ab r04a rc=$?"
run 300 python tools/time_decoder_ab.py > $O/ab_new2.txt 2>&1
tail -n 4 $O/ab_r04a.txt $O/ab_new.txt $O/ab_new2.txt
run 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -n 15 $O/tests.log
run 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json
PANGNN_FORCE_DIST=1 PANGNN_FORCE_EXCHANGE=1 run 600 python bench.py --no-cpu-baseline > $O/bench_forced_exchange.json 2> $O/bench_forced_exchange.err; echo "forced rc=$?"; cut -c1-300 $O/bench_forced_exchange.json
