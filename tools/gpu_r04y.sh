#!/bin/bash
# round-4 final evidence call (after the workgroup-finish rewrite): default bench line, rocprofv3 kernel stats of the same command,
# kernels, FETCH_SIZE / WRITE_SIZE passes -> profiles/traffic.json
set -o pipefail
O=gpurun_out/r04y; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return $rc; }

run 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-260 $O/bench.json
run 600 bash tools/profile_bench.sh $O r04y > $O/profile_bench.log 2>&1; echo "profile rc=$?"; tail -n 12 $O/profile_bench.log | cut -c1-200
run 900 bash tools/pmc_decoder.sh $O/pmc_dec > $O/pmc_decoder.log 2>&1; echo "pmc_decoder rc=$?"; tail -n 70 $O/pmc_decoder.log | cut -c1-160
run 900 bash tools/pmc_traffic.sh $O r04y > $O/pmc_traffic.log 2>&1; echo "pmc_traffic rc=$?"; tail -n 20 $O/pmc_traffic.log | cut -c1-200
cp profiles/traffic.json $O/traffic.json
