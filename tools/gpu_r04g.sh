#!/bin/bash
# round-4 call g: bf16-output propagate (config 5) — its tests, the bf16 / config-5 tests, cfg5slice timing
set -o pipefail
O=gpurun_out/r04g; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return $rc; }
run 900 python -m pytest tests -m gpu -x -q -k "bf16 or config5 or autocast or cfg5 or dispatcher or opcheck or compiled" > $O/tests_bf16.log 2>&1; rc=$?; echo "bf16 tests rc=$rc"; tail -n 25 $O/tests_bf16.log
run 600 python bench.py --workload cfg5slice --steps 8 --no-cpu-baseline > $O/bench_cfg5slice.json 2> $O/bench_cfg5slice.err; echo "cfg5slice rc=$?"; cut -c1-330 $O/bench_cfg5slice.json
