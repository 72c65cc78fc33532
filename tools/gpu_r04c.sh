#!/bin/bash
# round-4 third GPU call: padded collation + ReplayedFreshStep tests, fresh mini-batch bench lines, then the whole GPU suite
set -o pipefail
O=gpurun_out/r04c; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return $rc; }
run 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "padded or replayed_fresh or hip_graph_replay" > $O/tests_fresh.log 2>&1; rc=$?; echo "fresh tests rc=$rc"; tail -n 25 $O/tests_fresh.log
if [ $rc -eq 0 ]; then
  run 300 python bench.py --workload cfg2mb_fresh --steps 400 > $O/bench_cfg2mb_fresh.json 2> $O/bench_cfg2mb_fresh.err; echo "fresh bench rc=$?"; cut -c1-400 $O/bench_cfg2mb_fresh.json; tail -n 5 $O/bench_cfg2mb_fresh.err
  PANGNN_FRESH_REPLAY=0 run 300 python bench.py --workload cfg2mb_fresh --steps 400 > $O/bench_cfg2mb_fresh_eager.json 2> $O/bench_cfg2mb_fresh_eager.err; echo "eager fresh bench rc=$?"; cut -c1-300 $O/bench_cfg2mb_fresh_eager.json
  run 300 python bench.py --workload cfg2mb --steps 400 > $O/bench_cfg2mb.json 2> $O/bench_cfg2mb.err; echo "replay bench rc=$?"; cut -c1-300 $O/bench_cfg2mb.json
fi
run 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -n 12 $O/tests.log
