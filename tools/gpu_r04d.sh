#!/bin/bash
# round-4 fourth GPU call: padded fresh step after spreading the padding; kernel breakdown of the replayed fresh step
set -o pipefail
O=gpurun_out/r04d; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return $rc; }
run 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "padded or replayed_fresh or hip_graph_replay or minibatch or subgraph" > $O/tests_fresh.log 2>&1; rc=$?; echo "fresh tests rc=$rc"; tail -n 25 $O/tests_fresh.log
[ $rc -eq 0 ] || exit 1
run 300 python bench.py --workload cfg2mb_fresh --steps 400 > $O/bench_cfg2mb_fresh.json 2> $O/bench_cfg2mb_fresh.err; echo "fresh bench rc=$?"; cut -c1-330 $O/bench_cfg2mb_fresh.json
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_fresh
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_fresh -o p --output-format csv -- python3 $ROOT/bench.py --workload cfg2mb_fresh --steps 200 > $ROOT/$O/fresh_under_rocprof.json 2> $ROOT/$O/fresh_rocprof.log
cd $ROOT
f=$(find /tmp/prof_fresh -name '*kernel_stats.csv' | head -1); cp "$f" $O/fresh_kernel_stats.csv
python tools/kernel_counts.py /tmp/prof_fresh 224 --by-time > $O/fresh_kernel_counts.txt 2>&1; head -n 45 $O/fresh_kernel_counts.txt | cut -c1-170
