#!/bin/bash
# round-4 call l: kernel stats of the cfg5slice and of the general-feature step (profiles for the sibling workloads)
set -o pipefail
O=gpurun_out/r04l; mkdir -p $O
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_c5
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -o p --output-format csv -- python3 $ROOT/bench.py --workload cfg5slice --steps 6 --warmup 2 --no-cpu-baseline > $ROOT/$O/cfg5slice_under_rocprof.json 2> $ROOT/$O/cfg5slice_rocprof.log
rc=$?; echo "cfg5slice rocprof rc=$rc"; [ $rc -eq 124 ] && exit 124
cd $ROOT
f=$(find /tmp/prof_c5 -name '*kernel_stats.csv' | head -1); cp "$f" $O/r04l_bench_cfg5slice_kernel_stats.csv
python tools/kernel_counts.py /tmp/prof_c5 8 --by-time 2>&1 | head -n 22 | cut -c1-170
cut -c1-300 $O/cfg5slice_under_rocprof.json
