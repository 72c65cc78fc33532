#!/bin/bash
# rocprofv3 --kernel-trace --stats of `python bench.py` (one command, one summary) -> <out>/<tag>_bench_cfg4_kernel_stats.csv
# + the bench line printed under the profiler.  usage (GPU box, repo root): bash tools/profile_bench.sh <out_dir> <tag>
set -e
OUT=${1:-gpurun_out/prof}; TAG=${2:-r02}
ROOT=$(pwd); mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o p --output-format csv -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$ROOT/$OUT/${TAG}_bench_cfg4_under_rocprof.json" 2> "$ROOT/$OUT/${TAG}_rocprof.log"
f=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)
cp "$f" "$ROOT/$OUT/${TAG}_bench_cfg4_kernel_stats.csv"
head -12 "$ROOT/$OUT/${TAG}_bench_cfg4_kernel_stats.csv" | cut -c1-160
