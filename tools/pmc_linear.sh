#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes) of the node-level linear kernels under `python bench.py`.
# usage (GPU box, repo root): bash tools/pmc_linear.sh <out_dir> <tag>
set -e
OUT=${1:-gpurun_out/linear}; TAG=${2:-r02}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcl_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmcl_$c -o t --output-format csv -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-extras > "$ROOT/$OUT/pass_$c.log" 2>&1
  f=$(find /tmp/pmcl_$c -name '*counter_collection.csv' | head -1)
  { head -1 "$f"; grep -E "linear_fwd_kernel|linear_wgrad_kernel" "$f" || true; } > "$ROOT/$OUT/${TAG}_pmc_linear_$(echo $c | tr A-Z a-z).csv"
done
