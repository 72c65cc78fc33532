#!/bin/bash
# L2-fabric traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes) of the
# step's main kernels, from `python bench.py --steps 3`.  usage (GPU box, repo root): bash tools/pmc_traffic.sh <out_dir> <tag>
set -e
OUT=${1:-gpurun_out/traffic}; TAG=${2:-r02}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o t --output-format csv -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-siblings > "$ROOT/$OUT/pass_$c.log" 2>&1
  f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
  { head -1 "$f"; grep -E "decoder_train16|decoder_dgrad16|spmm_row_kernel<64, 4, false, 0>|decoder_infer16" "$f" || true; } > "$ROOT/$OUT/${TAG}_pmc_$(echo $c | tr A-Z a-z).csv"
done
cd "$ROOT"
python3 tools/update_traffic.py "$OUT" "$TAG"
