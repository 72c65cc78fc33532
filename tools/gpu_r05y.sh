#!/bin/bash
# round 5, call y: compute side of the 2-way and 4-way splits of cfg 4 (every rank emulated in turn on ONE GPU), beside the 8-way
set -o pipefail
O=gpurun_out/r05y; mkdir -p $O
for w in 2 4; do
  timeout -k 10 500 python tools/rank_emulation.py --workloads cfg4 --of $w --out $O/rank_emulation_cfg4_of$w.json 2>&1 | tee $O/rank_emulation_cfg4_of$w.log | cut -c1-200 || exit 1
done
