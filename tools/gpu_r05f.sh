#!/bin/bash
# round 5, call f: per-rank compute of the 8-way split on one GPU (cfg 4 first; cfg 5 in its own call)
set -o pipefail
O=gpurun_out/r05f; mkdir -p $O
timeout -k 10 ${2:-900} python tools/rank_emulation.py --workloads ${1:-cfg4} --out $O/rank_emulation_${1:-cfg4}.json 2>&1 | tee $O/rank_emulation_${1:-cfg4}.log | cut -c1-220
