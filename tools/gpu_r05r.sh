#!/bin/bash
# round 5, call r: the partitioned decoder's joint launches — its GPU tests, then emulated ranks of the 8-way and the 2-way split
set -o pipefail
O=gpurun_out/r05r; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_dist_gpu.py tests/test_bench_ranks.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -40 $O/tests.log | cut -c1-220; exit 1; }
tail -2 $O/tests.log
for spec in "0 8" "3 8" "0 2"; do
  set -- $spec
  timeout -k 10 300 python bench.py --emulate-rank $1 --of $2 --steps 10 --warmup 3 > $O/emu_$1_$2.json 2> $O/emu_$1_$2.err || { tail -20 $O/emu_$1_$2.err; exit 1; }
  python -c "
import json,sys
l=json.loads(open('$O/emu_$1_$2.json').read()); s=l['decoder_S_launch_ms']; t=l['decoder_T_ms_per_step']
print('rank', l['emulated_rank'], 'of', l['of'], round(l['ms_per_step'],3), 'ms; S', [round(x,3) for x in s], 'T', round(t,3), '-> rest', round(l['ms_per_step']-sum(s)-t,3))"
done
