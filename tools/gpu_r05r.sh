#!/bin/bash
# round 5, call r: the partitioned decoder on ONE joint P | Q product — its GPU tests, then the emulated ranks of the 8-way split
set -o pipefail
O=gpurun_out/r05r; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_dist_gpu.py tests/test_bench_ranks.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -40 $O/tests.log | cut -c1-220; exit 1; }
tail -2 $O/tests.log
for r in 0 3; do
  timeout -k 10 300 python bench.py --emulate-rank $r --of 8 --steps 10 --warmup 3 > $O/emu_$r.json 2> $O/emu_$r.err || { tail -20 $O/emu_$r.err; exit 1; }
  python -c "
import json,sys
l=json.loads(open('$O/emu_$r.json').read()); print('rank', l['emulated_rank'], round(l['ms_per_step'],3), 'ms; S', [round(x,3) for x in l['decoder_S_launch_ms']], 'T', round(l['decoder_T_ms_per_step'],3))"
done
