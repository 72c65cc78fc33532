#!/bin/bash
# round 5: emulated rank 0 of 2 three times (is the step time stable?), band path on / off
set -o pipefail
O=gpurun_out/r05r; mkdir -p $O
for band in 1 0 1 0; do
  PANGNN_DIST_BAND=$band timeout -k 10 300 python bench.py --emulate-rank 0 --of 2 --steps 20 --warmup 5 > $O/emu2_$band.json 2> $O/emu2_$band.err || { tail -20 $O/emu2_$band.err; exit 1; }
  python -c "
import json
l=json.loads(open('$O/emu2_$band.json').read()); s=l['decoder_S_launch_ms']; t=l['decoder_T_ms_per_step']
print('band', $band, round(l['ms_per_step'],3), 'ms; S', [round(x,3) for x in s], 'T', round(t,3), '-> rest', round(l['ms_per_step']-sum(s)-t,3))"
done
