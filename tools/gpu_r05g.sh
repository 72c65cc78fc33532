#!/bin/bash
# round 5, call g: what ONE emulated rank's step is made of (rank 3 of 8, cfg 4): kernel trace under rocprofv3, launches and
# kernel time per step against the step's wall time
set -o pipefail
O=$PWD/gpurun_out/r05g; mkdir -p $O
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_emu
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_emu -o p --output-format csv -- python3 "$ROOT/bench.py" --emulate-rank 3 --of 8 --steps 20 --warmup 3 --no-cpu-baseline > $O/emu3_under_rocprof.json 2> $O/rocprof.log || { tail -20 $O/rocprof.log; exit 1; }
cd $ROOT
f=$(find /tmp/prof_emu -name '*kernel_stats.csv' | head -1); cp "$f" $O/emu3_kernel_stats.csv
python tools/kernel_counts.py /tmp/prof_emu 23 --by-time | head -60 | tee $O/emu3_kernel_counts.txt | cut -c1-170
python -c "
import json; l=json.loads([x for x in open('$O/emu3_under_rocprof.json') if x.startswith('{')][-1]); print('ms/step under rocprof', l['ms_per_step'])"
