#!/bin/bash
# round-4 call r: ordered kernel sequence of one replayed fresh mini-batch step (which launches it is made of)
set -o pipefail
O=gpurun_out/r04r; mkdir -p $O
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_fresh
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_fresh -o p --output-format csv -- python3 $ROOT/bench.py --workload cfg2mb_fresh --steps 200 > $ROOT/$O/fresh_under_rocprof.json 2> $ROOT/$O/fresh_rocprof.log
rc=$?; cd $ROOT; echo "rocprof rc=$rc"
python tools/step_kernel_sequence.py /tmp/prof_fresh > $O/fresh_step_sequence.txt 2>&1; cat $O/fresh_step_sequence.txt | cut -c1-160
