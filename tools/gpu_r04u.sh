#!/bin/bash
# round-4 call u: S / T kernel durations on mini-batch-sized lists (where do 25 us go?)
set -o pipefail
O=gpurun_out/r04u; mkdir -p $O
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_small
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof_small -o p --output-format csv -- python3 $ROOT/tools/probe_decoder_small.py > $ROOT/$O/probe.log 2>&1
rc=$?; cd $ROOT; echo "rocprof rc=$rc"; tail -n 3 $O/probe.log
python tools/probe_decoder_small_report.py /tmp/prof_small | tee $O/decoder_small.txt
