#!/bin/bash
# round 5, call s: ordered kernel launches of ONE emulated-rank step (rank 3 of 8, cfg 4) with the idle gaps in front of them
set -o pipefail
O=$PWD/gpurun_out/r05s; mkdir -p $O
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/emu_trace
timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/emu_trace -o t --output-format csv -- python3 $ROOT/bench.py --emulate-rank 3 --of 8 --steps 6 --warmup 2 > $O/emu3.json 2> $O/emu3.err || { tail -20 $O/emu3.err; exit 1; }
cd $ROOT
python tools/step_kernel_sequence.py /tmp/emu_trace gen_linear_fwd_reg_kernel > $O/rank3_step_sequence.txt 2>&1 || { tail $O/rank3_step_sequence.txt; exit 1; }
tail -3 $O/rank3_step_sequence.txt
