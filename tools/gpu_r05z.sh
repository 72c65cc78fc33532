#!/bin/bash
# round 5, call z: ordered kernel launches of ONE step of emulated rank 0 of 2 (cfg 4)
set -o pipefail
O=$PWD/gpurun_out/r05z; mkdir -p $O
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/emu_trace
timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/emu_trace -o t --output-format csv -- python3 $ROOT/bench.py --emulate-rank 0 --of 2 --steps 6 --warmup 2 > $O/emu.json 2> $O/emu.err || { tail -20 $O/emu.err; exit 1; }
cd $ROOT
python tools/step_kernel_sequence.py /tmp/emu_trace gen_linear_fwd_reg_kernel > $O/rank0_of2_step_sequence.txt 2>&1 || { tail $O/rank0_of2_step_sequence.txt; exit 1; }
tail -1 $O/rank0_of2_step_sequence.txt
