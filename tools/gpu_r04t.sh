#!/bin/bash
# round-4 call t: tree-shaped workgroup finishes of the S / T kernels: decoder + mini-batch tests, mini-batch benches, kernel
# sequence of the fresh step, and the cfg-4 kernels same-box against the library before the change
set -o pipefail
O=gpurun_out/r04t; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return $rc; }
run 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "decoder or S_and_T or padded or replayed_fresh or hip_graph_replay or minibatch or subgraph or fused_loss or alternate_gcn or linear or first_layer or fused_embedding or first_dense or folded" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 6 $O/tests.log
[ $rc -eq 0 ] || exit 1
run 300 python bench.py --workload cfg2mb_fresh --steps 400 > $O/bench_cfg2mb_fresh.json 2> $O/bench_cfg2mb_fresh.err; echo "fresh bench rc=$?"; cut -c1-330 $O/bench_cfg2mb_fresh.json
run 300 python bench.py --workload cfg2mb --steps 400 > $O/bench_cfg2mb.json 2> $O/bench_cfg2mb.err; echo "mb bench rc=$?"; cut -c1-330 $O/bench_cfg2mb.json
run 300 python tools/time_decoder_ab.py > $O/ab_new.txt 2>&1; echo "ab new rc=$?"
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_fresh
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_fresh -o p --output-format csv -- python3 $ROOT/bench.py --workload cfg2mb_fresh --steps 200 > $ROOT/$O/fresh_under_rocprof.json 2> $ROOT/$O/fresh_rocprof.log
rc=$?; cd $ROOT; echo "rocprof rc=$rc"
python tools/step_kernel_sequence.py /tmp/prof_fresh > $O/fresh_step_sequence.txt 2>&1; cat $O/fresh_step_sequence.txt | cut -c1-150
