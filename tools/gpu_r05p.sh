#!/bin/bash
# round 5, call p: the float16 row format — its tests, then the whole GPU suite (every 16-bit entry point was touched), then the
# propagate / dense kernels timed (the f32 / bf16 instances must not have moved)
set -o pipefail
O=gpurun_out/r05p; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_f16_rows.py tests/test_accelerate_loop.py -q -m gpu -x -s > $O/f16.log 2>&1 || { tail -60 $O/f16.log | cut -c1-240; exit 1; }
grep -h "fp16 loop" $O/f16.log; tail -2 $O/f16.log
timeout -k 10 1000 python -m pytest tests -q -m gpu -x --deselect tests/test_f16_rows.py --deselect tests/test_accelerate_loop.py > $O/tests.log 2>&1 || { tail -40 $O/tests.log | cut -c1-220; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python tools/time_linear.py > $O/time_linear.txt 2>&1 || { tail -20 $O/time_linear.txt; exit 1; }
grep -E "^fwd|^bwd|^embed" $O/time_linear.txt
timeout -k 10 300 python tools/time_propagate_bf16.py > $O/time_propagate.txt 2>&1 || { tail -20 $O/time_propagate.txt; exit 1; }
tail -12 $O/time_propagate.txt
