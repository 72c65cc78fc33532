#!/bin/bash
# round 5, call p2: rest of the GPU suite after the autocast-policy test + kernel timings (f32 / bf16 instances must not have moved)
set -o pipefail
O=gpurun_out/r05p; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_torch_ops.py tests/test_bench_ranks.py tests/test_dist_gpu.py tests/test_construct.py tests/test_metrics.py tests/test_deferred.py -q -m gpu -x > $O/tests2.log 2>&1 || { tail -40 $O/tests2.log | cut -c1-220; exit 1; }
tail -2 $O/tests2.log
timeout -k 10 300 python tools/time_linear.py > $O/time_linear.txt 2>&1 || { tail -20 $O/time_linear.txt; exit 1; }
grep -E "^fwd|^bwd|^embed" $O/time_linear.txt
timeout -k 10 300 python tools/time_propagate_bf16.py > $O/time_propagate.txt 2>&1 || { tail -20 $O/time_propagate.txt; exit 1; }
tail -6 $O/time_propagate.txt
