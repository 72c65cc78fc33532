#!/bin/bash
# round 5, call a: the new accelerate-loop / deferred-logits tests, then the whole GPU suite
set -o pipefail
mkdir -p gpurun_out/r05a
python -m pytest tests/test_accelerate_loop.py tests/test_deferred.py -x -q > gpurun_out/r05a/accel.log 2>&1 || { tail -60 gpurun_out/r05a/accel.log; exit 1; }
tail -3 gpurun_out/r05a/accel.log
python -m pytest tests -x -q -m gpu > gpurun_out/r05a/gpu_all.log 2>&1 || { tail -60 gpurun_out/r05a/gpu_all.log; exit 1; }
tail -3 gpurun_out/r05a/gpu_all.log
