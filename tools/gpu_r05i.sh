#!/bin/bash
# round 5, call i: the remainder of the GPU suite after the first failure of call h (same code + the observer fix)
set -o pipefail
mkdir -p gpurun_out/r05i
python -m pytest tests -q -m gpu --deselect tests/test_bench_ranks.py -k "not full_size" > gpurun_out/r05i/gpu_rest.log 2>&1; rc=$?
tail -25 gpurun_out/r05i/gpu_rest.log | cut -c1-200
exit $rc
