#!/bin/bash
# round 5, call j: full-size tests + bench rank rehearsals on the C++ graph ops
set -o pipefail
mkdir -p gpurun_out/r05j
python -m pytest tests -q -m gpu -k "full_size" tests/test_hip_parity.py > gpurun_out/r05j/full_size.log 2>&1 || { tail -60 gpurun_out/r05j/full_size.log | cut -c1-220; exit 1; }
tail -3 gpurun_out/r05j/full_size.log
python -m pytest tests/test_bench_ranks.py tests/test_dist_gpu.py -q -m gpu > gpurun_out/r05j/ranks.log 2>&1 || { tail -60 gpurun_out/r05j/ranks.log | cut -c1-220; exit 1; }
tail -3 gpurun_out/r05j/ranks.log
