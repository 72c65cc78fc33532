#!/bin/bash
# round 5, call h: the graph-reading ops in C++ — dispatcher tests first, then the whole GPU suite
set -o pipefail
mkdir -p gpurun_out/r05h
python -m pytest tests/test_torch_ops.py -x -q > gpurun_out/r05h/torch_ops.log 2>&1 || { tail -80 gpurun_out/r05h/torch_ops.log; exit 1; }
tail -3 gpurun_out/r05h/torch_ops.log
python -m pytest tests -x -q -m gpu > gpurun_out/r05h/gpu_all.log 2>&1 || { tail -80 gpurun_out/r05h/gpu_all.log; exit 1; }
tail -3 gpurun_out/r05h/gpu_all.log
