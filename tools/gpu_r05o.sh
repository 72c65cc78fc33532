#!/bin/bash
# round 5, call o: whole GPU suite on the 3-wave split-bf16 128 x 128 layer + the layer's timing
set -o pipefail
O=gpurun_out/r05o; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1 || { tail -40 $O/tests.log | cut -c1-220; exit 1; }
tail -3 $O/tests.log
timeout -k 10 300 python tools/time_linear.py > $O/time_linear.txt 2>&1 || { tail -20 $O/time_linear.txt; exit 1; }
grep -v Warning $O/time_linear.txt | tail -22
