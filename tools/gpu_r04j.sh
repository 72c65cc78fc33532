#!/bin/bash
# round-4 call j: the GPU suite after the C++ pangnn::linear (the test that spied on the Python twin fixed)
set -o pipefail
O=gpurun_out/r04j; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return $rc; }
run 1100 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -n 8 $O/tests.log
