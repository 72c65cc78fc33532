#!/bin/bash
# round-4 call i: C++ pangnn::linear (implementation + autograd) — dispatcher tests first, then the suite, smoke, bench
set -o pipefail
O=gpurun_out/r04i; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return $rc; }
run 900 python -m pytest tests/test_torch_ops.py -m gpu -x -q > $O/tests_ops.log 2>&1; rc=$?; echo "ops tests rc=$rc"; tail -n 30 $O/tests_ops.log
[ $rc -eq 0 ] || exit 1
run 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -n 6 $O/tests.log
run 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $O/smoke.log
run 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-260 $O/bench.json
run 400 python tools/time_graphed_whole.py > $O/graphed_whole.txt 2>&1; echo "graphed rc=$?"; tail -n 6 $O/graphed_whole.txt
