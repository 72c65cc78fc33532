#!/bin/bash
# round-4 call k: suite + smoke + bench lines after the Python twins of linear / bce_with_logits were deleted
set -o pipefail
O=gpurun_out/r04k; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return $rc; }
run 1100 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -n 8 $O/tests.log
run 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $O/smoke.log
run 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-260 $O/bench.json
