#!/bin/bash
# round-4 call f: the GPU suite + smoke + bench with the dispatcher ops as the default route
set -o pipefail
O=gpurun_out/r04f; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return $rc; }
run 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -n 8 $O/tests.log
run 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $O/smoke.log
run 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-260 $O/bench.json
PANGNN_DISPATCHER_OPS=auto run 600 python bench.py --no-cpu-baseline --no-siblings > $O/bench_auto.json 2> $O/bench_auto.err; echo "bench auto rc=$?"; cut -c1-260 $O/bench_auto.json
