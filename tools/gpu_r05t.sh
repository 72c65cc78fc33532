#!/bin/bash
# round 5, call t: whole GPU suite + smoke + default bench line at the round's final tree
set -o pipefail
O=gpurun_out/r05t; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log | cut -c1-200; exit 1; }
tail -2 $O/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
/usr/bin/env bash -c 'SECONDS=0; timeout -k 10 600 python bench.py > '$O'/bench_cfg4.json 2> '$O'/bench_cfg4.err; rc=$?; echo "bench rc=$rc wall ${SECONDS}s"; exit $rc' || { tail -20 $O/bench_cfg4.err; exit 1; }
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r05t/bench_cfg4.json").read())
print("headline", round(l["ms_per_step"], 3), "ms", round(l["value"] / 1e9, 3), "G edges/s; frac", round(l["roofline"]["frac"], 3),
      "traffic", l["roofline"]["traffic"])
PY
