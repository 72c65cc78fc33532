#!/bin/bash
# round 5, call b: bench.py with the reference_loop / eval siblings — small graph first, then the default line
set -o pipefail
mkdir -p gpurun_out/r05b
python bench.py --genes 1000 --reference-loop --no-cpu-baseline --steps 5 > gpurun_out/r05b/bench_small.json 2> gpurun_out/r05b/bench_small.err || { tail -30 gpurun_out/r05b/bench_small.err; exit 1; }
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r05b/bench_small.json").read())
for k in ("reference_loop", "reference_loop_literal", "eval", "extras_error"):
    print(k, json.dumps(l.get(k))[:600])
PY
python bench.py > gpurun_out/r05b/bench_cfg4.json 2> gpurun_out/r05b/bench_cfg4.err || { tail -30 gpurun_out/r05b/bench_cfg4.err; exit 1; }
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r05b/bench_cfg4.json").read())
print("headline ms", l["ms_per_step"], "value", l["value"])
for k in ("reference_loop", "reference_loop_literal", "eval", "general_features", "extras_error"):
    print(k, json.dumps(l.get(k))[:900])
PY
