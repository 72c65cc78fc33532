#!/bin/bash
# round 5, call w (final tree: float16 decoder tables, joint P | Q in the partitioned decoder, shared curve): evidence set of the round's code — whole GPU suite, smoke, default bench line, rocprofv3 kernel stats of the
# same command, FETCH / WRITE counter passes -> profiles/traffic.json (stamped with the kernel-source hash and $PANGNN_HEAD)
set -o pipefail
O=gpurun_out/r05w; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return $rc; }
run 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log | cut -c1-200; exit 1; }
tail -2 $O/gpu_tests.log
run 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
run 900 bash tools/pmc_traffic.sh $O r05w > $O/traffic.log 2>&1 || { tail -20 $O/traffic.log; exit 1; }
tail -4 $O/traffic.log | cut -c1-200
cp profiles/traffic.json $O/traffic.json
run 600 python bench.py > $O/bench_cfg4.json 2> $O/bench_cfg4.err || { tail -20 $O/bench_cfg4.err; exit 1; }
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r05w/bench_cfg4.json").read())
print("headline", round(l["ms_per_step"], 3), "ms", round(l["value"] / 1e9, 3), "G edges/s; frac", round(l["roofline"]["frac"], 3),
      "traffic", l["roofline"]["traffic"])
for k in ("reference_loop", "reference_loop_literal", "eval", "general_features", "strict_fp32", "cfg5slice", "cfg2mb", "cfg2mb_fresh"):
    v = l.get(k, {})
    print(" ", k, round(v.get("ms_per_step", v.get("ms_per_pass", -1)), 3), v.get("error", ""))
print("  cpu_baseline", l.get("cpu_baseline", {}).get("value"))
PY
run 600 bash tools/profile_bench.sh $O r05w > $O/profile.log 2>&1 || { tail -20 $O/profile.log; exit 1; }
head -8 $O/r05w_bench_cfg4_kernel_stats.csv | cut -c1-150
