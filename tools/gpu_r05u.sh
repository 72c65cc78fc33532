#!/bin/bash
# round 5, call u: validation pass with ROC-AUC and PR-AUC sharing one sorted curve
set -o pipefail
O=gpurun_out/r05u; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_metrics.py tests/test_hip_parity.py -q -m gpu -x -k "metric or evaluate or confusion or auc" > $O/tests.log 2>&1 || { tail -40 $O/tests.log | cut -c1-200; exit 1; }
tail -2 $O/tests.log
timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python -c "
import json
l=json.loads(open('$O/bench.json').read()); print('headline', round(l['ms_per_step'],3), 'eval', l['eval'])"
