#!/bin/bash
# Round 3, GPU call 6: the restructured small operators (decoder fast path, SE3 / EvaluateError pixel walk) against the suite and in the bench.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03f; mkdir -p $OUT
export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
timeout 900 python bench.py --no-traffic --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03f/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print('value',d['value'],'ms',d['ms_per_step'],'kernel',r['kernel_us'],r['kernel_us_min'],r['kernel_us_max'],'frac',r['frac'])
for k,v in d.get('configs',{}).items(): print(k, json.dumps({a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a!='note'})[:300])
PY
timeout 120 python tools/profile_tracker.py 2>&1 | tail -5
