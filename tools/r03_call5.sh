#!/bin/bash
# Round 3, GPU call 5: pairs of several image sizes in one launch, the exchange step on real RCCL (world of one), the whole suite, and the
# full default bench line (secondary configurations, small-operator rooflines, PMC traffic, CPU baseline).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03e; mkdir -p $OUT
export TMPDIR=/tmp
t0=$(date +%s); lap() { echo "== $1 @ $(( $(date +%s) - t0 )) s"; }
timeout 700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_gpu.log; lap suite
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03e/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print('value',d['value'],'ms',d['ms_per_step'],'kernel',r['kernel_us'],r['kernel_us_min'],r['kernel_us_max'],'frac',r['frac'],'traffic',r['traffic'], r['traffic']/r['algorithmic_bytes_per_launch'] if r['traffic'] else None)
print(r['kernel'], '|', r['schedule'], '|', r['mfma'][:60])
for k,v in d.get('configs',{}).items(): print(k, json.dumps({a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a!='note'})[:420])
print('cpu', json.dumps(d.get('cpu_baseline'))[:400])
PY
lap bench
