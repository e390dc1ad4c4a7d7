#!/bin/bash
# Round 3, GPU call 20: confirmation on the final tree: whole GPU suite, smoke, default bench line.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03t; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03t/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value',round(d['value']),'ms',round(d['ms_per_step'],4),'kernel',round(r['kernel_us'],1),round(r['kernel_us_min'],1),round(r['kernel_us_max'],1),'frac',round(r['frac'],4),'traffic x',round(r['traffic']/r['algorithmic_bytes_per_launch'],4) if r['traffic'] else None)
PY
