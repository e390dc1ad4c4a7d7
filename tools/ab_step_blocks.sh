#!/bin/bash
# Same-box sweep of the step kernel's workgroups per pair (bench.py --step-blocks B; 0 = the library's choice): tools/ab_step_blocks.sh OUT ROUNDS B ...
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=${1:-gpurun_out/ab_step_blocks}; R=${2:-2}; shift 2; mkdir -p $O
for r in $(seq 1 $R); do
  for b in "$@"; do
    echo -n "step_blocks $b round $r: "
    timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-configs --steps 30 --warmup 15 --step-blocks $b 2> /dev/null < /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step', round(d['ms_per_step'],4), 'kernel_us', round(r['kernel_us'],1), 'min', round(r['kernel_us_min'],1), 'frac', round(r['frac'],4), 'value', round(d['value']))"
  done
done > $O/ab_step_blocks.txt 2>&1
cat $O/ab_step_blocks.txt
