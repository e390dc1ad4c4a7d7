#!/bin/bash
# Round 3, GPU call 8: deferred tail with the launch stream at the higher priority vs the tail in order.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03h; mkdir -p $OUT
export TMPDIR=/tmp
for r in 1 2 3; do for v in inorder deferred; do
  F=""; [ $v = deferred ] && F="--deferred-tail"
  timeout 200 python bench.py $F --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_${v}_$r.json 2> $OUT/bench_${v}_$r.err; python -c "
import json;d=json.loads(open('$OUT/bench_${v}_$r.json').read().strip().splitlines()[-1]);r=d['roofline'];print('$v $r', round(d['value']), round(d['ms_per_step']*1e3,1), round(r['kernel_us'],1), round(r['kernel_us_min'],1), round(r['kernel_us_max'],1), 'gap', round(d['ms_per_step']*1e3-r['kernel_us'],1))"
done; done
