#!/bin/bash
# Round 3, GPU call 21: workgroups per pair for the bf16-split step kernel.  It runs three workgroups per CU (147 registers), i.e. 768 at a time; the launch
# shape was tuned on the fp32 chain (four per CU: 40 per pair x 128 pairs = 5.0 rounds of 1024).  36 / 42 / 48 per pair are 6.0 / 7.0 / 8.0 rounds of 768.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03u; mkdir -p $OUT
export TMPDIR=/tmp
show() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);r=d['roofline'];print('$2', round(d['value']), 'ms', round(d['ms_per_step']*1e3,1), 'kernel', round(r['kernel_us'],1), round(r['kernel_us_min'],1), round(r['kernel_us_max'],1), 'frac', round(r['frac'],4), 'gap', round(d['ms_per_step']*1e3-r['kernel_us'],1))" 2>&1 | tail -1; }
for r in ${RUNS:-1 2}; do
  for b in ${BLOCKS:-0 30 36 42 48 54 60}; do
    timeout 200 python bench.py --step-blocks $b --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_b${b}_$r.json 2> $OUT/bench_b${b}_$r.err
    show $OUT/bench_b${b}_$r.json "blocks=$b run $r"
  done
done
