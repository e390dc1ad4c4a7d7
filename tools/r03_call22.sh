#!/bin/bash
# Round 3, GPU call 22: 30 vs 40 workgroups per pair (40 vs 30 chunks per wave) for the bf16 split at other batch sizes and at CS = 16.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03v; mkdir -p $OUT
export TMPDIR=/tmp
show() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);r=d['roofline'];print('$2', round(d['value']), 'ms', round(d['ms_per_step']*1e3,1), 'kernel', round(r['kernel_us'],1), round(r['kernel_us_min'],1), round(r['kernel_us_max'],1), 'frac', round(r['frac'],4), 'gap', round(d['ms_per_step']*1e3-r['kernel_us'],1))" 2>&1 | tail -1; }
for r in 1 2; do
  for cfg in "64 0 32" "64 30 32" "256 0 32" "256 30 32" "48 0 32" "48 30 32" "128 0 16" "128 30 16"; do
    set -- $cfg
    timeout 300 python bench.py --pairs $1 --step-blocks $2 --cs $3 --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_p$1_b$2_cs$3_$r.json 2> $OUT/bench_p$1_b$2_cs$3_$r.err
    show $OUT/bench_p$1_b$2_cs$3_$r.json "pairs=$1 blocks=$2 cs=$3 run $r"
  done
done
