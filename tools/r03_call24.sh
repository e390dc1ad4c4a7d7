#!/bin/bash
# Round 3, GPU call 24: workgroups per pair at CS = 64, 1280x960, 16 pairs (two workgroups per CU: 512 at a time; the library's choice is 160 per pair = 30 chunks per wave).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03x; mkdir -p $OUT
export TMPDIR=/tmp
show() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);r=d['roofline'];print('$2', round(d['value']), 'ms', round(d['ms_per_step']*1e3,1), 'kernel', round(r['kernel_us'],1), round(r['kernel_us_min'],1), round(r['kernel_us_max'],1), 'frac', round(r['frac'],4), 'gap', round(d['ms_per_step']*1e3-r['kernel_us'],1))" 2>&1 | tail -1; }
for r in 1 2; do
  for b in ${BLOCKS:-0 96 120 128 144 192 224 256}; do
    timeout 300 python bench.py --pairs 16 --width 1280 --height 960 --cs 64 --step-blocks $b --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_b${b}_$r.json 2> $OUT/bench_b${b}_$r.err
    show $OUT/bench_b${b}_$r.json "cs64 blocks=$b run $r"
  done
done
