#!/bin/bash
# Round 3, GPU call 14: a fourth wave per SIMD for the bf16-split step kernel (two-part epilogue fold: 20 KB of LDS per workgroup; half-size operand ring).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03n; mkdir -p $OUT
export TMPDIR=/tmp
for v in w4hr w4; do
  DFX_LIB=$PWD/gpurun_build/libdfx_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16x3.py tests/test_gpu_configs.py -x -q -m gpu > $OUT/pytest_$v.txt 2>&1
  echo "pytest $v exit $?"; tail -2 $OUT/pytest_$v.txt
done
show() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);r=d['roofline'];print('$2', round(d['value']), 'ms', round(d['ms_per_step']*1e3,1), 'kernel', round(r['kernel_us'],1), round(r['kernel_us_min'],1), round(r['kernel_us_max'],1), 'frac', round(r['frac'],4), 'gap', round(d['ms_per_step']*1e3-r['kernel_us'],1))" 2>&1 | tail -1; }
for r in 1 2 3; do
  for v in new w4hr w4 ep2 hr; do
    L=""; [ $v != new ] && L=gpurun_build/libdfx_$v.so
    DFX_LIB=${L:+$PWD/$L} timeout 200 python bench.py --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_${v}_$r.json 2> $OUT/bench_${v}_$r.err
    show $OUT/bench_${v}_$r.json "$v $r"
  done
done
for v in new w4hr w4; do
  L=""; [ $v != new ] && L=gpurun_build/libdfx_$v.so
  echo "== $v"
  DFX_LIB=${L:+$PWD/$L} timeout 120 python tools/idle_gap_probe.py --idle-us 0 --seconds 2.5 2>&1 | grep "^idle\|sclk" | cut -c1-330 | head -3
done
