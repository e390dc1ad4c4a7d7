#!/bin/bash
# Round 3, GPU call 13: v_dot2c_f32_bf16 split with the corrected selectors (probe, parity), A/B against the shift/mask/subtract form,
# deferred tail with the one-kernel tail, and where the power goes: step-kernel time and shader clock with parts of the kernel removed.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03m; mkdir -p $OUT
export TMPDIR=/tmp
gpurun_build/bf16x3_probe > $OUT/bf16x3_probe.txt 2>&1; cat $OUT/bf16x3_probe.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16x3.py tests/test_gpu_configs.py tests/test_gpu_vs_ref.py tests/test_golden_ref_vectors.py tests/test_gpu_fullsize.py tests/test_gpu_tail_assemble.py tests/test_gpu_tracker.py tests/test_gpu_convergence.py -x -q -m gpu > $OUT/pytest_focus.txt 2>&1
echo "pytest focus exit $?"; tail -3 $OUT/pytest_focus.txt
show() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);r=d['roofline'];print('$2', round(d['value']), 'ms', round(d['ms_per_step']*1e3,1), 'kernel', round(r['kernel_us'],1), round(r['kernel_us_min'],1), round(r['kernel_us_max'],1), 'frac', round(r['frac'],4), 'gap', round(d['ms_per_step']*1e3-r['kernel_us'],1))" 2>&1 | tail -1; }
for r in 1 2 3 4; do
  for v in new nodot2 deferred; do
    L=""; F=""
    [ $v = nodot2 ] && L=gpurun_build/libdfx_$v.so
    [ $v = deferred ] && F="--deferred-tail"
    DFX_LIB=${L:+$PWD/$L} timeout 200 python bench.py $F --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_${v}_$r.json 2> $OUT/bench_${v}_$r.err
    show $OUT/bench_${v}_$r.json "$v $r"
  done
done
for v in full abl1 abl2 abl4 abl6; do
  L=""; [ $v != full ] && L=gpurun_build/libdfx_$v.so
  echo "== $v"
  DFX_LIB=${L:+$PWD/$L} timeout 120 python tools/idle_gap_probe.py --idle-us 0 --seconds 2.5 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json,re
for l in sys.stdin:
    l=l.rstrip()
    if l.startswith('idle') or l.startswith('_sleep'): print(l)
    elif l.strip().startswith('{'):
        d=json.loads(l.strip()); print('   ', {k.split(' (')[0].replace(' clock speed:',''):v for k,v in d.items() if 'sclk clock speed' in k or 'Power' in k or 'junction' in k})
" | tee $OUT/ablate_$v.txt | head -6
done
