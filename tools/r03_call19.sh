#!/bin/bash
# Round 3, GPU call 19 (run at the commit "Reduction tail inside the step kernel", reverted since: the library no longer reads DFX_FOLD_TAIL): tests, then A/B against the tail kernel.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03s; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tail_assemble.py tests/test_gpu_deferred_tail.py tests/test_gpu_valid0_shadow.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_bf16x3.py tests/test_gpu_window.py tests/test_gpu_comm.py tests/test_gpu_factors.py -x -q -m gpu > $OUT/pytest_focus.txt 2>&1
echo "pytest focus exit $?"; tail -3 $OUT/pytest_focus.txt
show() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);r=d['roofline'];print('$2', round(d['value']), 'ms', round(d['ms_per_step']*1e3,1), 'kernel', round(r['kernel_us'],1), round(r['kernel_us_min'],1), round(r['kernel_us_max'],1), 'frac', round(r['frac'],4), 'gap', round(d['ms_per_step']*1e3-r['kernel_us'],1))" 2>&1 | tail -1; }
for r in 1 2 3; do
  for v in fold kernel; do
    E=1; [ $v = kernel ] && E=0
    DFX_FOLD_TAIL=$E timeout 200 python bench.py --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_${v}_$r.json 2> $OUT/bench_${v}_$r.err
    show $OUT/bench_${v}_$r.json "$v $r"
  done
done
for v in fold kernel; do
  E=1; [ $v = kernel ] && E=0
  DFX_FOLD_TAIL=$E timeout 400 python bench.py --no-cpu-baseline --no-traffic > $OUT/bench_full_$v.json 2> $OUT/bench_full_$v.err
  python - <<P
import json
d=json.loads(open('$OUT/bench_full_$v.json').read().strip().splitlines()[-1])
c=d['configs']
print('$v', 'value', round(d['value']), 'cs64', round(c['configs4_1280x960_cs64']['kernel_us'],1), round(c['configs4_1280x960_cs64']['evals_per_s']), 'pyr', round(c['configs1_pyramid3_128pairs']['one_launch_kernel_us'],1), round(c['configs1_pyramid3_128pairs']['evals_per_s']), 'lin', round(c['configs2_linearize_16kf_120pairs']['round_us'],1), 'win', c.get('configs3_window64',{}).get('ms_per_step'))
P
done
