#!/bin/bash
# Round 3, GPU call 23: the 40-chunk cap as the library default (v = 0; first run of this script: forced by DFX_CPW_MAX=40, which also hit CS = 64 and the shared-Jacobian round) against DFX_CPW_MAX=30: tests, the bench line, the secondary configurations.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03w; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_bf16x3.py tests/test_gpu_tail_assemble.py tests/test_gpu_deferred_tail.py tests/test_gpu_window.py -x -q -m gpu > $OUT/pytest_focus.txt 2>&1
echo "pytest focus exit $?"; tail -2 $OUT/pytest_focus.txt
for r in 1 2; do for v in 0 30; do
  env $( [ $v != 0 ] && echo DFX_CPW_MAX=$v ) timeout 400 python bench.py --no-cpu-baseline --no-traffic > $OUT/bench_full_cpw${v}_$r.json 2> $OUT/bench_full_cpw${v}_$r.err
  python - <<P
import json
d=json.loads(open('$OUT/bench_full_cpw${v}_$r.json').read().strip().splitlines()[-1])
c=d['configs']; r=d['roofline']
print('cpw_max=$v run $r', 'value', round(d['value']), 'kernel', round(r['kernel_us'],1), 'gap', round(d['ms_per_step']*1e3-r['kernel_us'],1), 'pyr', round(c['configs1_pyramid3_128pairs']['one_launch_kernel_us'],1), round(c['configs1_pyramid3_128pairs']['evals_per_s']), 'lin', round(c['configs2_linearize_16kf_120pairs']['round_us'],1), 'win', round(c.get('configs3_window64',{}).get('ms_per_step'),3), 'cs64', round(c['configs4_1280x960_cs64']['kernel_us'],1))
P
done; done
