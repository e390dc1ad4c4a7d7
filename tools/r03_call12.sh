#!/bin/bash
# Round 3, GPU call 12: fewer vector-ALU instructions in the bf16-split step kernel (v_dot2c_f32_bf16 remainders, the three divisions by q.z
# through one refined reciprocal): exactness probe, parity tests, A/B against the builds without them.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03l; mkdir -p $OUT
export TMPDIR=/tmp
gpurun_build/bf16x3_probe > $OUT/bf16x3_probe.txt 2>&1; cat $OUT/bf16x3_probe.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16x3.py tests/test_gpu_configs.py tests/test_gpu_vs_ref.py tests/test_golden_ref_vectors.py tests/test_gpu_fullsize.py tests/test_gpu_tail_assemble.py tests/test_gpu_tracker.py tests/test_gpu_convergence.py -x -q -m gpu > $OUT/pytest_focus.txt 2>&1
echo "pytest focus exit $?"; tail -3 $OUT/pytest_focus.txt
show() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);r=d['roofline'];print('$2', round(d['value']), 'ms', round(d['ms_per_step']*1e3,1), 'kernel', round(r['kernel_us'],1), round(r['kernel_us_min'],1), round(r['kernel_us_max'],1), 'frac', round(r['frac'],4), 'gap', round(d['ms_per_step']*1e3-r['kernel_us'],1))" 2>&1 | tail -1; }
for r in 1 2 3; do
  for v in new base nodot2 nodiv; do
    L=""; [ $v != new ] && L=gpurun_build/libdfx_$v.so
    DFX_LIB=${L:+$PWD/$L} timeout 200 python bench.py --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_${v}_$r.json 2> $OUT/bench_${v}_$r.err
    show $OUT/bench_${v}_$r.json "$v $r"
  done
done
for v in new base; do
  L=""; [ $v != new ] && L=gpurun_build/libdfx_$v.so
  DFX_LIB=${L:+$PWD/$L} timeout 400 python bench.py --no-cpu-baseline --no-traffic > $OUT/bench_full_$v.json 2> $OUT/bench_full_$v.err
  python - <<P
import json
d=json.loads(open('$OUT/bench_full_$v.json').read().strip().splitlines()[-1])
c=d['configs']
print('$v', 'se3', round(c['se3_step_batch_128pairs']['us'],1), 'err', round(c['sfm_error_batch_128pairs']['us'],1), 'cs64', round(c['configs4_1280x960_cs64']['kernel_us'],1), 'pyr', round(c['configs1_pyramid3_128pairs']['one_launch_kernel_us'],1), 'lin', round(c['configs2_linearize_16kf_120pairs']['round_us'],1), 'win', c.get('configs3_window64',{}).get('ms_per_step'))
P
done
