#!/bin/bash
# Round 3, GPU call 16: small operators: two-stage walk written out twice (no state copies), SLP vectoriser off for dfx_misc_kernels.hip.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03p; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker.py tests/test_gpu_vs_ref.py tests/test_golden_ref_vectors.py tests/test_gpu_fullsize.py tests/test_gpu_convergence.py tests/test_gpu_cpp_shim.py tests/test_gpu_window.py -x -q -m gpu > $OUT/pytest_focus.txt 2>&1
echo "pytest focus exit $?"; tail -3 $OUT/pytest_focus.txt
for r in 1 2; do
for v in new slp nopipe; do
  L=""; [ $v != new ] && L=gpurun_build/libdfx_$v.so
  DFX_LIB=${L:+$PWD/$L} timeout 400 python bench.py --no-cpu-baseline --no-traffic > $OUT/bench_full_${v}_$r.json 2> $OUT/bench_full_${v}_$r.err
  python - <<P
import json
d=json.loads(open('$OUT/bench_full_${v}_$r.json').read().strip().splitlines()[-1])
c=d['configs']
print('$v $r', 'value', round(d['value']), 'kernel', round(d['roofline']['kernel_us'],1), 'se3', round(c['se3_step_batch_128pairs']['us'],1), round(c['se3_step_batch_128pairs']['frac'],3), 'err', round(c['sfm_error_batch_128pairs']['us'],1), round(c['sfm_error_batch_128pairs']['frac'],3), 'dec', round(c['update_depth_batch_64kf']['us'],1), 'single', round(c['configs1_single_pair_blocking']['call_us'],1))
P
done; done
timeout 200 python tools/profile_tracker.py 2>&1 | tail -6
