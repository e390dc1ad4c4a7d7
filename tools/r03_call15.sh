#!/bin/bash
# Round 3, GPU call 15: SE3 step / EvaluateError with the tap loads one pixel ahead (two-stage walk): parity, then A/B of the batched and the single-pair forms.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03o; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker.py tests/test_gpu_vs_ref.py tests/test_golden_ref_vectors.py tests/test_gpu_fullsize.py tests/test_gpu_convergence.py tests/test_gpu_cpp_shim.py -x -q -m gpu > $OUT/pytest_focus.txt 2>&1
echo "pytest focus exit $?"; tail -3 $OUT/pytest_focus.txt
for r in 1 2; do
for v in new nopipe; do
  L=""; [ $v != new ] && L=gpurun_build/libdfx_$v.so
  DFX_LIB=${L:+$PWD/$L} timeout 400 python bench.py --no-cpu-baseline --no-traffic > $OUT/bench_full_${v}_$r.json 2> $OUT/bench_full_${v}_$r.err
  python - <<P
import json
d=json.loads(open('$OUT/bench_full_${v}_$r.json').read().strip().splitlines()[-1])
c=d['configs']
print('$v $r', 'value', round(d['value']), 'kernel', round(d['roofline']['kernel_us'],1), 'se3', round(c['se3_step_batch_128pairs']['us'],1), round(c['se3_step_batch_128pairs']['frac'],3), 'err', round(c['sfm_error_batch_128pairs']['us'],1), round(c['sfm_error_batch_128pairs']['frac'],3), 'single', round(c['configs1_single_pair_blocking']['call_us'],1))
P
done; done
make -C tests/cpp latency_bench > /dev/null 2>&1
for v in new nopipe; do
  L=""; [ $v != new ] && L=gpurun_build/libdfx_$v.so
  echo "== latency $v"; if [ -n "$L" ]; then LD_PRELOAD=$PWD/$L tests/cpp/latency_bench 2>&1 | tail -12; else tests/cpp/latency_bench 2>&1 | tail -12; fi
done
timeout 300 python tools/tracker_bench.py 2>/dev/null | tail -5
