#!/bin/bash
# Round 3, GPU call 18 (run at commit 909774f, reverted since: the library no longer reads DFX_FOLD): the second pass of the reduction operators folded into the first: tests, latencies, tracker.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03r; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fold.py tests/test_gpu_parity.py tests/test_gpu_tracker.py tests/test_gpu_vs_ref.py tests/test_golden_ref_vectors.py tests/test_gpu_fullsize.py tests/test_gpu_convergence.py tests/test_gpu_cpp_shim.py tests/test_gpu_factors.py tests/test_keyframe_store.py -x -q -m gpu > $OUT/pytest_focus.txt 2>&1
echo "pytest focus exit $?"; tail -3 $OUT/pytest_focus.txt
make -C tests/cpp latency_bench > /dev/null 2>&1
for r in 1 2; do for f in 1 0; do
  echo "== DFX_FOLD=$f run $r"; DFX_FOLD=$f tests/cpp/latency_bench 2>&1 | grep "mean" | tee $OUT/latency_fold${f}_$r.txt
  DFX_FOLD=$f timeout 200 python tools/profile_tracker.py 2>&1 | grep "ms per frame" | tee $OUT/tracker_fold${f}_$r.txt
done; done
for f in 1 0; do
  DFX_FOLD=$f timeout 400 python bench.py --no-cpu-baseline --no-traffic > $OUT/bench_full_fold$f.json 2> $OUT/bench_full_fold$f.err
  python - <<P
import json
d=json.loads(open('$OUT/bench_full_fold$f.json').read().strip().splitlines()[-1])
c=d['configs']
print('fold=$f', 'value', round(d['value']), 'kernel', round(d['roofline']['kernel_us'],1), 'se3', round(c['se3_step_batch_128pairs']['us'],1), 'err', round(c['sfm_error_batch_128pairs']['us'],1), 'single', round(c['configs1_single_pair_blocking']['call_us'],1))
P
done
