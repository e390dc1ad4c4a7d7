#!/bin/bash
# round 4, call 8: taps out of per-wave LDS windows: parity, then kernel durations vs the gather-only build, by workgroups per CU
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker.py tests/test_gpu_vs_ref.py -m gpu -x -q 2>&1 | tail -12
out=gpurun_out/r04_call8_ktrace.txt; : > $out
for v in win nowin; do
 for wg in 24 12; do
  rm -rf /tmp/kt_$v
  DFX_LIB=$PWD/gpurun_build/libdfx_$v.so DFX_BATCH_WGS_PER_CU=$wg REPS=40 timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$v -o kt -- python tools/small_ops_driver.py > /tmp/kt_$v.log 2>&1
  f=$(find /tmp/kt_$v -name '*kernel_trace.csv' | head -1)
  echo "== $v wg/CU $wg" >> $out
  python tools/kt_summary.py $f --last 20 | grep -E "se3_step_batch|sfm_error_batch" | awk -F, '{print $1, "last20 avg us", $8}' >> $out
 done
done
cat $out
TAG=default timeout 300 python tools/r04_small_ops.py 2>&1 | grep -v amdgpu.ids | tee -a $out
