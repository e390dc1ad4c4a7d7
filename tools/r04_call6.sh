#!/bin/bash
# round 4, call 6: the restructured row walk (branch-free warm-up, lane-value validity, exact vmcnt) -- parity, then kernel durations per
# pipeline depth / unroll under rocprofv3 --kernel-trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker.py tests/test_gpu_vs_ref.py -m gpu -x -q 2>&1 | tail -3
out=gpurun_out/r04_call6_ktrace.txt; : > $out
for v in s1e1u1 s2e2u1 s1e3u1 s1e1u2 s2e2u2; do
  rm -rf /tmp/kt_$v
  DFX_LIB=$PWD/gpurun_build/libdfx_$v.so REPS=40 timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$v -o kt -- python tools/small_ops_driver.py > /tmp/kt_$v.log 2>&1
  f=$(find /tmp/kt_$v -name '*kernel_trace.csv' | head -1)
  echo "== $v" >> $out
  python tools/kt_summary.py $f --last 20 | grep -E "se3_step_batch|sfm_error_batch" | awk -F, '{print $1, "last20 avg us", $8}' >> $out
done
cat $out
TAG=default timeout 300 python tools/r04_small_ops.py 2>&1 | grep -v amdgpu.ids | tee -a $out
