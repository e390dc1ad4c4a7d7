#!/bin/bash
# round 4, call 23: kernel durations of the reductions after the dword taps (kernel trace), and the launch shape again
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r04_call23_ktrace.txt; : > $out
for pose in true ident; do
  rm -rf /tmp/kt_$pose
  POSE=$pose REPS=40 timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$pose -o kt -- python tools/small_ops_driver.py > /tmp/kt_$pose.log 2>&1
  f=$(find /tmp/kt_$pose -name '*kernel_trace.csv' | head -1)
  echo "== SE3 pose $pose" >> $out
  python tools/kt_summary.py $f --last 20 | grep -E "se3_step_batch|sfm_error_batch|finalize_rows" | awk -F, '{print $1, "last20 avg us", $8}' >> $out
done
for wg in 12 16 24 32 48; do
  DFX_BATCH_WGS_PER_CU=$wg TAG=wg$wg BATCH_ONLY=1 timeout 100 python tools/r04_small_ops.py 2>&1 | grep -v "Warn\|amdgpu.ids" >> $out
done
cat $out
