#!/bin/bash
# round 4, call 5: ablations of the row-walk kernels (DFX_RW_ABLATE) under rocprofv3 --kernel-trace: which component holds the time
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r04_call5_ablate.txt; : > $out
for v in base a1 a2 a4 a8 a16 a9 a25; do
  rm -rf /tmp/kt_$v
  DFX_LIB=$PWD/gpurun_build/libdfx_$v.so REPS=40 timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$v -o kt -- python tools/small_ops_driver.py > /tmp/kt_$v.log 2>&1
  f=$(find /tmp/kt_$v -name '*kernel_trace.csv' | head -1)
  echo "== $v" >> $out
  python tools/kt_summary.py $f --last 20 | grep -E "se3_step_batch|sfm_error_batch" | awk -F, '{print $1, "last20 avg us", $8}' >> $out
done
cat $out
