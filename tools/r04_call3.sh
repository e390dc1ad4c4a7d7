#!/bin/bash
# round 4, call 3: kernel durations of the batched pixel reductions by rocprofv3 --kernel-trace (the event timings of call 2 did not move
# with the pipeline depth: host-bound enqueue?)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r04_call3_ktrace.txt; : > $out
for v in s1e1 s2e2 s3e3; do
  rm -rf /tmp/kt_$v
  DFX_LIB=$PWD/gpurun_build/libdfx_$v.so REPS=80 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$v -o kt -- python tools/small_ops_driver.py > /tmp/kt_$v.log 2>&1
  f=$(find /tmp/kt_$v -name '*kernel_trace.csv' | head -1)
  echo "== $v" >> $out
  python tools/kt_summary.py $f --last 30 | grep -E "kernel,|se3_step_batch|sfm_error_batch|finalize_rows" >> $out
done
cat $out
