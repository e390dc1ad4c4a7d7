#!/bin/bash
# round 4, call 2: pipeline depth of the row walk (taps issued DT rows ahead) x workgroups per CU of the batched forms
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker.py tests/test_gpu_vs_ref.py -m gpu -x -q 2>&1 | tail -3
out=gpurun_out/r04_call2_depth.txt; : > $out
for rep in 1 2; do
for v in s1e1 s2e2 s3e3 s2e4; do
  for wg in 12 24; do
    DFX_LIB=$PWD/gpurun_build/libdfx_$v.so DFX_BATCH_WGS_PER_CU=$wg BATCH_ONLY=1 REPS=40 WARM=200 TAG="$v wg$wg" timeout 300 python tools/r04_small_ops.py 2>&1 | grep -v amdgpu.ids >> $out
  done
done
done
cat $out
TAG=default timeout 300 python tools/r04_small_ops.py 2>&1 | grep -v amdgpu.ids | tee -a $out
