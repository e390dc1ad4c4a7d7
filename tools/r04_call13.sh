#!/bin/bash
# round 4, call 13: single-state ("consume first") order of the row walk: 68 instead of 93 VGPRs for the SE3 step (7 instead of 5 waves per SIMD)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r04_call13_eatfirst.txt; : > $out
DFX_LIB=$PWD/gpurun_build/libdfx_efboth.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker.py tests/test_gpu_vs_ref.py -m gpu -x -q 2>&1 | tail -3 | tee -a $out
for rep in 1 2; do
for v in base efse3 efboth; do
  for wg in 24 32; do
  DFX_LIB=$PWD/gpurun_build/libdfx_$v.so DFX_BATCH_WGS_PER_CU=$wg REPS=40 WARM=200 TAG="$v wg$wg" timeout 200 python tools/r04_small_ops.py 2>&1 | grep -v amdgpu.ids >> $out
  done
done
done
cat $out
