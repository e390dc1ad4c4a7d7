#!/bin/bash
# round 4, call 11: item order of the row walk (segment-major: a workgroup = 4 adjacent bands; band-major: a workgroup = 4 segments of one band)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r04_call11_order.txt; : > $out
for rep in 1 2; do
for v in segmajor bandmajor; do
  DFX_LIB=$PWD/gpurun_build/libdfx_$v.so BATCH_ONLY=1 REPS=40 WARM=200 TAG="$v" timeout 200 python tools/r04_small_ops.py 2>&1 | grep -v amdgpu.ids >> $out
done
done
cat $out
