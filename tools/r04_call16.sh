#!/bin/bash
# round 4, call 16: cache policy of the read-once streams (depth, intensity) of the row walk
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r04_call17_aux_taps.txt; : > $out
for rep in 1 2; do
for v in s2 s2a2 s2a2b2 s2a1 s2a3b1; do
  DFX_LIB=$PWD/gpurun_build/libdfx_$v.so BATCH_ONLY=1 REPS=40 WARM=200 TAG="$v" timeout 200 python tools/r04_small_ops.py 2>&1 | grep -v "amdgpu.ids\|blocking" >> $out
done
done
cat $out
