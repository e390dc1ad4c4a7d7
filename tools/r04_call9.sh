#!/bin/bash
# round 4, call 10: is the row walk bound by memory or inside the CU?  128 pairs over 128 / 8 / 1 distinct image sets
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r04_call10_resident.txt; : > $out
for v in nowin win; do
for nd in 128 8 1; do
  DFX_LIB=$PWD/gpurun_build/libdfx_$v.so DISTINCT=$nd BATCH_ONLY=1 REPS=40 WARM=200 TAG="$v distinct=$nd" timeout 200 python tools/r04_small_ops.py 2>&1 | grep -v amdgpu.ids >> $out
done
done
cat $out
