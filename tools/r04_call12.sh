#!/bin/bash
# round 4, call 12: does the alignment of the 640 image buffers matter (channel camping)?  every image staggered inside its allocation
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r04_call12_stagger.txt; : > $out
for st in "" 64 1024 16; do
  STAGGER=$st BATCH_ONLY=1 REPS=40 WARM=200 TAG="stagger=${st:-none}" timeout 200 python tools/r04_small_ops.py 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
