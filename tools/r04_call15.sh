#!/bin/bash
# round 4, call 15: where does the row walk fall from its cache-resident rate to its HBM rate?  128 pairs over 8 ... 128 distinct image sets
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r04_call15_distinct.txt; : > $out
for nd in 8 16 32 48 64 96 128; do
  DFX_LIB=$PWD/gpurun_build/libdfx_band.so DISTINCT=$nd BATCH_ONLY=1 REPS=40 WARM=200 TAG="distinct=$nd ($((nd*6)) MB)" timeout 200 python tools/r04_small_ops.py 2>&1 | grep -v "amdgpu.ids\|blocking" >> $out
done
cat $out
