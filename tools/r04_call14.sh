#!/bin/bash
# round 4, call 14: linear (row-major chunk) walk vs band walk: does the DRAM access pattern set the 4.5 TB/s both reductions run at?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r04_call14_linear.txt; : > $out
DFX_LIB=$PWD/gpurun_build/libdfx_linear.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker.py tests/test_gpu_vs_ref.py tests/test_golden_ref_vectors.py -m gpu -x -q 2>&1 | tail -3 | tee -a $out
for rep in 1 2; do
for v in band linear linear_ef linear_d2; do
  DFX_LIB=$PWD/gpurun_build/libdfx_$v.so REPS=40 WARM=200 TAG="$v" timeout 200 python tools/r04_small_ops.py 2>&1 | grep -v "amdgpu.ids\|blocking" >> $out
done
done
DFX_LIB=$PWD/gpurun_build/libdfx_linear.so DISTINCT=1 BATCH_ONLY=1 REPS=40 WARM=200 TAG="linear distinct=1" timeout 200 python tools/r04_small_ops.py 2>&1 | grep -v "amdgpu.ids\|blocking" >> $out
cat $out
