#!/bin/bash
# Round 3, GPU call 3: diagnosis of the run-to-run differences of the four-product-diagonal build on batches with regions out of view.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03c; mkdir -p $OUT
export TMPDIR=/tmp
for v in base nod4; do echo "== $v b3"; DFX_LIB=$PWD/gpurun_build/libdfx_$v.so timeout 120 python tools/diag_nan_batch.py --mode bf16x3 2>&1 | grep -v "^\[W\|amdgpu.ids"; done | tee $OUT/diag.txt
echo "== base b3 blocks=240" | tee -a $OUT/diag.txt; DFX_LIB=$PWD/gpurun_build/libdfx_base.so timeout 120 python tools/diag_nan_batch.py --mode bf16x3 --blocks 240 2>&1 | grep -v "amdgpu.ids" | tee -a $OUT/diag.txt
echo "== base b3 nonan" | tee -a $OUT/diag.txt; DFX_LIB=$PWD/gpurun_build/libdfx_base.so timeout 120 python tools/diag_nan_batch.py --mode bf16x3 --nonan 2>&1 | grep -v "amdgpu.ids" | tee -a $OUT/diag.txt
echo "== base f32" | tee -a $OUT/diag.txt; DFX_LIB=$PWD/gpurun_build/libdfx_base.so timeout 120 python tools/diag_nan_batch.py --mode f32 2>&1 | grep -v "amdgpu.ids" | tee -a $OUT/diag.txt
timeout 600 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.log
