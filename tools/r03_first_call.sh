#!/bin/bash
# Runbook for the first GPU call of round 3 (the bf16 split was validated in the last 100 GPU-seconds of round 2; its two follow-up
# variants are unmeasured build flags).  Build the variants first, here in the build container:
#     tools/ab_variants.sh base:"" psplit:"-DDFX_B3_PSPLIT=1" diag4:"-DDFX_B3_DIAG4=1" both:"-DDFX_B3_PSPLIT=1 -DDFX_B3_DIAG4=1" w4:"-DDFX_B3_MIN_WAVES=4" w4psplit:"-DDFX_B3_MIN_WAVES=4 -DDFX_B3_PSPLIT=1"
# then on the GPU box:  bash tools/r03_first_call.sh
#  1. the whole GPU suite with every context created in the bf16 split mode (DFX_MFMA=bf16x3 is read by dfx_ctx_create): what has to be green
#     before the split can become the default for 64-dim codes;
#  2. the split's own tests against each variant library;
#  3. fp32 chain vs split per variant: headline batch (CS 32, 128 pairs) and configs[4] (CS 64, 1280x960, 16 pairs).
mkdir -p gpurun_out/r03
cd "$(dirname "$0")/.."
DFX_MFMA=bf16x3 timeout 300 python -m pytest tests -m gpu -q > gpurun_out/r03/pytest_all_bf16x3.log 2>&1; echo "suite in bf16x3 mode rc=$?"; tail -5 gpurun_out/r03/pytest_all_bf16x3.log
for v in psplit diag4 both w4 w4psplit; do
  [ -f gpurun_build/libdfx_$v.so ] || continue
  DFX_LIB=$PWD/gpurun_build/libdfx_$v.so timeout 120 python -m pytest tests/test_gpu_bf16x3.py -m gpu -q > gpurun_out/r03/pytest_b3_$v.log 2>&1; echo "$v tests rc=$?"; tail -2 gpurun_out/r03/pytest_b3_$v.log
done
for v in base psplit diag4 both w4 w4psplit; do
  [ -f gpurun_build/libdfx_$v.so ] || continue
  DFX_LIB=$PWD/gpurun_build/libdfx_$v.so timeout 90 python tools/ab_mfma_modes.py > gpurun_out/r03/ab32_$v.txt 2>&1; grep ABMODES gpurun_out/r03/ab32_$v.txt || tail -3 gpurun_out/r03/ab32_$v.txt
  DFX_LIB=$PWD/gpurun_build/libdfx_$v.so timeout 90 python tools/ab_mfma_modes.py --pairs 16 --width 1280 --height 960 --cs 64 > gpurun_out/r03/ab64_$v.txt 2>&1; grep ABMODES gpurun_out/r03/ab64_$v.txt || tail -3 gpurun_out/r03/ab64_$v.txt
done
