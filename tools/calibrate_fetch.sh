#!/bin/bash
# Calibrates rocprofv3 FETCH_SIZE on THIS kernel's access pattern (8-byte-per-lane buffer loads): the ring-only ablation
# build (-DDFX_ABLATE=3) reads a known number of bytes (Jacobian stream + depth + intensity, no gathers).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/calib; mkdir -p $OUT
DFX_LIB=$PWD/gpurun_build/libdfx_onlyring.so timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_sfm_step" --output-format csv -d $OUT/fetch -o pmc -- \
  python tools/ab_bench.py --worker --blocks 96 --pairs 16 --steps 4 --mode 0 > $OUT/worker.log 2> $OUT/err.log < /dev/null
echo "rc=$?"
timeout 60 python tools/pmc_summary.py $OUT < /dev/null
