#!/bin/bash
# Round 3, GPU call 4: after the inline-assembly hazard fix (v_cvt_pk_bf16_f32 as a vector conversion): determinism diagnosis, the whole GPU
# suite in both evaluation modes, A/B of the evaluation modes, the bench line with the tail in order and deferred.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03d; mkdir -p $OUT
export TMPDIR=/tmp
B=$PWD/gpurun_build
t0=$(date +%s); lap() { echo "== $1 @ $(( $(date +%s) - t0 )) s"; }
for v in base d4ncb4; do echo "== $v b3"; DFX_LIB=$B/libdfx_$v.so timeout 120 python tools/diag_nan_batch.py --mode bf16x3 2>&1 | grep -v "^\[W\|amdgpu.ids" | cut -c1-300; done | tee $OUT/diag.txt; lap diag
timeout 700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.log; lap suite
DFX_MFMA=f32 timeout 700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_f32chain.log 2>&1; echo "pytest f32 rc=$?"; tail -5 $OUT/pytest_gpu_f32chain.log; lap suite_f32
DFX_LIB=$B/libdfx_d4ncb4.so timeout 300 python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -m gpu -q > $OUT/pytest_d4ncb4.log 2>&1; echo "d4ncb4 tests rc=$?"; tail -3 $OUT/pytest_d4ncb4.log; lap variant_tests
DFX_LIB=$B/libdfx_base.so timeout 150 python tools/ab_mfma_modes.py --clone > $OUT/ab32_base.txt 2>&1; grep ABMODES $OUT/ab32_base.txt | cut -c1-700
DFX_LIB=$B/libdfx_base.so timeout 150 python tools/ab_mfma_modes.py --clone --cs 16 > $OUT/ab16_base.txt 2>&1; grep ABMODES $OUT/ab16_base.txt | cut -c1-700
for v in base d4ncb4; do
  DFX_LIB=$B/libdfx_$v.so timeout 150 python tools/ab_mfma_modes.py --clone --pairs 16 --width 1280 --height 960 --cs 64 > $OUT/ab64_$v.txt 2>&1; grep ABMODES $OUT/ab64_$v.txt | cut -c1-700
done; lap ab
timeout 300 python bench.py --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_deferred.json 2> $OUT/bench_deferred.err; echo "bench deferred rc=$?"; python -c "
import json;d=json.loads(open('$OUT/bench_deferred.json').read().strip().splitlines()[-1]);r=d['roofline'];print('deferred', d['value'], d['ms_per_step'], r['kernel_us'], r['kernel_us_min'], r['kernel_us_max'], r['frac'])"
timeout 300 python bench.py --no-deferred-tail --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_inorder.json 2> $OUT/bench_inorder.err; echo "bench in-order rc=$?"; python -c "
import json;d=json.loads(open('$OUT/bench_inorder.json').read().strip().splitlines()[-1]);r=d['roofline'];print('in-order', d['value'], d['ms_per_step'], r['kernel_us'], r['kernel_us_min'], r['kernel_us_max'], r['frac'])"
lap bench
