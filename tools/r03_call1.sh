#!/bin/bash
# Round 3, GPU call 1: the valid0 shadow + DFX_MFMA_AUTO build against the whole GPU suite (default modes, then every context in the bf16
# split mode), the build-flag variants of tools/ab_variants.sh (dead-chunk skip, bf16 split follow-ups, XCD-local dynamic teams) A/B'd in
# one box, the bench line with PMC traffic for both schedules, and RCCL on one rank (torchrun --nproc-per-node=1 bench.py --window).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03a; mkdir -p $OUT
export TMPDIR=/tmp
B=$PWD/gpurun_build
t0=$(date +%s); lap() { echo "== $1 @ $(( $(date +%s) - t0 )) s"; }
timeout 700 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log; lap suite
DFX_MFMA=bf16x3 timeout 700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_bf16x3.log 2>&1; echo "pytest bf16x3 rc=$?"; tail -6 $OUT/pytest_gpu_bf16x3.log; lap suite_b3
DFX_LIB=$B/libdfx_skip.so timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_valid0_shadow.py -m gpu -q > $OUT/pytest_skip.log 2>&1; echo "skip tests rc=$?"; tail -2 $OUT/pytest_skip.log
for v in both bothw4; do
  DFX_LIB=$B/libdfx_$v.so timeout 200 python -m pytest tests/test_gpu_bf16x3.py -m gpu -q > $OUT/pytest_b3_$v.log 2>&1; echo "$v tests rc=$?"; tail -2 $OUT/pytest_b3_$v.log
done; lap variant_tests
for v in base skip skipw4 psplit diag4 both bothw4; do
  DFX_LIB=$B/libdfx_$v.so timeout 150 python tools/ab_mfma_modes.py --clone > $OUT/ab32_$v.txt 2>&1; grep ABMODES $OUT/ab32_$v.txt | cut -c1-600 || tail -3 $OUT/ab32_$v.txt
done; lap ab32
DFX_LIB=$B/libdfx_base.so timeout 150 python tools/ab_mfma_modes.py --clone --foreign-valid0 > $OUT/ab32_base_foreign.txt 2>&1; grep ABMODES $OUT/ab32_base_foreign.txt | cut -c1-600
for v in base psplit; do
  DFX_LIB=$B/libdfx_$v.so timeout 150 python tools/ab_mfma_modes.py --clone --pairs 16 --width 1280 --height 960 --cs 64 > $OUT/ab64_$v.txt 2>&1; grep ABMODES $OUT/ab64_$v.txt | cut -c1-600 || tail -3 $OUT/ab64_$v.txt
done; lap ab64
timeout 400 python bench.py --schedule static --no-cpu-baseline > $OUT/bench_static.json 2> $OUT/bench_static.err; echo "bench static rc=$?"; cut -c1-1500 $OUT/bench_static.json; lap bench_static
timeout 300 python bench.py --schedule dynamic --no-cpu-baseline --no-configs > $OUT/bench_dynamic.json 2> $OUT/bench_dynamic.err; echo "bench dynamic rc=$?"; cut -c1-1500 $OUT/bench_dynamic.json; lap bench_dynamic
DFX_LIB=$B/libdfx_rot0.so timeout 300 python bench.py --schedule dynamic --no-cpu-baseline --no-configs > $OUT/bench_dynamic_rot0.json 2> $OUT/bench_dynamic_rot0.err; echo "bench rot0 rc=$?"; cut -c1-1500 $OUT/bench_dynamic_rot0.json; lap bench_rot0
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --window --schedule static --no-cpu-baseline --no-traffic --no-configs > $OUT/bench_rccl_1rank_window.json 2> $OUT/bench_rccl_1rank_window.err; echo "rccl 1-rank rc=$?"; cut -c1-2500 $OUT/bench_rccl_1rank_window.json; tail -3 $OUT/bench_rccl_1rank_window.err; lap rccl
