#!/bin/bash
# Round 3, GPU call 2: the P-stacked bf16 split as the library default (DFX_MFMA_AUTO), the deferred tail, the batched EvaluateError / SE3
# step, against the whole GPU suite (default modes, then every context pinned to the fp32 chain); A/B of the split's build variants; the
# driver's bench line; the same under rocprofv3 --kernel-trace --stats; RCCL on one rank with the exchange on the tail stream.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03b; mkdir -p $OUT
export TMPDIR=/tmp
B=$PWD/gpurun_build
t0=$(date +%s); lap() { echo "== $1 @ $(( $(date +%s) - t0 )) s"; }
timeout 700 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.log; lap suite
DFX_MFMA=f32 timeout 700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_f32chain.log 2>&1; echo "pytest f32 rc=$?"; tail -6 $OUT/pytest_gpu_f32chain.log; lap suite_f32
for v in d4ncb4 nod4; do
  DFX_LIB=$B/libdfx_$v.so timeout 200 python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_configs.py -m gpu -q > $OUT/pytest_$v.log 2>&1; echo "$v tests rc=$?"; tail -2 $OUT/pytest_$v.log
done; lap variant_tests
for v in base nod4; do
  DFX_LIB=$B/libdfx_$v.so timeout 150 python tools/ab_mfma_modes.py --clone > $OUT/ab32_$v.txt 2>&1; grep ABMODES $OUT/ab32_$v.txt | cut -c1-700 || tail -3 $OUT/ab32_$v.txt
done
DFX_LIB=$B/libdfx_base.so timeout 150 python tools/ab_mfma_modes.py --clone --cs 16 > $OUT/ab16_base.txt 2>&1; grep ABMODES $OUT/ab16_base.txt | cut -c1-700
for v in base d4ncb4; do
  DFX_LIB=$B/libdfx_$v.so timeout 150 python tools/ab_mfma_modes.py --clone --pairs 16 --width 1280 --height 960 --cs 64 > $OUT/ab64_$v.txt 2>&1; grep ABMODES $OUT/ab64_$v.txt | cut -c1-700 || tail -3 $OUT/ab64_$v.txt
done; lap ab
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-2200 $OUT/bench.json; tail -3 $OUT/bench.err; lap bench
timeout 300 python bench.py --no-deferred-tail --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_inorder_tail.json 2> $OUT/bench_inorder_tail.err; echo "bench in-order rc=$?"; cut -c1-700 $OUT/bench_inorder_tail.json; lap bench_inorder
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --no-cpu-baseline --no-traffic --no-configs > $OUT/bench_under_rocprof.json 2> $OUT/kt.err; echo "kt rc=$?"
KT=$(find $OUT/kt -name "*kernel_trace.csv" | head -1); ST=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
[ -n "$KT" ] && python tools/kt_summary.py $KT --like dfx --last 30 > $OUT/kernel_trace_dfx.csv && cat $OUT/kernel_trace_dfx.csv
[ -n "$ST" ] && cp $ST $OUT/kernel_stats.csv && head -8 $OUT/kernel_stats.csv
rm -rf $OUT/kt; lap rocprof
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --window --no-cpu-baseline --no-traffic --no-configs > $OUT/bench_rccl_1rank_window.json 2> $OUT/bench_rccl_1rank_window.err; echo "rccl 1-rank rc=$?"; cut -c1-900 $OUT/bench_rccl_1rank_window.json; grep -v "^\[W" $OUT/bench_rccl_1rank_window.err | tail -3; lap rccl
timeout 400 tools/profile_sq.sh $OUT/sq $PWD/deepfactors_amd/libdfx.so --pairs 128 --distinct --steps 3 --preroll 5 --mode 1 > $OUT/pmc_sq_summary.txt 2>&1 < /dev/null; echo "sq rc=$?"; grep -v "^$" $OUT/pmc_sq_summary.txt | head -40
find $OUT -name "*.csv" -size +300k -delete; lap sq
