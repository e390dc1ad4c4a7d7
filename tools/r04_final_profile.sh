#!/bin/bash
# Round-4 measurement bundle (every step timeout-guarded, nothing reads stdin).  Output: gpurun_out/r04final/
#   bench.json                 the driver's default invocation (library defaults; PMC traffic child run, secondary configurations, CPU baseline)
#   bench_under_rocprof.json   the same command (short form) under rocprofv3 --kernel-trace --stats
#   kernel_trace_dfx.csv       per-kernel summary of that trace (all dispatches + the last 30 = the timed steps), kernel_stats.csv = rocprofv3's own --stats table
#   pmc_sq_*                   SQ counters of the default (bf16 split) step kernel at CS = 32 / 128 pairs and CS = 64 / 16 pairs of 1280x960
#   pmc_traffic_*              memory-side request counters by size of the same kernels
#   bench_f32chain.json        the line with the evaluation mode pinned to the fp32 chain
#   bench_rccl_1rank_window.json  one rank under torch.distributed.run: RCCL initialised, the exchange step on real streams, configs[3]
#   small_ops_ktrace.txt, small_ops_events.txt   the batched SE3 step / EvaluateError: kernel-trace durations + inter-dispatch gaps, event-timed calls, tracker
#   latency_cpp.txt, tracker.json, pytest_gpu.log
set -u
export OUT
cd "$(dirname "$0")/.."
OUT=${OUT:-gpurun_out/r04final}; mkdir -p $OUT
export TMPDIR=/tmp
LIB=$PWD/deepfactors_amd/libdfx.so
t0=$(date +%s); lap() { echo "== $1 @ $(( $(date +%s) - t0 )) s"; }
timeout 700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log | head -2; lap suite
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null; echo "bench rc=$?"; lap bench
timeout 300 python bench.py --mfma f32 --no-cpu-baseline --no-configs > $OUT/bench_f32chain.json 2>> $OUT/bench.err < /dev/null; echo "bench f32 rc=$?"; lap bench_f32
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --no-cpu-baseline --no-traffic --no-configs > $OUT/bench_under_rocprof.json 2> $OUT/kt.err < /dev/null; echo "kt rc=$?"
KT=$(find $OUT/kt -name "*kernel_trace.csv" | head -1); ST=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
[ -n "$KT" ] && python tools/kt_summary.py $KT --like dfx --last 30 > $OUT/kernel_trace_dfx.csv
[ -n "$ST" ] && grep -E "^\"Name\"|dfx::" $ST > $OUT/kernel_stats.csv
rm -rf $OUT/kt; lap rocprof
timeout 400 tools/profile_sq.sh $OUT/sq32 $LIB --pairs 128 --distinct --steps 3 --preroll 5 --mode 1 > $OUT/pmc_sq_cs32_128pairs.txt 2>&1 < /dev/null; echo "sq32 rc=$?"
timeout 400 tools/profile_sq.sh $OUT/sq64 $LIB --pairs 16 --width 1280 --height 960 --cs 64 --distinct --steps 3 --preroll 5 --mode 1 > $OUT/pmc_sq_cs64_1280x960_16pairs.txt 2>&1 < /dev/null; echo "sq64 rc=$?"
timeout 400 tools/profile_traffic.sh $OUT/tr32 $LIB --pairs 128 --distinct --steps 3 --preroll 5 --mode 1 > $OUT/pmc_traffic_cs32_128pairs.txt 2>&1 < /dev/null; echo "tr32 rc=$?"; lap pmc
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --window --no-cpu-baseline --no-traffic --no-configs > $OUT/bench_rccl_1rank_window.json 2> $OUT/bench_rccl_1rank_window.err < /dev/null; echo "rccl 1-rank rc=$?"; lap rccl
# the batched SE3 step / EvaluateError: kernel trace with the gaps between dispatches, at the pairs' true poses and at the identity
for pose in true ident; do
  rm -rf /tmp/kts; POSE=$pose REPS=30 timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/kts -o kt -- python tools/small_ops_driver.py > /dev/null 2>&1 < /dev/null
  f=$(find /tmp/kts -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && { echo "== SE3 pose $pose"; python tools/kt_summary.py $f --last 20 | grep -E "se3_step_batch|sfm_error_batch|finalize_rows" | awk -F, '{print $1, "last20 avg us", $8}'; python tools/kt_gaps.py $f --last 6; } >> $OUT/small_ops_ktrace.txt
done
timeout 150 python tools/r04_small_ops.py > $OUT/small_ops_events.txt 2>/dev/null < /dev/null; cat $OUT/small_ops_events.txt; lap small_ops
make -C tests/cpp latency_bench > /dev/null 2>&1
timeout 120 tests/cpp/latency_bench > $OUT/latency_cpp.txt 2>&1 < /dev/null; tail -12 $OUT/latency_cpp.txt
timeout 120 python tools/profile_tracker.py > $OUT/tracker.json 2>/dev/null < /dev/null; tail -2 $OUT/tracker.json
timeout 120 python tools/idle_gap_probe.py --idle-us 0 --seconds 3 > $OUT/clock_power_steady.txt 2>&1 < /dev/null; grep "^idle" $OUT/clock_power_steady.txt
find $OUT -name "*.csv" -size +300k -delete; find $OUT -name "*.db" -delete; lap done
python - <<'PY'
import json
import os
d=json.loads(open(os.environ.get('OUT','gpurun_out/r04final')+'/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print('value',d['value'],'ms',d['ms_per_step'],'kernel',r['kernel_us'],r['kernel_us_min'],r['kernel_us_max'],'frac',r['frac'],'traffic',r['traffic'], r['traffic']/r['algorithmic_bytes_per_launch'] if r['traffic'] else None)
for k,v in d.get('configs',{}).items(): print(k, json.dumps({a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a!='note'})[:330])
print('cpu', json.dumps(d.get('cpu_baseline'))[:300])
PY
cat $OUT/kernel_trace_dfx.csv | cut -c1-260
