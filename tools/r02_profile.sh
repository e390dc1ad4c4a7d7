#!/bin/bash
# Round-2 measurement bundle (every step timeout-guarded, nothing reads stdin).  Output: gpurun_out/r02final/
#   bench.json            the driver's default invocation (traffic child run, secondary configs, CPU baseline)
#   bench_static.json     same step with the static, bit-reproducible schedule
#   kernel_trace_dfx.csv  rocprofv3 --kernel-trace of `bench.py --steps 20`
#   traffic/ sq/          rocprofv3 --pmc passes (counters only) over the step kernel
#   *.txt                 microbenchmarks behind DESIGN.md section 5, the dynamic schedule's wave end times, call latencies
set -u
OUT=gpurun_out/r02final; mkdir -p $OUT
export TMPDIR=/tmp
LIB=$PWD/deepfactors_amd/libdfx.so
timeout 120 gpurun_build/read_bw > $OUT/ubench_read_bw.txt 2>&1 < /dev/null; echo "read_bw rc=$?"
timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null; echo "bench rc=$?"
timeout 200 python bench.py --schedule static --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_static.json 2>> $OUT/bench.err < /dev/null; echo "bench static rc=$?"
timeout 300 rocprofv3 --kernel-trace -d $OUT -o kt -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-configs > $OUT/bench_profiled.json 2> $OUT/kt.err < /dev/null; echo "kt rc=$?"
timeout 60 python tools/rocpd_summary.py $OUT/kt_results.db --like '%dfx::%' --last 20 > $OUT/kernel_trace_dfx.csv 2>> $OUT/kt.err < /dev/null
rm -f $OUT/kt_results.db
timeout 600 tools/profile_traffic.sh $OUT/traffic $LIB > $OUT/pmc_traffic_summary.txt 2>&1 < /dev/null; echo "traffic rc=$?"
timeout 600 tools/profile_sq.sh $OUT/sq $LIB > $OUT/pmc_sq_summary.txt 2>&1 < /dev/null; echo "sq rc=$?"
timeout 120 gpurun_build/issue_cost > $OUT/ubench_issue_cost.txt 2>&1 < /dev/null
timeout 60 gpurun_build/mfma4x4_bcast > $OUT/ubench_mfma4x4_bcast.txt 2>&1 < /dev/null
timeout 60 gpurun_build/scalar_atomic > $OUT/ubench_scalar_atomic.txt 2>&1 < /dev/null
DFX_LIB=$PWD/gpurun_build/libdfx_trace.so timeout 200 python tools/trace_dyn.py > $OUT/dyn_wave_end_times.txt 2>&1 < /dev/null
timeout 120 tests/cpp/latency_bench > $OUT/latency_cpp.txt 2>&1 < /dev/null
timeout 200 python tools/clock_series.py --n 2000 --win 50 2>&1 < /dev/null | grep -v amdgpu.ids > $OUT/clock_series.txt
timeout 300 python tools/run_configs.py > $OUT/configs.json 2> $OUT/configs.err < /dev/null
timeout 120 gpurun_build/read_bw >> $OUT/ubench_read_bw.txt 2>&1 < /dev/null
find $OUT -name "*.csv" -size +200k -delete
find $OUT -name "*.db" -delete
cat $OUT/bench.json; cat $OUT/bench_static.json; cat $OUT/kernel_trace_dfx.csv; cat $OUT/pmc_traffic_summary.txt $OUT/pmc_sq_summary.txt | grep -v "^$" | head -80; cat $OUT/ubench_read_bw.txt $OUT/dyn_wave_end_times.txt $OUT/latency_cpp.txt
