#!/bin/bash
# Round-end measurement bundle (every step timeout-guarded, nothing reads stdin):
#   1. bench.py (the driver's default invocation)            -> gpurun_out/final/bench.json
#   2. rocprofv3 --kernel-trace of the same command           -> gpurun_out/final/kt_results.db (+ csv summary)
#   3. rocprofv3 --pmc passes (counters only)                 -> gpurun_out/final/pmc/...
set -u
OUT=gpurun_out/final; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null; echo "bench rc=$?"
timeout 300 rocprofv3 --kernel-trace -d $OUT -o kt -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_profiled.json 2> $OUT/kt.err < /dev/null; echo "kt rc=$?"
timeout 60 python tools/rocpd_summary.py $OUT/kt_results.db --like '%dfx::%' --last 20 > $OUT/kernel_trace_dfx.csv 2>> $OUT/kt.err < /dev/null
timeout 900 tools/profile_pmc.sh $OUT/pmc < /dev/null
timeout 60 python tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt 2>&1 < /dev/null
rm -f $OUT/kt_results.db
cat $OUT/bench.json; cat $OUT/kernel_trace_dfx.csv; cat $OUT/pmc_summary.txt
