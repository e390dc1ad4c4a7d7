#!/bin/bash
# Round-3 measurement bundle for the bf16 split (DFX_MFMA_BF16X3), to be run once it has passed the whole GPU suite
# (tools/r03_first_call.sh): the driver-like bench line in that mode (with its own PMC traffic child run), the kernel trace of the
# same command, SQ counters of the split kernel (tools/ab_bench.py --mode 1 = DFX_MFMA_BF16X3).  Output: gpurun_out/r03b3/
set -u
OUT=gpurun_out/r03b3; mkdir -p $OUT
export TMPDIR=/tmp
LIB=$PWD/deepfactors_amd/libdfx.so
timeout 500 python bench.py --mfma bf16x3 > $OUT/bench_bf16x3.json 2> $OUT/bench.err < /dev/null; echo "bench rc=$?"
timeout 300 rocprofv3 --kernel-trace -d $OUT -o kt -- python bench.py --mfma bf16x3 --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-configs > $OUT/bench_profiled.json 2> $OUT/kt.err < /dev/null; echo "kt rc=$?"
timeout 60 python tools/rocpd_summary.py $OUT/kt_results.db --like '%dfx::%' --last 20 > $OUT/kernel_trace_dfx.csv 2>> $OUT/kt.err < /dev/null
rm -f $OUT/kt_results.db
timeout 400 tools/profile_sq.sh $OUT/sq $LIB --pairs 128 --distinct --steps 3 --preroll 5 --mode 1 > $OUT/pmc_sq_summary.txt 2>&1 < /dev/null; echo "sq rc=$?"
timeout 400 tools/profile_sq.sh $OUT/sq64 $LIB --pairs 16 --width 1280 --height 960 --cs 64 --distinct --steps 3 --preroll 5 --mode 1 > $OUT/pmc_sq64_summary.txt 2>&1 < /dev/null; echo "sq64 rc=$?"
find $OUT -name "*.csv" -size +200k -delete; find $OUT -name "*.db" -delete
cat $OUT/bench_bf16x3.json; cat $OUT/kernel_trace_dfx.csv; grep -v "^$" $OUT/pmc_sq_summary.txt $OUT/pmc_sq64_summary.txt | head -60
