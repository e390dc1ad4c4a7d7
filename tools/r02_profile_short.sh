#!/bin/bash
# Trimmed bundle for the default (static) schedule: driver-like bench line (with its own PMC traffic child run), kernel trace of the
# same command, SQ counters.  Output: gpurun_out/r02s/
set -u
OUT=gpurun_out/r02s; mkdir -p $OUT
export TMPDIR=/tmp
LIB=$PWD/deepfactors_amd/libdfx.so
timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null; echo "bench rc=$?"
timeout 300 rocprofv3 --kernel-trace -d $OUT -o kt -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-configs > $OUT/bench_profiled.json 2> $OUT/kt.err < /dev/null; echo "kt rc=$?"
timeout 60 python tools/rocpd_summary.py $OUT/kt_results.db --like '%dfx::%' --last 20 > $OUT/kernel_trace_dfx.csv 2>> $OUT/kt.err < /dev/null
rm -f $OUT/kt_results.db
timeout 400 tools/profile_sq.sh $OUT/sq $LIB > $OUT/pmc_sq_summary.txt 2>&1 < /dev/null; echo "sq rc=$?"
find $OUT -name "*.csv" -size +200k -delete; find $OUT -name "*.db" -delete
