#!/bin/bash
# Round 3, GPU call 11: the one-kernel reduction tail with device-scope stores instead of fences; rows in flight; idle-gap probe.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03k; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_tail_assemble.py tests/test_gpu_deferred_tail.py tests/test_gpu_valid0_shadow.py -x -q -m gpu > $OUT/pytest_focus.txt 2>&1
echo "pytest focus exit $?"; tail -3 $OUT/pytest_focus.txt
show() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);r=d['roofline'];print('$2', round(d['value']), 'ms', round(d['ms_per_step']*1e3,1), 'kernel', round(r['kernel_us'],1), round(r['kernel_us_min'],1), round(r['kernel_us_max'],1), 'frac', round(r['frac'],4), 'gap', round(d['ms_per_step']*1e3-r['kernel_us'],1))" 2>&1 | tail -1; }
for r in 1 2; do
  for v in fused twocall fused_rows5 fused_rows10 twocall_rows10 oldtail; do
    F=""; L=""
    case $v in twocall*|oldtail) F="--two-call-tail";; esac
    case $v in *rows5) L=gpurun_build/libdfx_rows5.so;; *rows10) L=gpurun_build/libdfx_rows10.so;; oldtail) L=gpurun_build/libdfx_oldtail.so;; esac
    DFX_LIB=${L:+$PWD/$L} timeout 200 python bench.py $F --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_${v}_$r.json 2> $OUT/bench_${v}_$r.err
    show $OUT/bench_${v}_$r.json "$v $r"
  done
done
for v in fused twocall; do
  F=""; [ $v = twocall ] && F="--two-call-tail"
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$v -o kt -- python bench.py $F --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_rocprof_$v.json 2> $OUT/kt_$v.err < /dev/null
  KT=$(find $OUT/kt_$v -name "*kernel_trace.csv" | head -1)
  [ -n "$KT" ] && python tools/kt_summary.py $KT --like dfx --last 30 > $OUT/kernel_trace_$v.csv
  rm -rf $OUT/kt_$v
  echo "== $v"; cut -d, -f1 $OUT/kernel_trace_$v.csv | cut -c1-40 | paste -d, - <(cut -d, -f2- $OUT/kernel_trace_$v.csv | rev | cut -d, -f1-9 | rev) | head -6
done
timeout 300 python tools/idle_gap_probe.py > $OUT/idle_gap_probe.txt 2>&1; cat $OUT/idle_gap_probe.txt | cut -c1-260
