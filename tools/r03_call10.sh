#!/bin/bash
# Round 3, GPU call 10: the one-kernel reduction tail (k_sfm_tail_b3 + folded graph assembly): tests, then A/B of the tail variants.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03j; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tail_assemble.py tests/test_gpu_deferred_tail.py tests/test_gpu_valid0_shadow.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_window.py tests/test_gpu_comm.py -x -q -m gpu > $OUT/pytest_focus.txt 2>&1
echo "pytest focus exit $?"; tail -5 $OUT/pytest_focus.txt
show() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);r=d['roofline'];print('$2', round(d['value']), 'ms', round(d['ms_per_step']*1e3,1), 'kernel', round(r['kernel_us'],1), round(r['kernel_us_min'],1), round(r['kernel_us_max'],1), 'frac', round(r['frac'],4), 'gap', round(d['ms_per_step']*1e3-r['kernel_us'],1))" 2>&1 | tail -1; }
for r in 1 2; do
  for v in fused twocall oldtail deferred; do
    F=""; L=""
    [ $v = twocall ] && F="--two-call-tail"
    [ $v = deferred ] && F="--deferred-tail"
    [ $v = oldtail ] && L="gpurun_build/libdfx_oldtail.so" && F="--two-call-tail"
    DFX_LIB=${L:+$PWD/$L} timeout 200 python bench.py $F --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_${v}_$r.json 2> $OUT/bench_${v}_$r.err
    show $OUT/bench_${v}_$r.json "$v $r"
  done
done
# per-kernel times of the default path
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o fused -- python $OLDPWD/bench.py --no-cpu-baseline --no-configs --no-traffic > $OLDPWD/$OUT/bench_under_rocprof.json 2> $OLDPWD/$OUT/rocprof.err; cd $OLDPWD
KT=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); python tools/kt_summary.py $KT --last 30 > $OUT/kernel_trace_summary.csv 2>&1; cut -c1-60,161- $OUT/kernel_trace_summary.csv | head -8
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1
echo "pytest all exit $?"; tail -4 $OUT/pytest_gpu.txt
