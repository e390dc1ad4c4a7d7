#!/bin/bash
# A/B of the staging-slot release events with / without a system-scope fence (DFX_STAGE_EVENT_FLAGS): interleaved runs of the pyramid build, the batched
# small operators and bench.py (short form).  usage: tools/ab_stage_events.sh OUT
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=${1:-gpurun_out/ab_stage}; mkdir -p $O
for r in 1 2; do
  for v in nofence fence; do
    if [ $v = fence ]; then export DFX_LIB=$PWD/tools/ab/libdfx_fence.so; else unset DFX_LIB; fi
    echo "== $v round $r"
    timeout 120 python tools/pyramid_bench.py 64 --build-only 2>&1 < /dev/null | grep build_pyramid
    timeout 120 python tools/pyramid_bench.py 1 --build-only 2>&1 < /dev/null | grep build_pyramid
    timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-configs 2> /dev/null < /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('bench ms_per_step', round(d['ms_per_step'],4), 'kernel_us', round(r['kernel_us'],1), 'gap_us', round(d['ms_per_step']*1e3-r['kernel_us'],1), 'frac', round(r['frac'],4))"
    timeout 300 python tools/small_ops_trace.py 2> /dev/null < /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d.items():
    if k!='_env': print(' ',k,'kernel',round(v['events_kernel_us_last30'],2),'call',round(v['call_us_last30'],2))"
  done
done > $O/ab.txt 2>&1
cat $O/ab.txt
