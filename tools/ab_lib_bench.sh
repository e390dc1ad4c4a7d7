#!/bin/bash
# Same-box A/B of library variants on bench.py (short form): tools/ab_lib_bench.sh OUT ROUNDS name=lib.so ... ("base" = the tree's libdfx.so)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=${1:-gpurun_out/ab_lib}; R=${2:-2}; shift 2; mkdir -p $O
for r in $(seq 1 $R); do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    if [ "$name" = base ]; then unset DFX_LIB; else export DFX_LIB=$PWD/$lib; fi
    echo -n "$name round $r: "
    timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-configs --steps 40 --warmup 20 2> /dev/null < /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step', round(d['ms_per_step'],4), 'kernel_us', round(r['kernel_us'],1), 'gap_us', round(d['ms_per_step']*1e3-r['kernel_us'],1), 'frac', round(r['frac'],4), 'value', round(d['value']))"
  done
done > $O/ab_bench.txt 2>&1
cat $O/ab_bench.txt
