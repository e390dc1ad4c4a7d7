#!/bin/bash
# tools/ab_lib_bench.sh with extra bench.py arguments: tools/ab_lib_bench_args.sh OUT ROUNDS "ARGS" name=lib.so ...
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$1; R=$2; ARGS=$3; shift 3; mkdir -p $O
for r in $(seq 1 $R); do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    if [ "$name" = base ]; then unset DFX_LIB; else export DFX_LIB=$PWD/$lib; fi
    echo -n "$name round $r: "
    timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-configs --steps 30 --warmup 15 $ARGS 2> /dev/null < /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step', round(d['ms_per_step'],4), 'kernel_us', round(r['kernel_us'],1), 'frac', round(r['frac'],4), 'value', round(d['value']), r['kernel'])"
  done
done
