#!/bin/bash
# Same-box A/B of library variants on the batched row-walk reductions (tools/small_ops_trace.py, warmed HIP-event timing): tools/ab_smallops.sh OUT ROUNDS name=lib.so ...
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=${1:-gpurun_out/ab_smallops}; R=${2:-2}; shift 2; mkdir -p $O
for r in $(seq 1 $R); do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    if [ "$name" = base ]; then unset DFX_LIB; else export DFX_LIB=$PWD/$lib; fi
    echo -n "$name round $r: "
    timeout 300 python tools/small_ops_trace.py 2> /dev/null < /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  '.join(f\"{k} kernel {v['events_kernel_us_last30']:.2f} call {v['call_us_last30']:.2f}\" for k,v in d.items() if k!='_env'))"
  done
done > $O/ab_smallops.txt 2>&1
cat $O/ab_smallops.txt
