#!/bin/bash
# Kernel-trace durations of k_sfm_tail_b3 for library variants (phase-profile builds: -DDFX_TAIL_STOP=k ends the kernel behind phase k): tools/ab_tail_phases.sh OUT name=lib.so ...
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=${1:-gpurun_out/tail_phases}; shift; mkdir -p $O
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  if [ "$name" = base ]; then unset DFX_LIB; else export DFX_LIB=$PWD/$lib; fi
  rm -rf /tmp/kt_$name
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$name -o kt -- python tools/tail_phase_driver.py > /dev/null 2> $O/$name.err < /dev/null
  KT=$(find /tmp/kt_$name -name "*kernel_trace.csv" | head -1)
  echo "== $name"; [ -n "$KT" ] && python tools/kt_summary.py $KT --like dfx --last 30 < /dev/null | cut -c1-60,150-400
done > $O/tail_phases.txt 2>&1
cat $O/tail_phases.txt
