#!/bin/bash
# Same-box A/B of pyramid-build variants: tools/ab_pyr.sh OUT name=lib.so ... (name "base" = the tree's libdfx.so); three interleaved rounds + a kernel trace each
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=${1:-gpurun_out/ab_pyr}; shift; mkdir -p $O
for r in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    if [ "$name" = base ]; then unset DFX_LIB; else export DFX_LIB=$PWD/$lib; fi
    echo -n "$name round $r: "; timeout 120 python tools/pyramid_bench.py ${PYR_FRAMES:-64} --build-only 2>&1 < /dev/null | grep build_pyramid | cut -c1-110
  done
done > $O/ab.txt 2>&1
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  if [ "$name" = base ]; then unset DFX_LIB; else export DFX_LIB=$PWD/$lib; fi
  rm -rf /tmp/pt; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o pt -- python tools/pyramid_bench.py ${PYR_FRAMES:-64} --build-only > /dev/null 2>&1 < /dev/null
  echo "== trace $name"; python tools/pyramid_trace.py /tmp/pt < /dev/null
done >> $O/ab.txt 2>&1
cat $O/ab.txt
