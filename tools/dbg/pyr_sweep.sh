#!/bin/bash
# usage: pyr_sweep.sh "4096:1 8192:1 ..."   (waves:NP)
cd /tmp; export TMPDIR=/tmp
for cfg in $1; do
  w=${cfg%%:*}; np=${cfg##*:}
  rm -rf /tmp/pt
  DFX_TUNE_PYR_WAVES=$w DFX_TUNE_PYR_NP=$np timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o pt -- python $GRAFT_REPO_ROOT/tools/pyramid_bench.py ${2:-64} > /tmp/pt.out 2>/tmp/pt.err < /dev/null
  echo "== waves=$w NP=$np: $(head -1 /tmp/pt.out | cut -c1-90)"
  python $GRAFT_REPO_ROOT/tools/pyramid_trace.py /tmp/pt | cut -c1-130
done
