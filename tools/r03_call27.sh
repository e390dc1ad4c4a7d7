#!/bin/bash
# Round 3, GPU call 27: the memory floor of the step kernel's own access pattern: phase B reduced to the operand stream (one xor per vector), with and without phase A.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03aa; mkdir -p $OUT
export TMPDIR=/tmp
for v in full streamA stream; do
  L=""; [ $v != full ] && L=gpurun_build/libdfx_$v.so
  echo "== $v"
  DFX_LIB=${L:+$PWD/$L} timeout 120 python tools/idle_gap_probe.py --idle-us 0 --seconds 2.5 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    l=l.rstrip()
    if l.startswith('idle'): print(l)
    elif l.strip().startswith('{'):
        d=json.loads(l.strip()); print('   ', {k.split(' (')[0].replace(' clock speed:',''):v for k,v in d.items() if 'sclk clock speed' in k or 'Power' in k})
" | tee $OUT/stream_$v.txt | head -4
done
