#!/bin/bash
# One gpurun call for the pyramid build: tests, event timing (64 frames / 1 frame), rocprofv3 timeline.  usage: tools/pyr_call.sh OUT [suite]
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=${1:-gpurun_out/pyr}; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_pyramid.py -m gpu -q -x > $O/pytest_pyramid.log 2>&1 < /dev/null; tail -3 $O/pytest_pyramid.log
for r in 1 2; do timeout 120 python tools/pyramid_bench.py 64 --build-only 2>&1 < /dev/null | grep build_pyramid; done > $O/pyr_events.txt; cat $O/pyr_events.txt
timeout 120 python tools/pyramid_bench.py 1 2>&1 < /dev/null | grep -v amdgpu.ids > $O/pyr_events_1frame.txt; cat $O/pyr_events_1frame.txt
rm -rf /tmp/pt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o pt -- python tools/pyramid_bench.py 64 --build-only > $O/pyr_under_trace.txt 2>&1 < /dev/null
KT=$(find /tmp/pt -name "*kernel_trace.csv" | head -1)
if [ -n "$KT" ]; then python tools/kt_gaps.py $KT --last 16 > $O/pyr_gaps.txt < /dev/null; python tools/pyramid_trace.py /tmp/pt > $O/pyr_levels.txt < /dev/null; cat $O/pyr_levels.txt $O/pyr_gaps.txt; fi
if [ "${2:-}" = suite ]; then timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1 < /dev/null; tail -3 $O/pytest_gpu.log; fi
exit 0
