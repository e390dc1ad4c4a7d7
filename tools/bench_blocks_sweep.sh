#!/bin/bash
# bench.py over a list of workgroups-per-pair values (distinct pairs, warm clocks); usage: tools/bench_blocks_sweep.sh 80 64 96 ...
for b in "$@"; do
  timeout 300 python bench.py --no-cpu-baseline --step-blocks $b < /dev/null 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('blocks', $b, round(d['value']), round(d['roofline']['kernel_us'], 1), round(d['roofline']['frac'], 3))"
done
