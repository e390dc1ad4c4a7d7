#!/bin/bash
# Round 3, GPU call 25: tail kernel chosen by the pair's partial volume (per-tile finalize for few pairs with many partials): tests, CS = 64 end to end, small batches.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03y; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tail_assemble.py tests/test_gpu_deferred_tail.py tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_window.py tests/test_gpu_comm.py -x -q -m gpu > $OUT/pytest_focus.txt 2>&1
echo "pytest focus exit $?"; tail -2 $OUT/pytest_focus.txt
show() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);r=d['roofline'];print('$2', round(d['value']), 'ms', round(d['ms_per_step']*1e3,1), 'kernel', round(r['kernel_us'],1), 'frac', round(r['frac'],4), 'gap', round(d['ms_per_step']*1e3-r['kernel_us'],1))" 2>&1 | tail -1; }
for r in 1 2; do
  timeout 300 python bench.py --pairs 16 --width 1280 --height 960 --cs 64 --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_cs64_$r.json 2> $OUT/bench_cs64_$r.err; show $OUT/bench_cs64_$r.json "cs64 16 pairs run $r"
  timeout 300 python bench.py --pairs 16 --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_p16_$r.json 2> $OUT/bench_p16_$r.err; show $OUT/bench_p16_$r.json "cs32 16 pairs run $r"
  timeout 300 python bench.py --pairs 32 --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_p32_$r.json 2> $OUT/bench_p32_$r.err; show $OUT/bench_p32_$r.json "cs32 32 pairs run $r"
  timeout 300 python bench.py --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_p128_$r.json 2> $OUT/bench_p128_$r.err; show $OUT/bench_p128_$r.json "cs32 128 pairs run $r"
done
