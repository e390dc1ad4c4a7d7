#!/bin/bash
# Round 3, GPU call 7: non-temporal partial stores A/B (step -> finalize boundary), the f3 reference pins on the GPU.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03g; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_vs_ref.py tests/test_golden_ref_vectors.py tests/test_gpu_sparse_geometric.py tests/test_gpu_comm.py -m gpu -q > $OUT/pytest_f3.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_f3.log
for r in 1 2; do for v in base pnt; do
  DFX_LIB=$PWD/gpurun_build/libdfx_$v.so timeout 200 python bench.py --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_${v}_$r.json 2> $OUT/bench_${v}_$r.err; python -c "
import json;d=json.loads(open('$OUT/bench_${v}_$r.json').read().strip().splitlines()[-1]);r=d['roofline'];print('$v $r', round(d['value']), round(d['ms_per_step']*1e3,1), round(r['kernel_us'],1), round(r['kernel_us_min'],1), round(r['kernel_us_max'],1), 'gap', round(d['ms_per_step']*1e3-r['kernel_us'],1))"
done; done
