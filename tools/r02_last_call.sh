#!/bin/bash
# The round's last GPU call (4 GPU-minutes left): bf16 split probes for the next round's kernel decision, then the GPU test suite at HEAD
# (new reference-vector test first) and a short bench line.  Everything lands under gpurun_out/last/.
mkdir -p gpurun_out/last
cd "$(dirname "$0")/.."
timeout 30 gpurun_build/bf16x3_probe > gpurun_out/last/bf16x3_probe.txt 2>&1; echo "probe rc=$?"
timeout 60 gpurun_build/issue_cost_bf16 > gpurun_out/last/issue_cost_bf16.txt 2>&1; echo "issue_cost rc=$?"
timeout 170 python -m pytest tests/test_golden_ref_vectors.py tests -m gpu -q > gpurun_out/last/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/last/pytest_gpu.log
timeout 120 python bench.py --no-configs --no-cpu-baseline --no-traffic > gpurun_out/last/bench_short.json 2> gpurun_out/last/bench_short.err; echo "bench rc=$?"
cat gpurun_out/last/bf16x3_probe.txt
