#!/bin/bash
# round 4, call 1: the row-walk SE3 step / EvaluateError kernels: parity tests, then timings
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker.py tests/test_gpu_vs_ref.py tests/test_golden_ref_vectors.py tests/test_gpu_cpp_shim.py -m gpu -x -q > gpurun_out/r04_call1_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r04_call1_tests.txt
tail -5 gpurun_out/r04_call1_tests.txt
TAG=new timeout 600 python tools/r04_small_ops.py > gpurun_out/r04_call1_ops.txt 2>&1
cat gpurun_out/r04_call1_ops.txt
