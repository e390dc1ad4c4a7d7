#!/bin/bash
# GPU call for the bf16x3 mode: its parity tests, then the one-process A/B against the fp32 chain (CS = 32 at 128 pairs, CS = 64 at 16 pairs of 1280x960)
mkdir -p gpurun_out/b3
cd "$(dirname "$0")/.."
timeout 100 python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/b3/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/b3/pytest.log
timeout 60 python tools/ab_mfma_modes.py > gpurun_out/b3/ab_cs32.txt 2>&1; echo "ab32 rc=$?"; grep ABMODES gpurun_out/b3/ab_cs32.txt || tail -5 gpurun_out/b3/ab_cs32.txt
timeout 60 python tools/ab_mfma_modes.py --pairs 16 --width 1280 --height 960 --cs 64 > gpurun_out/b3/ab_cs64.txt 2>&1; echo "ab64 rc=$?"; grep ABMODES gpurun_out/b3/ab_cs64.txt || tail -5 gpurun_out/b3/ab_cs64.txt
