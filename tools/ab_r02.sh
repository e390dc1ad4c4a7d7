L=gpurun_build
A="--pairs 128 --distinct --steps 30 --rounds 3 --preroll 450"
timeout 1200 python tools/ab_bench.py --libs $L/libdfx_prev.so,$L/libdfx_new.so $A 2>&1 | grep round | sed 's/"inliers.*//' > gpurun_out/ab_aux.txt
cat gpurun_out/ab_aux.txt
timeout 300 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2
