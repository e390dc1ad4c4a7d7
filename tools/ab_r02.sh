L=gpurun_build
A="--pairs 128 --distinct --steps 30 --rounds 3 --preroll 450"
timeout 1200 python tools/ab_bench.py --libs $L/libdfx_cur.so,$L/libdfx_vbuf.so,$L/libdfx_wgp.so,$L/libdfx_both.so $A 2>&1 | grep round | sed 's/"inliers.*//' > gpurun_out/ab_aux.txt
cat gpurun_out/ab_aux.txt
