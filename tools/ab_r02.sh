A="--distinct --steps 40 --preroll 600"
for i in 1 2 3; do
DFX_SCHEDULE=static timeout 600 python tools/ab_bench.py --worker --pairs 128 $A --blocks 0 2>&1 | grep ABRESULT | sed 's/"inliers.*//;s/^/static /'
timeout 600 python tools/ab_bench.py --worker --pairs 128 $A --blocks 0 2>&1 | grep ABRESULT | sed 's/"inliers.*//;s/^/dynamic /'
done
