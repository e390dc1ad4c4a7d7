A="--distinct --steps 100 --preroll 3000"
for spec in "32 0,172,40,0,172" "64 0,80,40,0,80" "16 0,240,80,0,240"; do set -- $spec
DFX_SCHEDULE=static timeout 600 python tools/ab_bench.py --worker --pairs $1 $A --blocks $2 2>&1 | grep ABRESULT | python -c "
import sys,json
d=json.loads(sys.stdin.read().split('ABRESULT ')[1]); print('$1 pairs', {k: round(v['kernel_us'],1) for k,v in d.items()})"
done
