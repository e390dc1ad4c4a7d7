A="--width 1280 --height 960 --cs 64 --distinct --steps 40 --preroll 600"
timeout 600 python tools/ab_bench.py --worker --pairs 4 $A --blocks 0,640,480,384,320,256,192,160,128,96 2>&1 | grep ABRESULT | python -c "
import sys,json
d=json.loads(sys.stdin.read().split('ABRESULT ')[1]); print('4 pairs', {k: round(v['kernel_us'],1) for k,v in d.items()})"
timeout 600 python tools/ab_bench.py --worker --pairs 16 $A --blocks 0,480,320,240,160,120,96,64 2>&1 | grep ABRESULT | python -c "
import sys,json
d=json.loads(sys.stdin.read().split('ABRESULT ')[1]); print('16 pairs', {k: round(v['kernel_us'],1) for k,v in d.items()})"
