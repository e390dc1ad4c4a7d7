#!/bin/bash
# Round 3, GPU call 26: tail kernel with a pair's partials in one batch of loads (8 rows per group) vs two (4); workgroups per CU of the batched SE3 step / EvaluateError.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03z; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_tail_assemble.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest_focus.txt 2>&1
echo "pytest focus exit $?"; tail -2 $OUT/pytest_focus.txt
show() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);r=d['roofline'];print('$2', round(d['value']), 'ms', round(d['ms_per_step']*1e3,1), 'kernel', round(r['kernel_us'],1), 'frac', round(r['frac'],4), 'gap', round(d['ms_per_step']*1e3-r['kernel_us'],1))" 2>&1 | tail -1; }
for r in 1 2 3; do
  for v in rows8 rows4; do
    L=""; [ $v = rows4 ] && L=gpurun_build/libdfx_rows4.so
    DFX_LIB=${L:+$PWD/$L} timeout 200 python bench.py --no-cpu-baseline --no-configs --no-traffic > $OUT/bench_${v}_$r.json 2> $OUT/bench_${v}_$r.err
    show $OUT/bench_${v}_$r.json "$v run $r"
  done
done
for k in 24 8 12 16 32 48; do
  DFX_BATCH_WGS_PER_CU=$k REPS=1 timeout 300 python - <<P
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
import deepfactors_amd as dfx
from deepfactors_amd import synth
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = dfx.Context(0)
W, H, CS, P = 640, 480, 32, 128
prs = [synth.make_pair(W, H, CS, seed=0x2200 + k, device=dev) for k in range(P)]
al, se3 = dfx.SfmAligner(code_size=CS, ctx=ctx), dfx.SE3Aligner(ctx=ctx)
sarr = se3.make_pairs([dict(se3=synth.IDENTITY, cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], grad1=p["grad1"]) for p in prs])
sitems = torch.zeros(P * dfx.item_size(6), dtype=torch.uint8, device=dev)
earr = al.make_pairs([dict(pose0=p["pose0"], pose1=p["pose1"], cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], prx0_jac=p["prx_jac"], grad1=p["grad1"]) for p in prs])
eitems = torch.zeros(P * 16, dtype=torch.uint8, device=dev)
a = bench.event_time_us(torch, lambda: se3.RunStepBatch(sarr, sitems), reps=60, warm=300)
b = bench.event_time_us(torch, lambda: al.EvaluateErrorBatch(earr, eitems), reps=60, warm=300)
print("workgroups per CU $k: se3_step_batch %.1f us  sfm_error_batch %.1f us" % (a, b), flush=True)
P
done 2>&1 | grep "workgroups per CU" | tee $OUT/batch_wgs.txt
