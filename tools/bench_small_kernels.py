#!/usr/bin/env python3
"""HBM-resident timing of the small kernels: 16 distinct 640x480x32 keyframes are swept back to back (750 MB working
set > 256 MB Infinity Cache).  Run under `rocprofv3 --kernel-trace` for per-kernel durations; prints wall figures too."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import deepfactors_amd as dfx
from deepfactors_amd import synth

dev = torch.device("cuda", 0)
ctx = dfx.Context(0)
base = synth.make_pair(640, 480, 32, seed=0xDF02, device=dev)
kfs = [{n: (v.clone() if isinstance(v, torch.Tensor) else v) for n, v in base.items()} for _ in range(16)]
out = [torch.empty_like(k["img0"]) for k in kfs]
se3, sfm = dfx.SE3Aligner(ctx), dfx.SfmAligner(code_size=32, ctx=ctx)
for rep in range(3):
    t0 = time.perf_counter()
    for k, o in zip(kfs, out):
        dfx.UpdateDepth(k["code"], k["prx_orig"], k["prx_jac"], 2.0, o, ctx)
    t1 = time.perf_counter()
    for k in kfs:
        se3.RunStep(k["pose10_true"], k["cam"], k["img0"], k["img1"], k["dpt0"], k["grad1"])
    t2 = time.perf_counter()
    for k in kfs:
        sfm.EvaluateError(k["pose0"], k["pose1"], k["cam"], k["img0"], k["img1"], k["dpt0"], None, k["grad1"])
    t3 = time.perf_counter()
print(f"blocking calls: update_depth {(t1-t0)/16*1e6:.1f} us, se3_step {(t2-t1)/16*1e6:.1f} us, sfm_error {(t3-t2)/16*1e6:.1f} us")
