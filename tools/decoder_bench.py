#!/usr/bin/env python3
"""UpdateDepth of K distinct 640x480 keyframes in one launch (k_update_depth_batch<32>): event time and fraction of 8 TB/s (136 B/px).  usage: decoder_bench.py [K=64]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import deepfactors_amd as dfx
from deepfactors_amd import synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H, CS = 640, 480, 32
ctx = dfx.Context(0)
kfs = [synth.make_pair(W, H, CS, seed=0x2200 + k, device="cuda") for k in range(K)]
codes = np.stack([np.asarray(k["code"], np.float32) for k in kfs])
outs = [torch.empty_like(k["img0"]) for k in kfs]
fn = lambda: dfx.UpdateDepthBatch(codes, [k["prx_orig"] for k in kfs], [k["prx_jac"] for k in kfs], 2.0, outs, ctx=ctx)
for _ in range(100): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40): fn()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 40
b = (8 + 4 * CS) * W * H * K
import hashlib
print(f"update_depth_batch {K} kf: {us:.1f} us = {b / us / 1e3:.0f} GB/s = {b / us / 1e3 / 8000:.3f} of 8 TB/s; digest {hashlib.sha256(outs[0].cpu().numpy().tobytes() + outs[-1].cpu().numpy().tobytes()).hexdigest()[:16]}")
