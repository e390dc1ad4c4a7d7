#!/usr/bin/env python3
"""Round 4: the batched decoder (UpdateDepthBatch over 64 distinct 640x480x32 keyframes, bench.py's update_depth_batch_64kf) and a relinearisation
round (dfx_sfm_linearize_batch over 16 keyframes / 120 pairs), event-timed back-to-back calls.  TAG from the environment."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import deepfactors_amd as dfx
from deepfactors_amd import synth
import bench


def main():
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    ctx = dfx.Context(0)
    W, H, CS, K = 640, 480, 32, 64
    kfs = [synth.make_pair(W, H, CS, seed=0x2200 + k, device=dev) for k in range(K)]
    codes = np.stack([np.asarray(k["code"], np.float32) for k in kfs])
    outs = [torch.empty_like(k["img0"]) for k in kfs]
    us = bench.event_time_us(torch, lambda: dfx.UpdateDepthBatch(codes, [k["prx_orig"] for k in kfs], [k["prx_jac"] for k in kfs], 2.0, outs, ctx=ctx), reps=40, warm=150)
    byts = (8 + 4 * CS) * W * H * K
    print(f"{os.environ.get('TAG', '')} update_depth_batch 64 keyframes: {us:.1f} us frac {byts / us / 1e3 / 8000:.3f}", flush=True)


if __name__ == "__main__":
    main()
