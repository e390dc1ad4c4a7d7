#!/usr/bin/env python3
"""Reads the DFX_TRACE stash (per-wave phase-A / phase-B s_memtime sums) after one batched launch."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import deepfactors_amd as dfx
from deepfactors_amd import synth, _lib

P, W, H, CS = 16, 640, 480, 32
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
base = synth.make_pair(W, H, CS, seed=0xDF02, device=dev)
ctx = dfx.Context(0)
ctx.set_mfma_mode(mode)
al = dfx.SfmAligner(dfx.SfmAlignerParams(step_blocks=blocks), code_size=CS, ctx=ctx)
keep = [{n: (v.clone() if isinstance(v, torch.Tensor) else v) for n, v in base.items()} for _ in range(P)]
arr = al.make_pairs([dict(pose0=t["pose0"], pose1=t["pose1"], cam=t["cam"], img0=t["img0"], img1=t["img1"], dpt0=t["dpt0"],
                          prx0_jac=t["prx_jac"], grad1=t["grad1"]) for t in keep])
items = torch.zeros(P * dfx.item_size(12 + CS), dtype=torch.uint8, device=dev)
for _ in range(3):
    al.RunStepBatchAsync(arr, items)
ctx.sync()
nb = P * blocks
buf = np.zeros(nb * 1536, np.float32)
_lib.check(_lib.lib().dfx_debug_read_partials(ctx.handle, buf.ctypes.data_as(C.c_void_p), buf.nbytes))
z = buf.reshape(nb, 1536)[:, 240:256].reshape(nb, 4, 4)
A, B, N, T = z[..., 0], z[..., 1], z[..., 2], z[..., 3]
print(f"mode={mode} blocks/pair={blocks} waves={nb*4} chunks/wave avg={N.mean():.2f}")
print(f"per chunk: phaseA {np.sum(A)/np.sum(N):.0f} cycles, phaseB {np.sum(B)/np.sum(N):.0f} cycles (s_memtime ticks = 100 MHz? see total)")
print(f"wave lifetime avg {T.mean():.0f} ticks, p10 {np.percentile(T,10):.0f} p50 {np.percentile(T,50):.0f} p90 {np.percentile(T,90):.0f} max {T.max():.0f}; A share {np.sum(A)/np.sum(T):.2f}, B share {np.sum(B)/np.sum(T):.2f}")
