#!/usr/bin/env python3
"""Reads the DFX_TRACE stash (per-wave phase-A / phase-B s_memtime sums) after one batched launch."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import deepfactors_amd as dfx
from deepfactors_amd import synth, _lib

P, W, H, CS = int(os.environ.get("DFX_TRACE_PAIRS", "16")), 640, 480, 32
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
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
ctx.set_profiling(True)
al.RunStepBatchAsync(arr, items)
n_l, ms_l = ctx.profile_read()
ctx.set_profiling(False)
print(f"step kernel {ms_l / n_l * 1e3:.1f} us (HIP events)")
nb = P * blocks
ncb = CS // 16
ZD = (1 + ncb * (ncb - 1) // 2 + ncb + 2 * ((ncb + 1) // 2)) * 256   # z-space partial: block 0 (P x P sums + trace slots) + packed MFMA blocks
buf = np.zeros(nb * ZD, np.float32)
_lib.check(_lib.lib().dfx_debug_read_partials(ctx.handle, buf.ctypes.data_as(C.c_void_p), buf.nbytes))
zz = buf.reshape(nb, ZD)
z = zz[:, 240:256].reshape(nb, 4, 4)
hw = zz[:, 176:192].reshape(nb, 4, 4)
A, B, S, T = z[..., 0].astype(np.float64), z[..., 1].astype(np.float64), z[..., 2].astype(np.float64), z[..., 3].astype(np.float64)
N = hw[..., 2].astype(np.float64)
hwid, xcc = hw[..., 0].astype(np.int64), hw[..., 1].astype(np.int64)
RT = hw[..., 3].astype(np.float64)   # wave lifetime in 100 MHz ticks
print(f"core clock while the waves ran: {np.sum(T) / np.sum(RT) * 100:.0f} MHz (s_memtime / s_memrealtime); mean wave lifetime {RT.mean() / 100:.1f} us")
S0 = S.copy()
for x in np.unique(xcc):   # the XCDs' counters need not be synchronized: unwrap the 24-bit window and align every XCD on its own first wave
    m = xcc == x
    v = S0[m]
    ref = np.median(v)
    v = np.where(v < ref - 2 ** 23, v + 2 ** 24, np.where(v > ref + 2 ** 23, v - 2 ** 24, v))
    S0[m] = v - v.min()
E = S0 + T
span = E.max()
print(f"mode={mode} blocks/pair={blocks} waves={nb*4} chunks/wave avg={N.mean():.2f}; kernel span {span:.0f} ticks = {span / (ms_l / n_l * 1e3):.0f} ticks/us")
print(f"per chunk per wave: phaseA {np.sum(A)/np.sum(N):.0f} ticks, phaseB {np.sum(B)/np.sum(N):.0f} ticks; prologue+epilogue per wave {np.mean(T-A-B):.0f} ticks")
print(f"wave lifetime avg {T.mean():.0f} ticks, p10 {np.percentile(T,10):.0f} p50 {np.percentile(T,50):.0f} p90 {np.percentile(T,90):.0f} max {T.max():.0f}; A share {np.sum(A)/np.sum(T):.2f}, B share {np.sum(B)/np.sum(T):.2f}")
print(f"wave-ticks total {T.sum():.3e} = {T.sum()/span:.0f} waves resident on average (capacity 1024 SIMDs x occupancy)")
# start-time histogram (launch ramp) and end-time histogram (tail), in 10 bins of the span
hs, _ = np.histogram(S0, bins=10, range=(0, span)); he, _ = np.histogram(E, bins=10, range=(0, span))
print("starts per decile of the span:", hs.tolist()); print("ends   per decile of the span:", he.tolist())
# resident waves over time
grid = np.linspace(0, span, 41)
res = [(int(((S0 <= g) & (E > g)).sum())) for g in grid]
print("resident waves at 41 time points:", res)
simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 15; se = (hwid >> 13) & 7
key = ((xcc * 8 + se) * 16 + cu) * 4 + simd
u, cnt = np.unique(key, return_counts=True)
print(f"distinct (xcc,se,cu,simd) slots used: {len(u)}; waves per SIMD min {cnt.min()} max {cnt.max()}; xcc values {np.unique(xcc).tolist()}")
busy = np.zeros(len(u)); idx = {k: i for i, k in enumerate(u)}
for k, a_, b_ in zip(key.ravel(), A.ravel(), B.ravel()): busy[idx[k]] += a_ + b_
print(f"per-SIMD sum of (A+B) wave-ticks / span: mean {busy.mean()/span:.2f} min {busy.min()/span:.2f} max {busy.max()/span:.2f}  (= average number of waves in a timed phase)")
