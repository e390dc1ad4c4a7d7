"""Wave end-time spread of the dynamic schedule (trace build, DFX_TRACE=1): how long slots sit idle after their pair's queue ran dry."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
import deepfactors_amd as dfx
from deepfactors_amd import synth, _lib
P, W, H, CS = 128, 640, 480, 32
dev = torch.device("cuda", 0)
ctx = dfx.Context(0)
al = dfx.SfmAligner(dfx.SfmAlignerParams(), code_size=CS, ctx=ctx)
keep = [synth.make_pair(W, H, CS, seed=0xDF02 + k, device=dev, motion_scale=0.6 + 0.05 * (k % 8)) for k in range(P)]
arr = al.make_pairs([dict(pose0=t["pose0"], pose1=t["pose1"], cam=t["cam"], img0=t["img0"], img1=t["img1"], dpt0=t["dpt0"], prx0_jac=t["prx_jac"], grad1=t["grad1"], valid0=t["valid0"]) for t in keep])
items = torch.zeros(P * dfx.item_size(12 + CS), dtype=torch.uint8, device=dev)
for _ in range(80):
    al.RunStepBatchAsync(arr, items)
ctx.sync()
ctx.set_profiling(True)
al.RunStepBatchAsync(arr, items)
n_l, ms_l = ctx.profile_read()
team = 4096 // P
nb = P * team
ncb = CS // 16
ZD = (1 + ncb * (ncb - 1) // 2 + ncb + 2 * ((ncb + 1) // 2)) * 256
buf = np.zeros(nb * ZD, np.float32)
_lib.check(_lib.lib().dfx_debug_read_partials(ctx.handle, buf.ctypes.data_as(C.c_void_p), buf.nbytes))
zz = buf.reshape(P, team, ZD)
start = zz[:, :, 242].astype(np.float64) / 100.0
life = zz[:, :, 243].astype(np.float64) / 100.0
n = zz[:, :, 178]
t0 = start.min()
end = start - t0 + life
kern = ms_l / n_l * 1e3
print(f"kernel {kern:.1f} us; wave loop start us: max {start.max() - t0:.1f}; wave end us: mean {end.mean():.1f} p5 {np.percentile(end, 5):.1f} p50 {np.percentile(end, 50):.1f} p95 {np.percentile(end, 95):.1f} max {end.max():.1f}")
pe = end.max(axis=1)
print(f"pair end us: min {pe.min():.1f} p25 {np.percentile(pe, 25):.1f} p50 {np.percentile(pe, 50):.1f} p75 {np.percentile(pe, 75):.1f} max {pe.max():.1f};  chunks/wave mean {n.mean():.1f} min {n.min():.0f} max {n.max():.0f}")
print(f"slot-time idle after the wave's exit: {(end.max() - end).sum() / (end.max() * end.size) * 100:.1f} % of slots x kernel loop time")
