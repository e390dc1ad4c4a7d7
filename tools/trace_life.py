import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
import deepfactors_amd as dfx
from deepfactors_amd import synth, _lib
P, W, H, CS = 128, 640, 480, 32
blocks = int(sys.argv[1])
dev = torch.device("cuda", 0)
ctx = dfx.Context(0)
al = dfx.SfmAligner(dfx.SfmAlignerParams(step_blocks=blocks), code_size=CS, ctx=ctx)
keep = [synth.make_pair(W, H, CS, seed=0xDF02 + k, device=dev, motion_scale=0.6 + 0.05 * (k % 8)) for k in range(P)]
arr = al.make_pairs([dict(pose0=t["pose0"], pose1=t["pose1"], cam=t["cam"], img0=t["img0"], img1=t["img1"], dpt0=t["dpt0"], prx0_jac=t["prx_jac"], grad1=t["grad1"], valid0=t["valid0"]) for t in keep])
items = torch.zeros(P * dfx.item_size(12 + CS), dtype=torch.uint8, device=dev)
for _ in range(80):
    al.RunStepBatchAsync(arr, items)
ctx.sync()
ctx.set_profiling(True)
al.RunStepBatchAsync(arr, items)
n_l, ms_l = ctx.profile_read()
nb = P * blocks
ncb = CS // 16
ZD = (1 + ncb * (ncb - 1) // 2 + ncb + 2 * ((ncb + 1) // 2)) * 256
buf = np.zeros(nb * ZD, np.float32)
_lib.check(_lib.lib().dfx_debug_read_partials(ctx.handle, buf.ctypes.data_as(C.c_void_p), buf.nbytes))
zz = buf.reshape(nb, ZD)
hw = zz[:, 176:192].reshape(nb, 4, 4)
z = zz[:, 240:256].reshape(nb, 4, 4)
RT = hw[..., 3].astype(np.float64) / 100.0   # wave loop lifetime in us (100 MHz counter)
A, B, T, N = z[..., 0], z[..., 1], z[..., 3], hw[..., 2]
print(f"blocks/pair {blocks}: kernel {ms_l / n_l * 1e3:.1f} us; wave loop lifetime us: mean {RT.mean():.1f} p5 {np.percentile(RT,5):.1f} p50 {np.percentile(RT,50):.1f} p95 {np.percentile(RT,95):.1f} max {RT.max():.1f}; chunks/wave {N.mean():.1f}")
print(f"   sum of wave lifetimes / (4096 slots) = {RT.sum() / 4096:.1f} us  (= kernel time if every slot were busy all the time)")
