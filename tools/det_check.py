import sys, numpy as np, torch
sys.path.insert(0, '.')
import deepfactors_amd as dfx
from deepfactors_amd import synth, _lib
p = synth.make_pair(320, 240, 32, seed=3, device="cpu"); n = synth.to_numpy(p); g = synth.to_device(p, "cuda")
for mode in (0,):   # the fp32 chain is the only evaluation mode
    for blocks in (300, 300, 0, 37):
        ctx = dfx.Context(0); ctx.set_mfma_mode(mode)
        al = dfx.SfmAligner(dfx.SfmAlignerParams(step_blocks=blocks), code_size=32, ctx=ctx)
        runs = [al.RunStep(n["pose0"], n["pose1"], n["code"], n["cam"], g["img0"], g["img1"], g["dpt0"], None, None, g["prx_jac"], g["grad1"]) for _ in range(12)]
        a = runs[0]
        for k, b in enumerate(runs[1:]):
            if not np.array_equal(a.raw, b.raw):
                d = np.abs(a.JtJ.astype(np.float64) - b.JtJ); i = int(d.argmax())
                print(f"mode {mode} blocks {blocks} run {k+1}: differs; inliers {a.inliers} vs {b.inliers}; max dJtJ {d.max():.3e} at {i} (val {a.JtJ[i]:.6e}); ndiff {int((d>0).sum())}; residual {a.residual} vs {b.residual}")
                break
        else:
            print(f"mode {mode} blocks {blocks}: deterministic")
