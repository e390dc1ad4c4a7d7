"""Does the step kernel run faster when the GPU idles between launches?  (It does: the sustained clock under back-to-back launches is
power-limited.  rocprofv3 --kernel-trace inserts such gaps, which is why a traced run reports a shorter kernel than an untraced one.)
Prints the HIP-event kernel time of the 128-pair step for back-to-back launches and for launches separated by idle gaps."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import deepfactors_amd as dfx
from deepfactors_amd import synth
P, W, H, CS = 128, 640, 480, 32
dev = torch.device("cuda", 0)
ctx = dfx.Context(0)
al = dfx.SfmAligner(dfx.SfmAlignerParams(), code_size=CS, ctx=ctx)
keep = [synth.make_pair(W, H, CS, seed=0xDF02 + k, device=dev, motion_scale=0.6 + 0.05 * (k % 8)) for k in range(P)]
arr = al.make_pairs([dict(pose0=t["pose0"], pose1=t["pose1"], cam=t["cam"], img0=t["img0"], img1=t["img1"], dpt0=t["dpt0"], prx0_jac=t["prx_jac"], grad1=t["grad1"], valid0=t["valid0"]) for t in keep])
items = torch.zeros(P * dfx.item_size(12 + CS), dtype=torch.uint8, device=dev)


def run(n, gap_s):
    ctx.set_profiling(True)
    for _ in range(n):
        al.RunStepBatchAsync(arr, items)
        if gap_s is not None:
            ctx.sync()
            time.sleep(gap_s)
    nl, ms = ctx.profile_read()
    ctx.set_profiling(False)
    return ms / nl * 1e3


for _ in range(80):
    al.RunStepBatchAsync(arr, items)
ctx.sync()
for label, gap in (("back to back", None), ("sync only", 0.0), ("gap 0.2 ms", 2e-4), ("gap 1 ms", 1e-3), ("gap 5 ms", 5e-3), ("back to back", None), ("back to back x200", None)):
    n = 200 if "x200" in label else 40
    print(f"{label:22s} kernel {run(n, gap):8.1f} us", flush=True)
