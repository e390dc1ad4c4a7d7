import sys, torch
sys.path.insert(0, '/root/repo')
import deepfactors_amd as dfx
from deepfactors_amd import synth
p = synth.make_pair(640, 480, 32, seed=5, device="cuda")
al = dfx.SfmAligner(code_size=32)
for i in range(4):
    print("---- call", i, flush=True)
    al.RunStep(p["pose0"], p["pose1"], None, p["cam"], p["img0"], p["img1"], p["dpt0"], None, p["valid0"], p["prx_jac"], p["grad1"])
    torch.cuda.synchronize()
