import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import deepfactors_amd as dfx
from deepfactors_amd import synth
dev = torch.device("cuda", 0)
ctx = dfx.Context(0)
p = synth.make_pair(640, 480, 16, seed=0xDF01, device=dev, with_decoder=False)
cams = synth.camera_pyramid(p["cam"], 3)
lv = [dict(img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], grad1=p["grad1"])]
for _ in range(2):
    q = lv[-1]; h, w = q["img0"].shape; n = {}
    for k in ("img0", "img1"):
        n[k] = torch.empty((h // 2, w // 2), dtype=torch.float32, device=dev); dfx.GaussianBlurDown(q[k], n[k], ctx)
    n["dpt0"] = q["dpt0"][::2, ::2].contiguous()
    n["grad1"] = torch.empty((h // 2, w // 2, 2), dtype=torch.float32, device=dev); dfx.SobelGradients(n["img1"], n["grad1"], ctx)
    lv.append(n)
trk = dfx.CameraTracker(cams, dfx.TrackerConfig(3, (10, 5, 5), 0.1), ctx)
trk.SetKeyframe([l["img0"] for l in lv], [l["dpt0"] for l in lv])
for _ in range(3):
    trk.Reset(); trk.TrackFrame([l["img1"] for l in lv], [l["grad1"] for l in lv])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    trk.Reset(); trk.TrackFrame([l["img1"] for l in lv], [l["grad1"] for l in lv])
print("ms per frame", (time.perf_counter() - t0) / 20 * 1e3)
