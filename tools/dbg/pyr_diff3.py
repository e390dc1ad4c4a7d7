import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import deepfactors_amd as dfx
w, h = 256, 100
rng = np.random.default_rng(1)
img = torch.from_numpy(rng.random((h, w), dtype=np.float32)).cuda()
pi = [[img.clone(), torch.full((h // 2, w // 2), float("nan"), device="cuda")]]
pg = [[torch.full((h, w, 2), float("nan"), device="cuda"), None]]
dfx.BuildPyramids(pi, pg); torch.cuda.synchronize()
b = torch.empty((h // 2, w // 2), device="cuda"); dfx.GaussianBlurDown(img, b)
got, ref = pi[0][1].cpu().numpy(), b.cpu().numpy()
bad = np.argwhere(got != ref)
print(len(bad), "bad; nan", int(np.isnan(got).sum()))
for (y, x) in bad[:25]:
    print(y, x, got[y, x], ref[y, x], "ulps", (got[y, x].view(np.int32) - ref[y, x].view(np.int32)))
import collections
print("by row parity", collections.Counter((bad[:, 0] % 2).tolist()), "by x%64", sorted(collections.Counter((bad[:, 1] % 64).tolist()).items())[:10])
