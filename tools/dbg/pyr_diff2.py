import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import deepfactors_amd as dfx
w, h = 256, 8
rng = np.random.default_rng(1)
img = torch.from_numpy(rng.random((h, w), dtype=np.float32)).cuda()
pi = [[img.clone(), torch.full((h // 2, w // 2), float("nan"), device="cuda")]]
pg = [[torch.full((h, w, 2), float("nan"), device="cuda"), torch.full((h // 2, w // 2, 2), float("nan"), device="cuda")]]
dfx.BuildPyramids(pi, pg); torch.cuda.synchronize()
g = torch.empty((h, w, 2), device="cuda"); dfx.SobelGradients(img, g)
print("nan grad", int(torch.isnan(pg[0][0]).sum()), "of", pg[0][0].numel())
print("got row0", pg[0][0][0, :4].cpu().numpy().ravel(), "ref", g[0, :4].cpu().numpy().ravel())
print("got row3", pg[0][0][3, 126:131].cpu().numpy().ravel(), "ref", g[3, 126:131].cpu().numpy().ravel())
b = torch.empty((h // 2, w // 2), device="cuda"); dfx.GaussianBlurDown(img, b)
print("nan img1", int(torch.isnan(pi[0][1]).sum()), "got", pi[0][1][1, :4].cpu().numpy(), "ref", b[1, :4].cpu().numpy())
