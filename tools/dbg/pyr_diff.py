import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import deepfactors_amd as dfx
for (w, h, levels, n) in [(640, 480, 2, 3), (256, 100, 2, 1), (128, 4, 1, 1), (130, 66, 1, 1), (640, 480, 1, 64)]:
    rng = np.random.default_rng(1)
    imgs = [torch.from_numpy(rng.random((h, w), dtype=np.float32)).cuda() for _ in range(n)]
    pi = [[torch.full((h >> i, w >> i), float("nan"), device="cuda") for i in range(levels)] for _ in range(n)]
    pg = [[torch.full((h >> i, w >> i, 2), float("nan"), device="cuda") for i in range(levels)] for _ in range(n)]
    for k in range(n):
        pi[k][0].copy_(imgs[k])
    dfx.BuildPyramids(pi, pg); torch.cuda.synchronize()
    for k in (0, n - 1):
        ref = imgs[k]
        for i in range(levels):
            if i > 0:
                nxt = torch.empty((h >> i, w >> i), device="cuda"); dfx.GaussianBlurDown(ref, nxt); ref = nxt
                d = (pi[k][i] != ref) | torch.isnan(pi[k][i])
                ys, xs = torch.nonzero(d, as_tuple=True)
                print(f"{w}x{h} n={n} frame {k} level {i} img: {int(d.sum())} bad", (int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max())) if d.any() else "")
            g = torch.empty((h >> i, w >> i, 2), device="cuda"); dfx.SobelGradients(ref, g)
            d = ((pg[k][i] != g) | torch.isnan(pg[k][i])).any(-1)
            ys, xs = torch.nonzero(d, as_tuple=True)
            print(f"{w}x{h} n={n} frame {k} level {i} grad: {int(d.sum())} bad", (int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max())) if d.any() else "", "rows", sorted(set(ys.tolist()))[:12] if d.any() else "")
