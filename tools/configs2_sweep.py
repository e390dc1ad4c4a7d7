#!/usr/bin/env python3
"""BASELINE configs[2] (relinearisation round of a 16-keyframe window: UpdateDepth of the keyframes + one batched RunStep over the 120 / 240 pairs, as bench.py times it)
for several workgroups-per-pair settings of the step kernel.  usage: configs2_sweep.py B [B ...]   (0 = the library's choice)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import numpy as np
    import torch
    import deepfactors_amd as dfx
    from deepfactors_amd import synth
    from deepfactors_amd.dist import PairGraph
    Bs = [int(b) for b in sys.argv[1:]] or [0]
    dev = torch.device("cuda", 0)
    ctx = dfx.Context(0)
    W, H, CS, K = 640, 480, 32, 16
    kfs = [synth.make_pair(W, H, CS, seed=0x1600 + k, device=dev) for k in range(K)]
    for both in (False, True):
        graph = PairGraph.all_pairs(K, both_directions=both)
        plist, prx, codes = [], [], []
        for (i, j) in graph.pairs:
            a, b = kfs[int(i)], kfs[int(j)]
            plist.append(dict(pose0=a["pose0"], pose1=b["pose1"], cam=a["cam"], img0=a["img0"], img1=b["img0"], dpt0=a["dpt0"], valid0=a["valid0"], prx0_jac=a["prx_jac"], grad1=b["grad1"]))
            prx.append(a["prx_orig"])
            codes.append(np.asarray(a["code"].cpu() if hasattr(a["code"], "cpu") else a["code"], np.float32))
        codes = np.stack(codes)
        items = torch.zeros(len(plist) * dfx.item_size(12 + CS), dtype=torch.uint8, device=dev)
        res = {b: [] for b in Bs}
        for rnd in range(3):
            for b in Bs:
                al = dfx.SfmAligner(dfx.SfmAlignerParams(step_blocks=b), code_size=CS, ctx=ctx)
                arr = al.make_pairs(plist)
                for _ in range(100 if rnd else 200):
                    al.LinearizeBatch(arr, prx, codes, items)
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(40):
                    al.LinearizeBatch(arr, prx, codes, items)
                ctx.sync()
                res[b].append((time.perf_counter() - t0) / 40 * 1e6)
        print(f"{len(plist)} pairs: " + "  ".join(f"B={b}: " + "/".join(f"{v:.0f}" for v in res[b]) for b in Bs) + " us per round")


if __name__ == "__main__":
    main()
