#!/usr/bin/env python3
"""Drives the batched step + fused assembly (k_sfm_step + k_sfm_tail_b3<.., ASM>) of bench.py's headline shape for a kernel trace: 128 distinct 640x480 pairs,
chain graph, 150 launches.  Results are NOT checked (phase-profile builds of the tail kernel end early).  usage: tail_phase_driver.py [pairs=128] [launches=150]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    import deepfactors_amd as dfx
    from deepfactors_amd import synth
    from deepfactors_amd.dist import PairGraph, NormalEquations
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    W, H, CS = 640, 480, 32
    dev = torch.device("cuda", 0)
    ctx = dfx.Context(0)
    al = dfx.SfmAligner(dfx.SfmAlignerParams(), code_size=CS, ctx=ctx)
    keep, pairs = [], []
    for k in range(P):
        t = synth.make_pair(W, H, CS, seed=0xDF02 + k, device=dev, motion_scale=0.6 + 0.05 * (k % 8))
        t["valid0"] = ctx.alloc_image(W, H)
        keep.append(t)
        pairs.append(dict(pose0=t["pose0"], pose1=t["pose1"], cam=t["cam"], img0=t["img0"], img1=t["img1"], dpt0=t["dpt0"], prx0_jac=t["prx_jac"], grad1=t["grad1"], valid0=t["valid0"]))
    arr = al.make_pairs(pairs)
    items = torch.zeros(P * dfx.item_size(12 + CS), dtype=torch.uint8, device=dev)
    graph = PairGraph.chain(P)
    sysb = NormalEquations(graph, CS, dev)
    for _ in range(N):
        al.RunStepBatchAssembleAsync(arr, items, sysb, 0, fused=True)
    ctx.sync()
    print("done")


if __name__ == "__main__":
    main()
