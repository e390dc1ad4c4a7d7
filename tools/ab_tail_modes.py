#!/usr/bin/env python3
"""Interleaved A/B of the batched step's tail modes inside ONE process and box (the boxes differ by +-3 %, consecutive processes by +-1 %): windows of
`--steps` steps (bench.py's workload: 128 distinct 640x480 pairs, CS 32, step + assembly per step) alternate between the in-order tail and the deferred
tail (dfx_set_tail_stream: slim tail kernel beside the next step kernel; zero-copy descriptors unless DFX_STEP_DESC_ZEROCOPY=0).  Per mode: wall clock per
step (synchronised per window), the step kernel's HIP-event time, and their difference."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=128)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=6)
    a = ap.parse_args()
    import numpy as np
    import torch
    import deepfactors_amd as dfx
    from deepfactors_amd import synth
    from deepfactors_amd.dist import NormalEquations, PairGraph
    dev = torch.device("cuda", 0)
    ctx = dfx.Context(0)
    W, H, CS, P = 640, 480, 32, a.pairs
    al = dfx.SfmAligner(code_size=CS, ctx=ctx)
    pairs, keep = [], []
    for k in range(P):
        p = synth.make_pair(W, H, CS, seed=0xDF02 + k, device=dev, motion_scale=0.6 + 0.05 * (k % 8))
        p["valid0"] = ctx.alloc_image(W, H)
        keep.append(p)
        pairs.append(dict(pose0=p["pose0"], pose1=p["pose1"], cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], prx0_jac=p["prx_jac"], grad1=p["grad1"], valid0=p["valid0"]))
    arr = al.make_pairs(pairs)
    items = torch.zeros(P * dfx.item_size(12 + CS), dtype=torch.uint8, device=dev)
    neq = NormalEquations(PairGraph.chain(P), CS, dev)
    tail = torch.cuda.Stream(device=dev)

    def window(n):
        ctx.set_profiling(True)
        ctx.profile_read()
        ctx.tail_join(); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            al.RunStepBatchAssembleAsync(arr, items, neq, 0)
        ctx.tail_join(); ctx.sync()
        dt = (time.perf_counter() - t0) / n
        nl, ms = ctx.profile_read()
        ctx.set_profiling(False)
        return dt * 1e6, ms / nl * 1e3
    for _ in range(8):          # clock ramp
        window(50)
    res = {"in_order": [], "deferred": []}
    for r in range(a.rounds):
        for mode in ("in_order", "deferred"):
            ctx.set_tail_stream(tail if mode == "deferred" else None)
            window(30)
            res[mode].append(window(a.steps))
    ctx.set_tail_stream(None)
    out = {}
    for mode, v in res.items():
        step = float(np.median([x[0] for x in v])); kern = float(np.median([x[1] for x in v]))
        out[mode] = dict(step_us=step, kernel_us=kern, gap_us=step - kern, windows=[[round(x[0], 1), round(x[1], 1)] for x in v])
    out["env"] = {k: os.environ[k] for k in ("DFX_STEP_DESC_ZEROCOPY", "DFX_TAIL_SLIM", "DFX_TAIL_ORDERED") if k in os.environ}
    print(json.dumps(out))
    for mode in ("in_order", "deferred"):
        print(f"# {mode:9s} step {out[mode]['step_us']:8.1f} us  kernel {out[mode]['kernel_us']:8.1f} us  gap {out[mode]['gap_us']:6.1f} us", file=sys.stderr)


if __name__ == "__main__":
    main()
