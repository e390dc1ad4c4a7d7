#!/usr/bin/env python3
"""Time series of the SfM step kernel over consecutive launches of one process (HIP events inside the library, windows of
`--win` launches): shows the clock ramp after idle and any later power/thermal throttling.  usage: clock_series.py [--n 600]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import deepfactors_amd as dfx
from deepfactors_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=600)
ap.add_argument("--win", type=int, default=10)
ap.add_argument("--pairs", type=int, default=128)
ap.add_argument("--sleep-at", type=int, default=-1, help="window index after which to idle for --sleep seconds")
ap.add_argument("--sleep", type=float, default=0.5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
ctx = dfx.Context(0)
al = dfx.SfmAligner(dfx.SfmAlignerParams(), code_size=32, ctx=ctx)
keep, pairs = [], []
for k in range(a.pairs):
    t = synth.make_pair(640, 480, 32, seed=0xDF02 + k, device=dev, motion_scale=0.6 + 0.05 * (k % 8))
    keep.append(t)
    pairs.append(dict(pose0=t["pose0"], pose1=t["pose1"], cam=t["cam"], img0=t["img0"], img1=t["img1"], dpt0=t["dpt0"], prx0_jac=t["prx_jac"], grad1=t["grad1"], valid0=t["valid0"]))
arr = al.make_pairs(pairs)
items = torch.zeros(a.pairs * dfx.item_size(44), dtype=torch.uint8, device=dev)
ctx.sync()
time.sleep(1.0)
ctx.set_profiling(True)
out = []
t0 = time.perf_counter()
for w in range(a.n // a.win):
    for _ in range(a.win):
        al.RunStepBatchAsync(arr, items)
    n, ms = ctx.profile_read()
    out.append((time.perf_counter() - t0, ms / n * 1e3))
    if w == a.sleep_at:
        time.sleep(a.sleep)
for i, (t, us) in enumerate(out):
    print(f"win {i:3d}  t={t * 1e3:8.1f} ms  kernel {us:8.1f} us")
