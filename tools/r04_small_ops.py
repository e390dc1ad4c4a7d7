#!/usr/bin/env python3
"""Round 4: the pixel reductions on 128 distinct 640x480 pairs per launch (bench.py's small-operator workload) + the device-resident tracker.
Prints one line per measurement; REPS / WARM from the environment."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import deepfactors_amd as dfx
from deepfactors_amd import synth


def ev_us(fn, reps, warm):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    ctx = dfx.Context(0)
    W, H, CS, P = 640, 480, 16, int(os.environ.get("PAIRS", "128"))
    reps, warm = int(os.environ.get("REPS", "60")), int(os.environ.get("WARM", "300"))
    nd = int(os.environ.get("DISTINCT", str(P)))   # distinct image sets (1: everything cache-resident)
    base = [synth.make_pair(W, H, CS, seed=0x2200 + k, device=dev, with_decoder=False) for k in range(nd)]
    prs = [base[k % nd] for k in range(P)]
    if os.environ.get("STAGGER"):   # every image at its own offset inside its allocation (decorrelates the channel / bank pattern of the 640 buffers)
        unit = int(os.environ["STAGGER"])
        def stag(t, k, j):
            n = t.numel(); off = ((k * 5 + j) * 37 % 61) * unit
            big = torch.empty(n + 64 * unit, dtype=t.dtype, device=t.device)
            v = big[off:off + n].view(t.shape); v.copy_(t); return v
        prs = [dict(p, img0=stag(p["img0"], k, 0), img1=stag(p["img1"], k, 1), dpt0=stag(p["dpt0"], k, 2), grad1=stag(p["grad1"], k, 3)) for k, p in enumerate(prs)]
    al, se3 = dfx.SfmAligner(code_size=CS, ctx=ctx), dfx.SE3Aligner(ctx=ctx)
    sitems = torch.zeros(P * dfx.item_size(6), dtype=torch.uint8, device=dev)
    eitems = torch.zeros(P * 16, dtype=torch.uint8, device=dev)
    tag = os.environ.get("TAG", "")
    for name, pose in (("identity", lambda p: synth.IDENTITY), ("true_pose", lambda p: p["pose10_true"])):
        sarr = se3.make_pairs([dict(se3=pose(p), cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], grad1=p["grad1"]) for p in prs])
        us = ev_us(lambda: se3.RunStepBatch(sarr, sitems), reps, warm)
        print(f"{tag} se3_step_batch {P} pairs {name}: {us:.1f} us frac {20 * W * H * P / us / 1e3 / 8000:.3f}", flush=True)
    if True:
        jac = torch.zeros((H, W * CS), dtype=torch.float32, device=dev)
        earr = al.make_pairs([dict(pose0=p["pose0"], pose1=p["pose1"], cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], prx0_jac=jac,
                                   grad1=p["grad1"]) for p in prs])
    us = ev_us(lambda: al.EvaluateErrorBatch(earr, eitems), reps, warm)
    print(f"{tag} sfm_error_batch {P} pairs: {us:.1f} us frac {12 * W * H * P / us / 1e3 / 8000:.3f}", flush=True)
    if os.environ.get("BATCH_ONLY"): return
    # tracker, 3 levels, 20 iterations (tools/profile_tracker.py)
    p = prs[0]
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
    for _ in range(20):
        trk.Reset(); trk.TrackFrame([l["img1"] for l in lv], [l["grad1"] for l in lv])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        trk.Reset(); trk.TrackFrame([l["img1"] for l in lv], [l["grad1"] for l in lv])
    print(f"{tag} tracker ms per frame {(time.perf_counter() - t0) / 100 * 1e3:.4f}", flush=True)
    # blocking single calls
    for nm, fn in (("se3 RunStep", lambda: se3.RunStep(p["pose10_true"], p["cam"], p["img0"], p["img1"], p["dpt0"], p["grad1"])),
                   ("EvaluateError", lambda: al.EvaluateError(p["pose0"], p["pose1"], p["cam"], p["img0"], p["img1"], p["dpt0"], None, p["grad1"]))):
        for _ in range(300): fn()
        t0 = time.perf_counter()
        for _ in range(500): fn()
        print(f"{tag} blocking {nm}: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per call (python)", flush=True)


if __name__ == "__main__":
    main()
