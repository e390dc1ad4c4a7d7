#!/usr/bin/env python3
"""Measures the BASELINE.json configurations that fit one GPU (results table of BASELINE.md section 4).
Writes one JSON object to stdout.  GPU only; CPU-oracle timings are taken on a bounded sample (level 0, few repetitions)."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import deepfactors_amd as dfx
from deepfactors_amd import synth

dev = torch.device("cuda", 0)
ctx = dfx.Context(0)
out = {}


def timed(fn, n, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def pyramid(p, levels):
    """Image pyramid by GaussianBlurDown + Sobel per level (frame.h:80-94); depth by 2x2 subsampling of the synthetic depth."""
    lv = [dict(img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], grad1=p["grad1"])]
    for _ in range(1, levels):
        q = lv[-1]
        h, w = q["img0"].shape
        n = {}
        for k in ("img0", "img1"):
            n[k] = torch.empty((h // 2, w // 2), dtype=torch.float32, device=dev)
            dfx.GaussianBlurDown(q[k], n[k], ctx)
        n["dpt0"] = q["dpt0"][::2, ::2].contiguous()
        n["grad1"] = torch.empty((h // 2, w // 2, 2), dtype=torch.float32, device=dev)
        dfx.SobelGradients(n["img1"], n["grad1"], ctx)
        lv.append(n)
    return lv


# ---- config 1: SE3-only, 640x480, 3-level pyramid, tracker schedule 5,5,10 (common.flags:9), host-driven GN
p = synth.make_pair(640, 480, 16, seed=0xDF01, device=dev, with_decoder=False)
cams = synth.camera_pyramid(p["cam"], 3)
lv = pyramid(p, 3)
se3 = dfx.SE3Aligner(ctx); se3.SetHuberDelta(0.1)
from oracle import dfx_oracle as orc   # solver + CPU baseline only


def track():
    qt = synth.IDENTITY.copy()
    for level, iters in ((2, 5), (1, 5), (0, 10)):
        for _ in range(iters):
            r = se3.RunStep(qt, cams[level], lv[level]["img0"], lv[level]["img1"], lv[level]["dpt0"], lv[level]["grad1"])
            qt = orc.se3_solve_update(r.JtJ, r.Jtr, qt)
    return qt, r


qt, r = track()
gt = p["pose10_true"]
out["cfg1_se3_tracker"] = dict(ms_per_frame=timed(lambda: track(), 5, 1) * 1e3, iterations=20,
                               us_per_step_level0=timed(lambda: se3.RunStep(gt, cams[0], lv[0]["img0"], lv[0]["img1"], lv[0]["dpt0"], lv[0]["grad1"]), 50) * 1e6,
                               final_err=float(r.residual / max(r.inliers, 1)), dt=float(np.linalg.norm(qt[4:] - gt[4:])), dq=float(np.linalg.norm(qt[:4] - gt[:4])))
trk = dfx.CameraTracker(cams, dfx.TrackerConfig(3, (10, 5, 5), 0.1), ctx)
trk.SetKeyframe([l["img0"] for l in lv], [l["dpt0"] for l in lv])


def track_dev():
    trk.Reset()
    return trk.TrackFrame([l["img1"] for l in lv], [l["grad1"] for l in lv])


pd = track_dev()
out["cfg1_se3_tracker"].update(device_resident_ms_per_frame=timed(track_dev, 20, 2) * 1e3, device_dt=float(np.linalg.norm(pd[4:] - gt[4:])),
                               device_dq=float(np.linalg.norm(pd[:4] - gt[:4])))
# Relocalize (deepfactors.cpp:713-743): the live frame against 16 keyframes -- one after the other vs one batched schedule
kf16 = []
for k in range(16):
    q = synth.make_pair(640, 480, 16, seed=0xDF10 + k, device=dev, with_decoder=False)
    lq = pyramid(q, 3)
    kf16.append(([l["img0"] for l in lq], [l["dpt0"] for l in lq]))
kf16[5] = ([l["img0"] for l in lv], [l["dpt0"] for l in lv])
live_img, live_grad = [l["img1"] for l in lv], [l["grad1"] for l in lv]


def reloc_seq():
    best = (float("inf"), -1)
    for k, (ki, kd) in enumerate(kf16):
        trk.SetKeyframe(ki, kd); trk.Reset(); trk.TrackFrame(live_img, live_grad)
        best = min(best, (trk.GetError(), k))
    return best[1]


assert reloc_seq() == 5 and trk.Relocalize(kf16, live_img, live_grad)[0] == 5
out["relocalize_16_keyframes"] = dict(sequential_ms=timed(reloc_seq, 5, 1) * 1e3,
                                      batched_ms=timed(lambda: trk.Relocalize(kf16, live_img, live_grad), 10, 2) * 1e3)
trk.SetKeyframe([l["img0"] for l in lv], [l["dpt0"] for l in lv])
n0 = synth.to_numpy({k: lv[0][k] for k in lv[0]})
t0 = time.perf_counter()
for _ in range(5):
    orc.se3_step(gt, cams[0], n0["img0"], n0["img1"], n0["dpt0"], n0["grad1"], 0.1, accum_f64=False, threads=1)
out["cfg1_se3_tracker"]["cpu_1thread_us_per_step_level0"] = (time.perf_counter() - t0) / 5 * 1e6

# ---- config 2: single 640x480 pair, CS=32: blocking RunStep latency (like the reference's synchronous call)
p2 = synth.make_pair(640, 480, 32, seed=0xDF02, device=dev)
al = dfx.SfmAligner(code_size=32, ctx=ctx)
args = (p2["pose0"], p2["pose1"], p2["code"], p2["cam"], p2["img0"], p2["img1"], p2["dpt0"], None, p2["valid0"], p2["prx_jac"], p2["grad1"])
lat = timed(lambda: al.RunStep(*args), 100)
ctx.set_profiling(True)
for _ in range(50):
    al.RunStep(*args)
n, ms = ctx.profile_read(); ctx.set_profiling(False)
dd = torch.empty_like(p2["img0"])
out["cfg2_single_pair"] = dict(blocking_call_us=lat * 1e6, step_kernel_us=ms / n * 1e3,
                               update_depth_blocking_us=timed(lambda: dfx.UpdateDepth(p2["code"], p2["prx_orig"], p2["prx_jac"], 2.0, dd, ctx), 100) * 1e6,
                               note="one pair = 45 MB: fits the 256 MB Infinity Cache when re-evaluated back to back")

# ---- config 3: 16-keyframe window, 120 pairs (all i<j), one batched launch; pairs share their keyframe's Jacobian
kfs = [synth.make_pair(640, 480, 32, seed=0xDF03 + k, device=dev, motion_scale=0.3) for k in range(16)]
pairs = []
for i in range(16):
    for j in range(i + 1, 16):
        pairs.append(dict(pose0=kfs[i]["pose0"], pose1=kfs[i]["pose1"], cam=kfs[i]["cam"], img0=kfs[i]["img0"], img1=kfs[j]["img1"],
                          dpt0=kfs[i]["dpt0"], prx0_jac=kfs[i]["prx_jac"], grad1=kfs[j]["grad1"]))
arr = al.make_pairs(pairs)
items = torch.zeros(len(pairs) * dfx.item_size(44), dtype=torch.uint8, device=dev)
ctx.set_profiling(True)
t = timed(lambda: al.RunStepBatchAsync(arr, items), 10, 2)
n, ms = ctx.profile_read(); ctx.set_profiling(False)
out["cfg3_window16_120pairs"] = dict(pairs=len(pairs), ms_per_sweep=t * 1e3, pair_evals_per_s=len(pairs) / t, step_kernel_us=ms / n * 1e3,
                                     algorithmic_GBs=148 * 640 * 480 * len(pairs) / (ms / n * 1e-3) / 1e9)
del kfs, pairs, arr

# ---- config 5: 1280x960, CS=64, level 0 (HBM stress): 4 pairs per launch
big = [synth.make_pair(1280, 960, 64, seed=0xDF05 + k, device=dev) for k in range(4)]
al64 = dfx.SfmAligner(code_size=64, ctx=ctx)
arr = al64.make_pairs([dict(pose0=b["pose0"], pose1=b["pose1"], cam=b["cam"], img0=b["img0"], img1=b["img1"], dpt0=b["dpt0"],
                            prx0_jac=b["prx_jac"], grad1=b["grad1"]) for b in big])
items = torch.zeros(4 * dfx.item_size(76), dtype=torch.uint8, device=dev)
ctx.set_profiling(True)
t = timed(lambda: al64.RunStepBatchAsync(arr, items), 10, 2)
n, ms = ctx.profile_read(); ctx.set_profiling(False)
out["cfg5_1280x960_cs64"] = dict(pairs=4, ms_per_sweep=t * 1e3, pair_evals_per_s=4 / t, step_kernel_us=ms / n * 1e3,
                                 algorithmic_GBs=276 * 1280 * 960 * 4 / (ms / n * 1e-3) / 1e9)
print(json.dumps(out))
