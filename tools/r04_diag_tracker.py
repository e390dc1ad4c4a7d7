#!/usr/bin/env python3
"""Round 4 diagnosis: bench.py's configs0 tracker entry measured 0.99 ms per frame in the final bundle and 0.26 ms in tools/profile_tracker.py
of the same call.  Which earlier stage of bench.py's process makes the difference?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import deepfactors_amd as dfx
from deepfactors_amd import synth
import bench


def tracker_ms(ctx, dev, tag):
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
    for _ in range(20):
        trk.Reset(); trk.TrackFrame([l["img1"] for l in lv], [l["grad1"] for l in lv])
    t0 = time.perf_counter()
    for _ in range(100):
        trk.Reset(); trk.TrackFrame([l["img1"] for l in lv], [l["grad1"] for l in lv])
    print(f"{tag}: {(time.perf_counter() - t0) / 100 * 1e3:.4f} ms per frame", flush=True)


def main():
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    ctx = dfx.Context(0)
    tracker_ms(ctx, dev, "fresh context")
    if os.environ.get("HEADLINE"):
        from deepfactors_amd.dist import NormalEquations, PairGraph
        from deepfactors_amd import _lib as _dl
        W, H, CS, P = 640, 480, 32, 128
        ctx.set_mfma_mode(_dl.DFX_MFMA_AUTO); ctx.set_schedule(_dl.DFX_SCHEDULE_AUTO)
        al = dfx.SfmAligner(dfx.SfmAlignerParams(step_blocks=0), code_size=CS, ctx=ctx)
        pairs, keep = bench.build_pairs(dfx, synth, dev, 0, P, W, H, CS, ctx=ctx if os.environ.get("HEADLINE") != "foreign" else None)
        arr = al.make_pairs(pairs)
        items = torch.zeros(P * dfx.item_size(12 + CS), dtype=torch.uint8, device=dev)
        if os.environ.get("HEADLINE") == "noassemble":
            for _ in range(60): al.RunStepBatchAsync(arr, items)
        else:
            neq = NormalEquations(PairGraph.chain(P), CS, dev)
            ctx.set_profiling(True)
            for _ in range(60): al.RunStepBatchAssembleAsync(arr, items, neq, 0, fused=True)
            ctx.sync(); print("headline kernel us", [round(x, 1) for x in (lambda r: (r[1] / max(r[0], 1) * 1e3,))(ctx.profile_read())])
            ctx.set_profiling(False)
        ctx.sync()
        tracker_ms(ctx, dev, "after the headline steps (pairs alive)")
        ctx.set_tail_stream(None)
        tracker_ms(ctx, dev, "after set_tail_stream(None)")
        del keep, pairs, arr
        torch.cuda.empty_cache()
        tracker_ms(ctx, dev, "after freeing the pairs")
    if os.environ.get("SECONDARY"):
        out = bench.secondary_configs(dfx, synth, ctx, dev)
        print({k: {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("kernel_us", "frac", "round_us", "call_us")} for k, v in out.items()})
        tracker_ms(ctx, dev, "after secondary_configs")
    out = bench.small_operator_rooflines(dfx, synth, ctx, dev)
    print({k: round(v["frac"], 3) for k, v in out.items()})
    tracker_ms(ctx, dev, "after small_operator_rooflines")
    torch.cuda.empty_cache()
    tracker_ms(ctx, dev, "after empty_cache")
    ctx2 = dfx.Context(0)
    tracker_ms(ctx2, dev, "second context")


if __name__ == "__main__":
    main()
