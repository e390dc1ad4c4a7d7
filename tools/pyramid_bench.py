#!/usr/bin/env python3
"""Frame ingest: dfx_build_pyramid_batch_async over F frames of 640x480 (4 levels) -- event time per enqueue and roofline fraction (bytes: 4 read + 8 + 1 written
per pixel and level), against the per-level operators frame by frame.  usage: pyramid_bench.py [frames=64] [--build-only]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    import deepfactors_amd as dfx
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    F = int(args[0]) if args else 64
    W, H, LV = 640, 480, 4
    dev = torch.device("cuda", 0)
    ctx = dfx.Context(0)
    pyr_i = [[torch.rand((H >> i, W >> i), dtype=torch.float32, device=dev) for i in range(LV)] for _ in range(F)]
    pyr_g = [[torch.empty((H >> i, W >> i, 2), dtype=torch.float32, device=dev) for i in range(LV)] for _ in range(F)]
    arr = dfx.make_pyramids(pyr_i, pyr_g)

    def timed(fn, reps, warm):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    us = timed(lambda: dfx.BuildPyramids(arr, ctx=ctx), 50, 200)
    byts = sum((W >> i) * (H >> i) * (12 + (1 if i + 1 < LV else 0)) for i in range(LV)) * F
    print(f"build_pyramid_batch {F} frames x {LV} levels: {us:.1f} us per enqueue = {byts / us / 1e3:.0f} GB/s = {byts / us / 1e3 / 8000:.3f} of 8 TB/s ({us / F:.2f} us per frame)")

    if "--build-only" in sys.argv:
        return

    # the yardstick: a device-to-device copy moving the volume of the level-0 launch (F x (4 B read + 8 + 1 B written) per pixel ~ half read, half written)
    vol = F * W * H * 13
    a = torch.empty(vol // 8, dtype=torch.float32, device=dev); b = torch.empty_like(a)
    cus = timed(lambda: b.copy_(a), 50, 50)
    print(f"torch copy_ of {vol / 2e6:.0f} MB -> {vol / 2e6:.0f} MB (the level-0 launch's volume): {cus:.1f} us = {vol / cus / 1e3:.0f} GB/s read + written")

    def per_level():
        for k in range(min(F, 8)):
            for i in range(LV):
                if i > 0:
                    dfx.GaussianBlurDown(pyr_i[k][i - 1], pyr_i[k][i], ctx)
                dfx.SobelGradients(pyr_i[k][i], pyr_g[k][i], ctx)
    us8 = timed(per_level, 5, 3)
    print(f"per-level blocking operators: {us8 / min(F, 8):.1f} us per frame")


if __name__ == "__main__":
    main()
