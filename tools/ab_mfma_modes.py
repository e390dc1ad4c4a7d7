#!/usr/bin/env python3
"""One-process A/B of the step kernel's evaluation modes (fp32 chain vs exact bf16 split) on the same device-resident batch:
two contexts on the same stream, interleaved windows, kernel time from the library's HIP events.
usage: python tools/ab_mfma_modes.py [--pairs 128 --width 640 --height 480 --cs 32] [--rounds 3 --steps 60]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=128)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--cs", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--preroll", type=int, default=300)
    ap.add_argument("--schedule", default="static")
    ap.add_argument("--clone", action="store_true", help="generate 8 scenes and clone them into --pairs distinct allocations (fast set-up for A/B runs)")
    ap.add_argument("--foreign-valid0", action="store_true", help="valid0 maps in torch tensors (re-read every step) instead of library-owned images")
    a = ap.parse_args()
    import torch
    import deepfactors_amd as dfx
    from deepfactors_amd import _lib, synth
    dev = torch.device("cuda", 0)
    keep, pairs = [], []
    own = dfx.Context(0)   # owner of the library-owned valid0 maps (shadowed; shared by both contexts below)
    protos = {}
    for k in range(a.pairs):
        if a.clone:   # 8 generated scenes, cloned into distinct allocations (same HBM footprint and access pattern, 1/16 of the set-up time)
            if k % 8 not in protos:
                protos[k % 8] = synth.make_pair(a.width, a.height, a.cs, seed=0xDF02 + k, device=dev, motion_scale=0.6 + 0.05 * (k % 8))
            t = {n: (v.clone() if isinstance(v, torch.Tensor) else v) for n, v in protos[k % 8].items()}
        else:
            t = synth.make_pair(a.width, a.height, a.cs, seed=0xDF02 + k, device=dev, motion_scale=0.6 + 0.05 * (k % 8))
        if not a.foreign_valid0:
            t["valid0"] = own.alloc_image(a.width, a.height)
        keep.append(t)
        pairs.append(dict(pose0=t["pose0"], pose1=t["pose1"], cam=t["cam"], img0=t["img0"], img1=t["img1"], dpt0=t["dpt0"], prx0_jac=t["prx_jac"], grad1=t["grad1"],
                          valid0=t["valid0"]))
    modes = {"f32_chain": _lib.DFX_MFMA_F32_CHAIN, "bf16x3": _lib.DFX_MFMA_BF16X3}
    ctxs, als, arrs = {}, {}, {}
    items = torch.zeros(a.pairs * dfx.item_size(12 + a.cs), dtype=torch.uint8, device=dev)
    for name, m in modes.items():
        c = dfx.Context(0)
        c.set_mfma_mode(m)
        c.set_schedule(_lib.DFX_SCHEDULE_DYNAMIC if a.schedule == "dynamic" else _lib.DFX_SCHEDULE_STATIC)
        ctxs[name] = c
        als[name] = dfx.SfmAligner(code_size=a.cs, ctx=c)
        arrs[name] = als[name].make_pairs(pairs)
    first = {}
    for name in modes:   # results of both modes on pair 0 (sanity)
        als[name].RunStepBatchAsync(arrs[name], items)
        ctxs[name].sync()
        it = als[name].items_from_bytes(items.cpu().numpy(), a.cs)[0]
        first[name] = dict(inliers=int(it.inliers), residual=float(it.residual), jtj_max=float(abs(it.JtJ).max()))
    for _ in range(a.preroll):
        als["f32_chain"].RunStepBatchAsync(arrs["f32_chain"], items)
    ctxs["f32_chain"].sync()
    table = {n: [] for n in modes}
    for r in range(a.rounds):
        for name in modes:
            c, al, arr = ctxs[name], als[name], arrs[name]
            for _ in range(10):
                al.RunStepBatchAsync(arr, items)
            c.sync()
            c.set_profiling(True)
            for _ in range(a.steps):
                al.RunStepBatchAsync(arr, items)
            n, ms = c.profile_read()
            c.set_profiling(False)
            table[name].append(round(ms / n * 1e3, 1))
    bpl = (20 + 4 * a.cs) * a.width * a.height * a.pairs
    out = dict(config=dict(pairs=a.pairs, width=a.width, height=a.height, cs=a.cs, schedule=a.schedule), kernel_us=table, first_pair=first,
               frac_of_8TBs={n: round(bpl / (min(v) * 1e-6) / 8e12, 4) for n, v in table.items()})
    print("ABMODES " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
