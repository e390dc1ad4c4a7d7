#!/usr/bin/env python3
"""Diagnosis aid: the batch of tests/test_gpu_configs.py::test_regions_out_of_view_are_deterministic_and_match_the_oracle run several times;
prints, per run, which items / entries differ from run 0 and by how much.  usage: python tools/diag_nan_batch.py [--mode bf16x3|f32] [--runs 6]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="bf16x3")
    ap.add_argument("--runs", type=int, default=6)
    ap.add_argument("--blocks", type=int, default=0)
    ap.add_argument("--nonan", action="store_true")
    a = ap.parse_args()
    import torch
    import deepfactors_amd as dfx
    from deepfactors_amd import _lib, synth
    w, h, cs = 640, 480, 32
    p = synth.make_pair(w, h, cs, seed=0xDF08, device="cuda"); n = synth.to_numpy(p)
    p2 = synth.make_pair(w, h, cs, seed=0xDF09, device="cuda"); n2 = synth.to_numpy(p2)
    R = synth.so3_exp(np.array([0.0, 0.35, 0.02]))
    pose1 = synth.pose_qt(R.T, -R.T @ np.array([0.4, 0.05, 0.0]))
    dpt = p["dpt0"].clone()
    if not a.nonan:
        dpt[100:140, :] = float("nan")
        dpt[300:330, 200:520] = float("nan")
    ctx = dfx.Context()
    ctx.set_schedule(_lib.DFX_SCHEDULE_STATIC)
    ctx.set_mfma_mode(_lib.DFX_MFMA_BF16X3 if a.mode == "bf16x3" else _lib.DFX_MFMA_F32_CHAIN)
    al = dfx.SfmAligner(dfx.SfmAlignerParams(step_blocks=a.blocks), code_size=cs, ctx=ctx)
    bad = dict(pose0=n["pose0"], pose1=pose1, cam=n["cam"], img0=p["img0"], img1=p["img1"], dpt0=dpt, prx0_jac=p["prx_jac"], grad1=p["grad1"])
    good = dict(pose0=n2["pose0"], pose1=n2["pose1"], cam=n2["cam"], img0=p2["img0"], img1=p2["img1"], dpt0=p2["dpt0"], prx0_jac=p2["prx_jac"], grad1=p2["grad1"])
    arr = al.make_pairs([good, bad, good, bad, bad, good, bad, good] * 4)
    NP = 12 + cs
    tri = [(i, j) for i in range(NP) for j in range(i, NP)]
    base = None
    for r in range(a.runs):
        items = al.RunStepBatch(arr)
        raw = np.stack([it.raw for it in items])
        if base is None:
            base = raw; base_items = items
            print("run 0: inliers", [it.inliers for it in items[:4]])
            continue
        bad_items = [k for k in range(len(items)) if not np.array_equal(raw[k], base[k])]
        print(f"run {r}: {len(bad_items)} items differ: {bad_items}")
        for k in bad_items[:4]:
            d = np.abs(items[k].JtJ.astype(np.float64) - base_items[k].JtJ.astype(np.float64))
            idx = np.nonzero(d)[0]
            scale = np.abs(base_items[k].JtJ).max()
            where = sorted({("pp" if tri[i][1] < 12 else ("pc" if tri[i][0] < 12 else ("cc_diag" if (tri[i][0] - 12) % 2 == (tri[i][1] - 12) % 2 else "cc_off"))) for i in idx})
            dj = np.abs(items[k].Jtr.astype(np.float64) - base_items[k].Jtr.astype(np.float64))
            print(f"   item {k}: {len(idx)} JtJ entries differ, max rel {d.max() / scale:.3e}, blocks {where}; first entries {[tri[i] for i in idx[:6]]}; Jtr differing {np.nonzero(dj)[0][:8].tolist()}; "
                  f"residual {items[k].residual!r} vs {base_items[k].residual!r}; inliers {items[k].inliers} vs {base_items[k].inliers}")


if __name__ == "__main__":
    main()
