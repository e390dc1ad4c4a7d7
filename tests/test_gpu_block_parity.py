"""Per-block parity of SfmAligner::RunStep (k_sfm_step, both MFMA evaluation modes) against the fp64 oracle, on inputs WITH DYNAMIC RANGE.

What GTSAM receives from a PhotometricFactor is six blocks G11 G12 G13 G22 G23 G33 and three gradients (photometric_factor.cpp:135-161);
on the default synthetic pair they span six orders of magnitude (pose-pose 1e6, pose-code 1e2, code-code 1e0), so a tolerance scaled by the
largest entry of the whole matrix pins only the 78 pose-pose entries.  Every comparison here is per entry at the entry's Cauchy-Schwarz
scale sqrt(JtJ_ii JtJ_jj) (tests/helpers.py), on:

  * jac_amp 0.05 / 5 / 50 / 500: code-code = 1e-6 / 1e-2 / 1 / 1e2 x pose-pose (equal magnitude at 50, code-dominant at 500) (the bf16 split must not need a rescue at
    any amplitude: its three-way split is relative to each operand, include/dfx.h DFX_MFMA_BF16X3);
  * a depth decoded from a code != the generating one, so the code gradient Jtr[12:] is far from zero;
  * img1 / grad1 of ANOTHER scene: |r| > huber_delta on > 30 % of the inliers (asserted), so the Huber branch carries the sums;
  * 640x480x32 and 1280x960x64, single pair and batch, DFX_MFMA_F32_CHAIN and DFX_MFMA_BF16X3.

The module writes its measurements to gpurun_out/r06_block_parity.txt (copied to profiles/ by hand)."""
import os

import numpy as np
import pytest
import torch

from helpers import assert_blocks_below, assert_item_close, format_block_errors

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = ("f32", "bf16x3")


@pytest.fixture(scope="module")
def report():
    rows = []
    yield rows
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "r06_block_parity.txt"), "w") as f:
        f.write("Per-block parity of k_sfm_step vs the fp64-accumulating oracle (tests/test_gpu_block_parity.py).\n"
                "cs  = max over the block's entries of |got_ij - ref_ij| / sqrt(ref_ii ref_jj)   (Jtr: / sqrt(ref_ii sum r^2))\n"
                "blk = max|got - ref| over the block / max|ref| over the block\n\n")
        for r in rows:
            f.write(r + "\n")


def _ctx(dfx, mode):
    from deepfactors_amd import _lib
    ctx = dfx.Context(0)
    ctx.set_mfma_mode(_lib.DFX_MFMA_BF16X3 if mode == "bf16x3" else _lib.DFX_MFMA_F32_CHAIN)
    return ctx


def _rel_pose_qt(synth, pose0, pose1):
    """pose_10 = pose1^-1 * pose0 (warping.h:98-103) as (q, t), for the oracle's Warp."""
    R0, R1 = synth.quat_to_R(pose0[:4]), synth.quat_to_R(pose1[:4])
    R10 = R1.T @ R0
    t10 = R1.T @ (np.asarray(pose0[4:], np.float64) - np.asarray(pose1[4:], np.float64))
    return synth.pose_qt(R10, t10)


def _scenario(oracle, w, h, cs, jac_amp, kind, seed):
    """(host dict, device dict, pose1, expected-huber-fraction or None)"""
    from deepfactors_amd import synth
    dev = "cuda"
    p = synth.make_pair(w, h, cs, seed=seed, device=dev, jac_amp=jac_amp, code_sigma=0.3 * 0.05 / jac_amp)
    if kind == "unrelated":
        q = synth.make_pair(w, h, cs, seed=seed + 7777, device=dev, with_decoder=False, motion_scale=2.0)
        # another scene, contrast stretched x2.5 about mid-grey (the synthetic texture is a low-contrast sum of 24 sinusoids), gradient
        # recomputed with the reference's Sobel / 8 taps -- so |r| > huber_delta on well over 30 % of the pixels
        p["img1"] = (0.5 + 2.5 * (q["img1"] - 0.5)).clamp(0.0, 1.0).contiguous()
        p["grad1"] = synth.sobel_torch(p["img1"])
    n = synth.to_numpy(p)
    if kind == "code_offset":
        rng = np.random.default_rng(seed + 1)
        code = (n["code"] + rng.normal(0, 0.5 * np.abs(n["code"]).max() + 1e-9, cs)).astype(np.float32)
        n["dpt0"] = oracle.update_depth(code, n["prx_orig"], n["prx_jac"], 2.0)
        assert np.isfinite(n["dpt0"]).all() and n["dpt0"].min() > 0.3
        p["dpt0"] = torch.from_numpy(n["dpt0"]).to(dev)
    pose1 = n["pose1"].copy(); pose1[4] += 0.01
    return n, p, pose1


def _huber_fraction(oracle, synth, n, pose1, delta):
    img2, _, inl = oracle.se3_warp(_rel_pose_qt(synth, n["pose0"], pose1), n["cam"], n["img0"], n["img1"], n["dpt0"])
    # Warp zeroes img2 where there is no correspondence (cu_se3aligner.cpp:61-113); synthetic img1 is >= 0.0 only by accident, so take
    # the inlier count from the item and count |r| > delta over pixels whose warped sample is non-zero
    r = np.abs(n["img0"] - img2)[img2 != 0]
    return float((r > delta).sum()) / max(inl, 1)


@pytest.mark.parametrize("cs", [16, 32, 64])
def test_block_report_both_modes(dfx, oracle, report, cs):
    """the per-block numbers for CS 16 / 32 / 64 in both modes, each block below 1e-5 (the tolerance is 1e-4)."""
    w, h = 320, 240
    from deepfactors_amd import synth
    p = synth.make_pair(w, h, cs, seed=0xDF02 + 320, device="cuda")
    n = synth.to_numpy(p)
    pose1 = n["pose1"].copy(); pose1[4] += 0.01
    ref = oracle.sfm_step(n["pose0"], pose1, n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"], threads=8)
    for mode in MODES:
        al = dfx.SfmAligner(code_size=cs, ctx=_ctx(dfx, mode))
        got = al.RunStep(n["pose0"], pose1, n["code"], n["cam"], p["img0"], p["img1"], p["dpt0"], None, None, p["prx_jac"], p["grad1"])
        assert got.inliers == ref.inliers
        errs = assert_blocks_below(got, ref, 1e-5, what=f"{mode} cs={cs}")
        report.append(f"default pair {w}x{h} cs={cs} mode={mode}: {format_block_errors(errs)}")


@pytest.mark.parametrize("kind", ["truth", "code_offset", "unrelated"])
@pytest.mark.parametrize("jac_amp", [0.05, 5.0, 50.0, 500.0])
@pytest.mark.parametrize("w,h,cs", [(640, 480, 32), (1280, 960, 64)])
def test_dynamic_range_single_and_batch(dfx, oracle, report, w, h, cs, jac_amp, kind):
    from deepfactors_amd import synth
    delta = 0.1
    n, p, pose1 = _scenario(oracle, w, h, cs, jac_amp, kind, seed=0x6B00 + cs)
    ref = oracle.sfm_step(n["pose0"], pose1, n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"], huber_delta=delta, threads=8)
    assert ref.inliers > 0.7 * w * h
    if kind == "unrelated":
        frac = _huber_fraction(oracle, synth, n, pose1, delta)
        assert frac > 0.30, f"Huber branch active on only {frac:.2f} of the inliers"
    else:
        frac = None
    # the magnitudes this case is about: ratio of the code-code to the pose-pose block
    M = ref.dense()
    ratio = float(np.abs(M[12:, 12:]).max() / np.abs(M[:12, :12]).max())
    if kind == "code_offset":
        assert np.abs(ref.Jtr[12:]).max() > 1e-2 * np.sqrt(np.diag(M)[12:].max() * ref.residual)   # a real code gradient
    for mode in MODES:
        ctx = _ctx(dfx, mode)
        al = dfx.SfmAligner(code_size=cs, ctx=ctx)
        got = al.RunStep(n["pose0"], pose1, n["code"], n["cam"], p["img0"], p["img1"], p["dpt0"], None, None, p["prx_jac"], p["grad1"])
        what = f"{kind} {w}x{h} cs={cs} jac_amp={jac_amp:g} mode={mode}"
        errs = assert_item_close(got, ref, w, h, what=what)
        report.append(f"{what} (code/pose block ratio {ratio:.1e}" + (f", huber fraction {frac:.2f}" if frac is not None else "") + f"): {format_block_errors(errs)}")
        arr = al.make_pairs([dict(pose0=n["pose0"], pose1=pose1, cam=n["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"], prx0_jac=p["prx_jac"],
                                  grad1=p["grad1"])] * 3)
        for k, it in enumerate(al.RunStepBatch(arr)):
            assert_item_close(it, ref, w, h, what=f"batch item {k}: {what}")
