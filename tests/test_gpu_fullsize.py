"""Full-size (BASELINE.json configs) checks through size-independent properties of the path (the direct comparison with the
CPU oracle at these sizes, incl. 1280x960x64, is tests/test_gpu_configs.py):
  * masking additivity: the sums over a full image equal the sums over its top half plus its bottom half, where "half" is
    selected by making the other half's depth NaN (NaN depth = no correspondence, warping.h:221-224);
  * Cauchy-Schwarz / PSD structure of the normal equations; determinism; batch == single;
  * decoder linearity in proximity space (ut_decoder.cpp:161-199) at full size;
  * Warp at the identity pose reproduces img1 on the valid interior (cu_se3aligner.cpp:61-113)."""
import numpy as np
import pytest
import torch

from helpers import assert_blocks_below, item_sum

pytestmark = pytest.mark.gpu


def _dev_pair(w, h, cs, seed):
    from deepfactors_amd import synth
    return synth.make_pair(w, h, cs, seed=seed, device="cuda")


def _step(dfx, al, p, dpt):
    return al.RunStep(p["pose0"], p["pose1"], p["code"], p["cam"], p["img0"], p["img1"], dpt, None, None, p["prx_jac"], p["grad1"])


@pytest.mark.parametrize("w,h,cs", [(640, 480, 32), (1280, 960, 64)])
def test_masking_additivity_and_structure(dfx, w, h, cs):
    p = _dev_pair(w, h, cs, seed=0xDF05 if cs == 64 else 0xDF02)
    p["pose1"] = p["pose1"].copy(); p["pose1"][4] += 0.01
    al = dfx.SfmAligner(code_size=cs)
    full = _step(dfx, al, p, p["dpt0"])
    top = p["dpt0"].clone(); top[h // 2:, :] = float("nan")
    bot = p["dpt0"].clone(); bot[: h // 2, :] = float("nan")
    a, b = _step(dfx, al, p, top), _step(dfx, al, p, bot)
    assert full.inliers == a.inliers + b.inliers and full.inliers > 0.8 * w * h
    # additivity per block: each entry of (top + bottom) against the full image at that entry's Cauchy-Schwarz scale
    assert_blocks_below(item_sum(a, b), full, 5e-6, what=f"additivity {w}x{h} cs={cs}")
    assert abs((a.residual + b.residual) - full.residual) <= 2e-6 * full.residual
    # structure of a Gauss-Newton system: PSD, Cauchy-Schwarz between Jtr, diag(JtJ) and the residual
    M = full.toDenseMatrix().astype(np.float64)
    ev = np.linalg.eigvalsh(M)
    assert ev.min() >= -1e-5 * ev.max()
    assert np.all(np.abs(full.Jtr.astype(np.float64)) <= np.sqrt(np.diag(M) * full.residual) * (1 + 1e-4) + 1e-6)
    # determinism + batch == single
    again = _step(dfx, al, p, p["dpt0"])
    assert np.array_equal(full.raw, again.raw)
    arr = al.make_pairs([dict(pose0=p["pose0"], pose1=p["pose1"], cam=p["cam"], img0=p["img0"], img1=p["img1"], dpt0=p["dpt0"],
                              prx0_jac=p["prx_jac"], grad1=p["grad1"])] * 2)
    it = al.RunStepBatch(arr)
    assert it[0].inliers == it[1].inliers == full.inliers
    assert_blocks_below(it[0], full, 5e-6, what="batch of two vs single")
    assert np.array_equal(it[0].JtJ, it[1].JtJ)


@pytest.mark.parametrize("w,h,cs", [(640, 480, 32), (1280, 960, 64)])
def test_decoder_linearity_full_size(dfx, w, h, cs):
    p = _dev_pair(w, h, cs, seed=9)
    a = 2.0
    rng = np.random.default_rng(4)
    c1, c2 = rng.normal(0, 0.3, cs).astype(np.float32), rng.normal(0, 0.3, cs).astype(np.float32)
    outs = []
    for c in (np.zeros(cs, np.float32), c1, c2, c1 + c2):
        o = torch.empty_like(p["img0"])
        dfx.UpdateDepth(c, p["prx_orig"], p["prx_jac"], a, o)
        outs.append((a / (a + o.double())))          # back to proximity: prx = a / (a + dpt)
    p0, p1, p2, p12 = outs
    assert float((p1 + p2 - p0 - p12).abs().max()) <= 5e-6
    assert float((p0 - p["prx_orig"].double()).abs().max()) <= 1e-6


def test_warp_identity_reproduces_img1(dfx):
    from deepfactors_amd import synth
    w, h = 640, 480
    p = _dev_pair(w, h, 16, seed=10)
    al = dfx.SE3Aligner()
    img2 = torch.full_like(p["img0"], -1.0)
    r = al.Warp(synth.IDENTITY, p["cam"], p["img0"], p["img1"], p["dpt0"], img2)
    inner = img2[2:-2, 2:-2]
    assert float((inner - p["img1"][2:-2, 2:-2]).abs().max()) <= 1e-5
    assert r.inliers >= (w - 4) * (h - 4)
    assert float(img2.min()) >= 0.0   # everything was written (0 where invalid)


def test_se3_step_full_size_additivity(dfx):
    from deepfactors_amd import synth
    w, h = 640, 480
    p = _dev_pair(w, h, 16, seed=12)
    al = dfx.SE3Aligner()
    full = al.RunStep(p["pose10_true"], p["cam"], p["img0"], p["img1"], p["dpt0"], p["grad1"])
    left = p["dpt0"].clone(); left[:, w // 2:] = float("nan")
    right = p["dpt0"].clone(); right[:, : w // 2] = float("nan")
    a = al.RunStep(p["pose10_true"], p["cam"], p["img0"], p["img1"], left, p["grad1"])
    b = al.RunStep(p["pose10_true"], p["cam"], p["img0"], p["img1"], right, p["grad1"])
    assert full.inliers == a.inliers + b.inliers
    assert np.abs((a.JtJ.astype(np.float64) + b.JtJ) - full.JtJ).max() <= 3e-6 * float(np.abs(full.JtJ).max())
