"""Known-answer vectors produced by the REFERENCE'S OWN code (tests/golden/ref_vectors.npz, made by tests/golden/make_ref_vectors.py
from oracle/_ref = the reference's unmodified warping.h / dense_sfm.h / lucas_kanade_se3.h / pinhole_camera_impl.h / m_estimators.h):
inputs and outputs are both in the file, so these tests need neither /root/reference nor the prebuilt oracle/_ref library.

CPU (not gpu): the oracle reproduces them (the oracle's pin).  GPU: the HIP path through the C ABI reproduces them, at the
stated tolerance of tests/helpers.py (1e-4 of max|JtJ|; the reference's own GPU-vs-CPU bar is 1e-1 absolute, ut_sfmaligner.cpp:320-326)."""
import os
import types

import numpy as np
import pytest

from helpers import assert_item_close

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.npz")
CASES = ["a", "b"]


@pytest.fixture(scope="module")
def gold():
    z = np.load(GOLD)
    return {k: z[k] for k in z.files}


def _case(gold, name):
    g = {k[len(name) + 1:]: v for k, v in gold.items() if k.startswith(name + "_")}
    h, w = g["img0"].shape
    return g, w, h, int(g["cs"]), float(g["huber"])


def _item(g, prefix):
    return types.SimpleNamespace(JtJ=g[prefix + "_JtJ"], Jtr=g[prefix + "_Jtr"], residual=float(g[prefix + "_residual"]), inliers=int(g[prefix + "_inliers"]))


def test_vectors_come_from_the_reference_headers(gold):
    src = bytes(gold["sources"]).decode()
    assert "dense_sfm" in src and "lucas_kanade_se3" in src and "warping" in src and "unmodified" in src
    for name in CASES:
        g, w, h, cs, _ = _case(gold, name)
        assert g["prx_jac"].shape == (h, w * cs) and g["sfm_JtJ"].shape == ((12 + cs) * (13 + cs) // 2,)
        assert g["sfm_inliers"] > 0.5 * w * h and g["sfm_valid0"].sum() == g["sfm_inliers"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_the_reference_vectors(oracle, gold, name):
    g, w, h, cs, huber = _case(gold, name)
    v = np.zeros((h, w), np.float32)
    got = oracle.sfm_step(g["pose0"], g["pose1"], g["cam"], g["img0"], g["img1"], g["dpt0"], g["prx_jac"], g["grad1"], huber_delta=huber, valid0=v,
                          accum_f64=True)
    assert_item_close(got, _item(g, "sfm"), w, h, rel=2e-5, what=f"oracle sfm_step vs reference vectors ({name})")
    assert got.inliers == int(g["sfm_inliers"]) and np.array_equal(v.astype(np.uint8), g["sfm_valid0"])
    e_res, e_inl = oracle.sfm_error(g["pose0"], g["pose1"], g["cam"], g["img0"], g["img1"], g["dpt0"], huber_delta=huber)
    assert e_inl == int(g["err_inliers"]) and abs(e_res - float(g["err_residual"])) <= 2e-5 * float(g["err_residual"])
    R, t, ja, jb = oracle.relative_pose(g["pose1"], g["pose0"])          # RelativePose(pose1, pose0, J1, J0), cu_sfmaligner.cpp:166
    assert np.abs(R - oracle.quat_to_R(g["rel_pose"][:4])).max() <= 1e-6 and np.abs(t - g["rel_pose"][4:]).max() <= 1e-6
    assert np.abs(ja - g["rel_Ja"]).max() <= 1e-6 and np.abs(jb - g["rel_Jb"]).max() <= 1e-6
    k = oracle.se3_step(g["rel_pose"], g["cam"], g["img0"], g["img1"], g["dpt0"], g["grad1"], huber)
    assert_item_close(k, _item(g, "se3"), w, h, rel=2e-5, what=f"oracle se3_step vs reference vectors ({name})")
    d = oracle.update_depth(g["code"], g["prx_orig"], g["prx_jac"], 2.0)
    assert np.abs(d - g["dpt"]).max() <= 2e-6 * float(((2.0 + g["dpt"]) ** 2 / 2.0).max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_path_reproduces_the_reference_vectors(dfx, gold, name):
    import torch
    g, w, h, cs, huber = _case(gold, name)
    dev = {k: torch.from_numpy(np.ascontiguousarray(g[k])).cuda() for k in ("img0", "img1", "dpt0", "grad1", "prx_orig", "prx_jac")}
    al = dfx.SfmAligner(dfx.SfmAlignerParams(dfx.DenseSfmParams(huber_delta=huber)), code_size=cs)
    valid = torch.zeros((h, w), dtype=torch.float32, device="cuda")
    got = al.RunStep(g["pose0"], g["pose1"], g["code"], g["cam"], dev["img0"], dev["img1"], dev["dpt0"], None, valid, dev["prx_jac"], dev["grad1"])
    assert_item_close(got, _item(g, "sfm"), w, h, what=f"HIP sfm_step vs reference vectors ({name})")
    assert int((valid.cpu().numpy().astype(np.uint8) != g["sfm_valid0"]).sum()) <= 1
    e = al.EvaluateError(g["pose0"], g["pose1"], g["cam"], dev["img0"], dev["img1"], dev["dpt0"], None, dev["grad1"])
    assert abs(int(e.inliers) - int(g["err_inliers"])) <= 1 and abs(e.residual - float(g["err_residual"])) <= 1e-4 * float(g["err_residual"])
    se3 = dfx.SE3Aligner()
    se3.SetHuberDelta(huber)
    k = se3.RunStep(g["rel_pose"], g["cam"], dev["img0"], dev["img1"], dev["dpt0"], dev["grad1"])
    assert_item_close(k, _item(g, "se3"), w, h, what=f"HIP se3_step vs reference vectors ({name})")
    out = torch.empty((h, w), dtype=torch.float32, device="cuda")
    dfx.UpdateDepth(g["code"], dev["prx_orig"], dev["prx_jac"], 2.0, out)
    assert np.abs(out.cpu().numpy() - g["dpt"]).max() <= 2e-6 * float(((2.0 + g["dpt"]) ** 2 / 2.0).max())


# ---- the reference's SparseGeometricFactor::linearize and DepthAligner kernel on stored inputs (ref_vectors_f3.npz) -----------
GOLD_F3 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors_f3.npz")


@pytest.fixture(scope="module")
def gold_f3():
    z = np.load(GOLD_F3)
    return {k: z[k] for k in z.files}


def _check_rows(got, want):
    zg, zw = ~np.asarray(got, np.float64).any(axis=1), ~want.any(axis=1)
    assert int((zg != zw).sum()) <= 1 and 0 < zw.sum() < len(want)
    same = zg == zw
    scale = np.abs(want).max(axis=0) + 1e-12
    return float((np.abs(np.asarray(got, np.float64)[same] - want[same]) / scale).max())


def test_oracle_reproduces_the_reference_f3_vectors(oracle, gold_f3):
    g = gold_f3
    assert b"sparse_geometric_factor.cpp" in bytes(g["sources"]) and g["sg_rows"].shape == (len(g["points"]), 77)
    rows = oracle.sparse_geometric(g["pose0"], g["pose1"], g["code0"], g["code1"], g["cam"], g["points"], g["prx0"], g["jac0"], g["prx1"], g["jac1"], g["dgrad1"],
                                   float(g["huber"]), avg_dpt=2.0)
    assert _check_rows(rows, g["sg_rows"]) <= 2e-5
    d = oracle.depth_aligner_step(g["da_code"], g["da_tgt"], g["prx0"], g["jac0"], 2.0, accum_f64=True)
    sj = float(np.abs(g["da_JtJ"]).max())
    assert d.inliers == int(g["da_inliers"]) and np.abs(d.JtJ - g["da_JtJ"]).max() <= 2e-4 * sj and abs(d.residual - float(g["da_residual"])) <= 2e-4 * float(g["da_residual"])


@pytest.mark.gpu
def test_hip_path_reproduces_the_reference_f3_vectors(dfx, gold_f3):
    import torch
    g = gold_f3
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()   # noqa: E731
    fac = dfx.SparseGeometricFactor(g["cam"], g["points"], dict(prx_orig=t(g["prx0"]), prx_jac=t(g["jac0"])),
                                    dict(prx_orig=t(g["prx1"]), prx_jac=t(g["jac1"]), dpt_grad=t(g["dgrad1"])), float(g["huber"]), code_size=32)
    rows = fac.linearize(g["pose0"], g["pose1"], g["code0"], g["code1"])
    assert _check_rows(rows, g["sg_rows"]) <= 2e-4
    da = dfx.DepthAligner(code_size=32)
    got = da.RunStep(g["da_code"], t(g["da_tgt"]), t(g["prx0"]), t(g["jac0"]), 2.0)
    sj = float(np.abs(g["da_JtJ"]).max())
    assert got.inliers == int(g["da_inliers"])
    assert np.abs(got.JtJ.astype(np.float64) - g["da_JtJ"]).max() <= 2e-4 * sj        # the reference item itself is a float sum over w*h pixels
    assert np.abs(got.Jtr.astype(np.float64) - g["da_Jtr"]).max() <= 2e-4 * max(float(np.abs(g["da_Jtr"]).max()), float(np.sqrt(sj * float(g["da_residual"]))))
    assert abs(got.residual - float(g["da_residual"])) <= 2e-4 * float(g["da_residual"])


# ---- f1 (pyramid construction) and a4 (SE3Aligner::Warp): outputs of the reference's kernel bodies, cut out at build time ---------------------
GOLD_F1 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors_f1.npz")


@pytest.fixture(scope="module")
def gold_f1():
    z = np.load(GOLD_F1)
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", ["a", "b"])
def test_oracle_reproduces_the_reference_f1_vectors(oracle, gold_f1, name):
    g = {k[2:]: v for k, v in gold_f1.items() if k.startswith(name + "_")}
    assert b"kernel_sobel_gradients" in bytes(gold_f1["sources"])
    h, w = g["img0"].shape
    assert np.array_equal(oracle.sobel(g["img0"]), g["sobel"]) and np.array_equal(oracle.blur_down(g["img0"]), g["blur"])
    assert oracle.squared_error(g["img0"], g["img1"], accum_f64=False) == float(g["sqerr"])
    img2, r, k = oracle.se3_warp(g["pose10"], g["cam"], g["img0"], g["img1"], g["dpt0"], accum_f64=False)
    assert abs(k - int(g["warp_inliers"])) <= 1 and int((np.abs(img2 - g["warp_img2"]) > 1e-5).sum()) <= 2
    assert abs(r - float(g["warp_residual"])) <= 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["a", "b"])
def test_hip_path_reproduces_the_reference_f1_vectors(dfx, gold_f1, name):
    import torch
    g = {k[2:]: v for k, v in gold_f1.items() if k.startswith(name + "_")}
    h, w = g["img0"].shape
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()   # noqa: E731
    i0, i1, d0 = t(g["img0"]), t(g["img1"]), t(g["dpt0"])
    grad = torch.empty((h, w, 2), dtype=torch.float32, device="cuda")
    dfx.SobelGradients(i0, grad)
    assert np.array_equal(grad.cpu().numpy(), g["sobel"])                                   # exact: taps x1, x2, /8
    half = torch.empty((h // 2, w // 2), dtype=torch.float32, device="cuda")
    dfx.GaussianBlurDown(i0, half)
    assert np.abs(half.cpu().numpy() - g["blur"]).max() <= 2e-7                                # 25 taps, one rounding each
    se = dfx.SquaredError(i0, i1)
    assert abs(se - float(g["sqerr"])) <= 2e-5 * float(g["sqerr"])                             # the reference's own float sum in pixel order
    img2 = torch.empty_like(i0)
    got = dfx.SE3Aligner().Warp(g["pose10"], g["cam"], i0, i1, d0, img2)
    assert abs(int(got.inliers) - int(g["warp_inliers"])) <= 1
    assert int((np.abs(img2.cpu().numpy() - g["warp_img2"]) > 1e-5).sum()) <= 2
    assert abs(got.residual - float(g["warp_residual"])) <= 1e-3
