"""Pins the CPU oracle (oracle/dfx_oracle.cpp, our restatement) to the REFERENCE'S OWN per-pixel code: oracle/_ref/libdfx_ref.so is
built from the reference's unmodified L0 headers (warping.h, dense_sfm.h, lucas_kanade_se3.h, pinhole_camera{,_impl}.h,
m_estimators.h, reduction_items.h, kernel_utils.h -- oracle/ref_harness.cpp) against stand-in Eigen / Sophus / VisionCore headers.
Reference-derived through this pin: every formula of the path (RelativePose and its Jacobians, correspondence + validity,
projection and warp Jacobians, DepthJacobianPrx, the 12 + CS column row, Huber weight, the accumulated products, the GN update).
Still fixed by spec only: bilinear sampling and the packed upper-triangular order (stand-in VisionCore).

Differences allowed: the oracle forms R (3x3) from the quaternion in double and rounds once, Sophus rotates through the quaternion
in float -- projected coordinates differ by ~1 ulp, so a pixel that lands within that distance of the validity border may flip,
and sums differ at the 1e-6 level."""
import os

import numpy as np
import pytest

from helpers import assert_item_close, block_errors, format_block_errors
from test_oracle_kat import load_fixture, scenenet_cam

ref = pytest.importorskip("oracle.dfx_ref")
pytestmark = pytest.mark.skipif(not (ref.available() or os.path.isdir(os.path.join(ref.REFERENCE, "sources"))),
                                reason="oracle/_ref not built and no reference tree to build it from")


@pytest.fixture(scope="module")
def refl():
    ref.build()
    ref.lib()
    return ref


def _pair(w, h, cs, seed, **kw):
    from deepfactors_amd import synth
    return synth.to_numpy(synth.make_pair(w, h, cs, seed=seed, device="cpu", **kw))


def _close(got, want, w, h, rel=2e-5):
    """Per entry, at the entry's own Cauchy-Schwarz scale (tests/helpers.py) -- pose-code / code-code / code gradient included."""
    assert_item_close(got, want, w, h, rel=rel, what="oracle vs reference code")


@pytest.mark.parametrize("w,h,cs,seed", [(160, 120, 32, 11), (96, 64, 16, 12), (128, 96, 64, 13), (320, 240, 32, 14), (100, 77, 32, 15)])
def test_sfm_step_oracle_equals_reference_code(oracle, refl, w, h, cs, seed):
    n = _pair(w, h, cs, seed)
    pose1 = n["pose1"].copy(); pose1[4] += 0.01
    v_o, v_r = np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)
    want = refl.sfm_step(n["pose0"], pose1, n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"], valid0=v_r)
    got = oracle.sfm_step(n["pose0"], pose1, n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"], valid0=v_o, accum_f64=True)
    assert want.inliers > 0.5 * w * h
    _close(got, want, w, h)
    assert int((v_o != v_r).sum()) <= max(1, int(1e-5 * w * h))


def test_sfm_step_reference_test_poses_and_huber(oracle, refl):
    """ut_sfmaligner.cpp:254-268: pose0 = I, pose1 = inverse(exp(0.1, 0.1, 0), t = (-.5, -.5, 0)) (scaled by 0.1 for overlap), huber 0.5;
    plus non-default min_dpt / border."""
    from deepfactors_amd import synth
    w, h, cs = 192, 144, 32
    n = _pair(w, h, cs, 21)
    R = synth.so3_exp(np.array([0.1, 0.1, 0.0]) * 0.1)
    t = np.array([-0.5, -0.5, 0.0]) * 0.1
    pose1 = synth.pose_qt(R.T, -R.T @ t)
    for kw in (dict(huber_delta=0.5), dict(huber_delta=0.02, min_dpt=1.5, valid_border=7), dict(avg_dpt=3.0)):
        want = refl.sfm_step(n["pose0"], pose1, n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"], **kw)
        got = oracle.sfm_step(n["pose0"], pose1, n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"], **kw)
        assert want.inliers > 0
        _close(got, want, w, h)


def test_single_pixel_items_match(oracle, refl):
    """Per-pixel (J J^T, J^T r, r^2, valid): everything but one pixel masked by NaN depth (NaN depth = no correspondence,
    warping.h:221-224), so the sum IS that pixel's item -- 60 pixels incl. the border band where validity is decided."""
    w, h, cs = 48, 40, 16
    n = _pair(w, h, cs, 31)
    rng = np.random.default_rng(5)
    pix = [(int(rng.integers(0, w)), int(rng.integers(0, h))) for _ in range(40)] + [(x, y) for x in (0, 1, 2, 3, w - 3, w - 1) for y in (0, 2, h - 3)][:20]
    nvalid = 0
    for (x, y) in pix:
        d = np.full((h, w), np.nan, np.float32)
        d[y, x] = n["dpt0"][y, x]
        want = refl.sfm_step(n["pose0"], n["pose1"], n["cam"], n["img0"], n["img1"], d, n["prx_jac"], n["grad1"])
        got = oracle.sfm_step(n["pose0"], n["pose1"], n["cam"], n["img0"], n["img1"], d, n["prx_jac"], n["grad1"])
        assert got.inliers == want.inliers, (x, y)
        nvalid += want.inliers
        if want.inliers:
            # one pixel: sqrt(JtJ_ii JtJ_jj) = |J_i J_j| would be a relative check of single products, and a row entry that is itself
            # a cancelling sum (-g C J1) legitimately differs by 1e-4 of ITS value between two fp32 evaluation orders -- so here each of
            # the nine blocks is compared at the block's own maximum instead (still never at the whole matrix's)
            errs = block_errors(got, want)
            assert all(v["blk"] <= 5e-5 for v in errs.values()), ((x, y), format_block_errors(errs))
            # r = img0 - bilinear(img1) carries a few ulp(1) whatever its size: d(r^2) <= 2 |r| * 8 ulp
            assert abs(got.residual - want.residual) <= 5e-5 * want.residual + 16 * np.finfo(np.float32).eps * np.sqrt(want.residual)
    assert 20 <= nvalid < len(pix)   # both valid and invalid pixels were exercised


def test_se3_step_and_error_on_the_reference_fixture(oracle, refl):
    img0, img1, dpt0 = load_fixture()
    cam = scenenet_cam(320, 240)
    grad1 = oracle.sobel(img1)
    for qt in (np.array([0, 0, 0, 1, 0, 0, 0], np.float32), np.array([0.002, -0.004, 0.001, 1, 0.01, -0.02, 0.005], np.float32)):
        qt = qt / np.float32(np.linalg.norm(qt[:4])) if False else qt
        qt[:4] /= np.linalg.norm(qt[:4])
        want = refl.se3_step(qt, cam, img0, img1, dpt0, grad1, 0.1)
        got = oracle.se3_step(qt, cam, img0, img1, dpt0, grad1, 0.1, accum_f64=True)
        _close(got, want, 320, 240)
        ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
        r_res, r_inl = refl.sfm_error(ident, qt, cam, img0, img1, dpt0, grad1, 0.1)
        o_res, o_inl = oracle.sfm_error(ident, qt, cam, img0, img1, dpt0, 0.1)
        assert abs(r_inl - o_inl) <= 1 and abs(r_res - o_res) <= 2e-5 * r_res


def test_image_alignment_kat_through_the_reference_code(refl, oracle):
    """ut_se3aligner.cpp:173-211 ImageAlignmentTest with the reference's own LucasKanadeSE3 + SE3SolveAndUpdate: 40 iterations from
    identity reach residual / inliers <= 1e-3 -- and the oracle-driven loop lands on the same pose."""
    img0, img1, dpt0 = load_fixture()
    cam = scenenet_cam(320, 240)
    grad1 = oracle.sobel(img1)
    qt_r = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    qt_o = qt_r.copy()
    for _ in range(40):
        r = refl.se3_step(qt_r, cam, img0, img1, dpt0, grad1, 0.1)
        qt_r = refl.se3_solve_and_update(r.JtJ, r.Jtr, qt_r)
        o = oracle.se3_step(qt_o, cam, img0, img1, dpt0, grad1, 0.1, accum_f64=True)
        qt_o = oracle.se3_solve_update(o.JtJ.astype(np.float32), o.Jtr.astype(np.float32), qt_o)
    assert r.residual / r.inliers <= 1e-3 and r.inliers > 0.9 * 320 * 240
    assert np.abs(qt_r - qt_o).max() <= 2e-5, (qt_r, qt_o)


@pytest.mark.parametrize("cs", [16, 32, 64])
def test_update_depth_is_bit_identical(oracle, refl, cs):
    n = _pair(80, 60, cs, 41)
    rng = np.random.default_rng(cs)
    code = rng.normal(0, 0.3, cs).astype(np.float32)
    a = refl.update_depth(code, n["prx_orig"], n["prx_jac"], 2.0)
    b = oracle.update_depth(code, n["prx_orig"], n["prx_jac"], 2.0)
    # the reference's Eigen product (1 x CS) * (CS x 1) is a plain left-to-right fp32 dot product; so is the oracle's
    assert np.abs(a - b).max() <= 2e-6 * float(((2.0 + b) ** 2 / 2.0).max())


def test_relative_pose_and_small_math(oracle, refl):
    rng = np.random.default_rng(3)
    for _ in range(5):
        a = np.concatenate([rng.normal(size=4), rng.normal(size=3)]).astype(np.float32); a[:4] /= np.linalg.norm(a[:4])
        b = np.concatenate([rng.normal(size=4), rng.normal(size=3)]).astype(np.float32); b[:4] /= np.linalg.norm(b[:4])
        qt, Ja, Jb = refl.relative_pose(a, b)
        R, t, Oa, Ob = oracle.relative_pose(a, b, dtype=np.float64)
        Rr = oracle.quat_to_R(qt.astype(np.float64))
        assert np.abs(Rr - R).max() <= 2e-6 and np.abs(qt[4:] - t).max() <= 2e-6 * max(1.0, float(np.abs(t).max()))
        assert np.abs(Ja - Oa).max() <= 5e-6 * max(1.0, float(np.abs(Oa).max())) and np.abs(Jb - Ob).max() <= 5e-6
    for x in (0.0, 0.05, -0.0999, 0.1, 0.1001, -0.7, 3.0):
        assert abs(refl.huber_weight(x, 0.1) - oracle.huber_weight(x, 0.1)) <= 1e-7
    assert refl.huber_weight(0.05, 0.1) == 1.0 and abs(refl.huber_weight(0.4, 0.1) - np.sqrt(0.1 * (0.8 - 0.1)) / 0.4) < 1e-6   # abs() is the float overload
    for d in (0.5, 2.0, 7.5):
        assert abs(refl.depth_jacobian_prx(d, 2.0) - oracle.depth_jacobian_prx(d, 2.0, dtype=np.float32)) <= 1e-5 * abs(oracle.depth_jacobian_prx(d, 2.0))


def test_camera_pyramid_matches_the_reference_code():
    """CameraPyramid (camera_pyramid.h:35-48: every level is the ORIGINAL camera resized to the integer-halved size of the level above)
    against synth.camera_pyramid (level by level): identical for even sizes (the ratios are powers of two), within one float ulp of the
    intrinsics for sizes that halve with a remainder."""
    from deepfactors_amd import synth
    for cam, levels in ((synth.scenenet_cam(640, 480), 4), (synth.scenenet_cam(1280, 960), 4), (synth.scenenet_cam(320, 240), 3)):
        want = ref.camera_pyramid(cam, levels)
        got = np.stack(synth.camera_pyramid(cam, levels))
        assert np.array_equal(got, want), (got, want)
    odd = np.array([100.0, 98.5, 50.5, 37.25, 101, 75], np.float32)
    want = ref.camera_pyramid(odd, 3)
    got = np.stack(synth.camera_pyramid(odd, 3))
    assert np.array_equal(got[:, 4:], want[:, 4:]) and list(want[:, 4]) == [101, 50, 25] and list(want[:, 5]) == [75, 37, 18]
    assert np.allclose(got[:, :4], want[:, :4], rtol=2.5e-7, atol=0)


# ---- SparseGeometricFactor::linearize and the DepthAligner kernel, against the reference's own code -------------------
def _two_keyframes(w, h, seed):
    from deepfactors_amd import synth
    p0 = synth.to_numpy(synth.make_pair(w, h, 32, seed=seed, device="cpu"))
    p1 = synth.to_numpy(synth.make_pair(w, h, 32, seed=seed + 1, device="cpu", motion_scale=0.7))
    return p0, p1


@pytest.mark.parametrize("w,h,npts,seed", [(160, 120, 500, 31), (320, 240, 700, 33), (96, 64, 300, 35)])
def test_sparse_geometric_oracle_equals_reference_code(oracle, refl, w, h, npts, seed):
    """core/gtsam/sparse_geometric_factor.cpp:147-275, compiled unmodified (oracle/ref_harness_f3.cpp): every row [pose0 | pose1 | code0 | code1 |
    err] of the JacobianFactor, incl. the all-zero rows of points without a correspondence, for poses that push a share of the points out
    of view."""
    p0, p1 = _two_keyframes(w, h, seed)
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.integers(0, w, npts), rng.integers(0, h, npts)], axis=1).astype(np.int32)
    dgrad1 = oracle.sobel(p1["dpt0"])
    pose1 = p0["pose1"].copy(); pose1[4] += 0.08; pose1[5] -= 0.03
    for huber in (0.1, 0.02):
        want = refl.sparse_geometric(p0["pose0"], pose1, p0["code"], p1["code"], p0["cam"], pts, p0["prx_orig"], p0["prx_jac"], p1["prx_orig"], p1["prx_jac"],
                                     dgrad1, huber)
        got = oracle.sparse_geometric(p0["pose0"], pose1, p0["code"], p1["code"], p0["cam"], pts, p0["prx_orig"], p0["prx_jac"], p1["prx_orig"], p1["prx_jac"],
                                      dgrad1, huber, avg_dpt=2.0)
        zero_w, zero_g = ~want.any(axis=1), ~got.astype(np.float64).any(axis=1)
        assert 0 < zero_w.sum() < npts, "the test must exercise both branches"
        # a point within an ulp of the view border may flip (quaternion vs matrix rotation, see the module docstring)
        assert int((zero_w != zero_g).sum()) <= 1
        same = zero_w == zero_g
        scale = np.abs(want).max(axis=0) + 1e-12
        assert (np.abs(got[same] - want[same]) / scale).max() <= 2e-5


@pytest.mark.parametrize("w,h,seed", [(160, 120, 41), (100, 77, 43)])
def test_depth_aligner_oracle_equals_reference_code(oracle, refl, w, h, seed):
    """cuda/cu_depthaligner.cpp:32-72 (the kernel template, cut out at build time and run over all pixels in order)."""
    from deepfactors_amd import synth
    n = synth.to_numpy(synth.make_pair(w, h, 32, seed=seed, device="cpu"))
    rng = np.random.default_rng(seed)
    code = (n["code"] + rng.normal(0, 0.05, 32)).astype(np.float32)
    tgt = (n["dpt0"] + rng.normal(0, 0.02, n["dpt0"].shape)).astype(np.float32)
    want = refl.depth_aligner_step(code, tgt, n["prx_orig"], n["prx_jac"])
    got = oracle.depth_aligner_step(code, tgt, n["prx_orig"], n["prx_jac"], 2.0, accum_f64=True)
    assert got.inliers == want.inliers == w * h
    # the reference accumulates w*h terms in float in pixel order: 1e-4 of the scale is its own rounding
    assert_item_close(got, want, w, h, rel=2e-4, what="depth aligner, oracle vs reference kernel body")


@pytest.mark.parametrize("w,h,seed", [(160, 120, 51), (101, 67, 52), (64, 48, 53)])
def test_image_proc_oracle_equals_reference_kernel_bodies(oracle, refl, w, h, seed):
    """cuda/cu_image_proc.cpp:57-92 (Sobel), :134-164 (5 x 5 blur + decimate), :190-206 (squared error): the kernel bodies, cut out at build
    time and run over every pixel (oracle/ref_harness_f1.cpp).  Sobel and the blur are exact: same taps, same order, same division."""
    n = _pair(w, h, 16, seed, with_decoder=False)
    assert np.array_equal(oracle.sobel(n["img0"]), refl.sobel_gradients(n["img0"]))
    assert np.array_equal(oracle.blur_down(n["img0"]), refl.gaussian_blur_down(n["img0"]))
    want = refl.squared_error(n["img0"], n["img1"])
    got32 = oracle.squared_error(n["img0"], n["img1"], accum_f64=False)
    got64 = oracle.squared_error(n["img0"], n["img1"], accum_f64=True)
    assert got32 == want                                  # the float sum in pixel order is the reference's
    assert abs(got64 - want) <= 2e-5 * want               # and the fp64 oracle is within the float sum's own rounding


@pytest.mark.parametrize("w,h,seed", [(160, 120, 61), (101, 67, 62)])
def test_se3_warp_oracle_equals_reference_kernel_body(oracle, refl, w, h, seed):
    """cuda/cu_se3aligner.cpp:61-113: rendered image, SIGNED residual sum, inlier count -- identity and a real relative pose."""
    from deepfactors_amd import synth
    n = _pair(w, h, 16, seed, with_decoder=False)
    for qt in (synth.IDENTITY, n["pose10_true"]):
        img2_w, r_w, n_w = refl.se3_warp(qt, n["cam"], n["img0"], n["img1"], n["dpt0"])
        img2_g, r_g, n_g = oracle.se3_warp(qt, n["cam"], n["img0"], n["img1"], n["dpt0"], accum_f64=False)
        # Sophus rotates through the quaternion in float, the oracle through R rounded once: coordinates differ by ~1 ulp
        assert abs(n_g - n_w) <= max(1, int(1e-5 * w * h))
        diff = np.abs(img2_g - img2_w)
        assert int((diff > 1e-5).sum()) <= max(1, int(1e-5 * w * h)) + abs(n_g - n_w)
        assert abs(r_g - r_w) <= 2e-5 * max(1.0, float(np.abs(n["img0"]).sum()) * 1e-2)
