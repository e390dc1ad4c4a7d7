"""The HIP path against the real-image SfM known-answer fixture: the reference's own test images (data/testimg/0.jpg -> 25.jpg, 0.png depth
with its zero pixels) at the poses / huber_delta of ut_sfmaligner.cpp:254-268,:69, compared with the OUTPUTS OF THE REFERENCE'S OWN CODE
(oracle/_ref, committed in tests/golden/sfm_fixture_0_25.npz; tests/sfm_fixture.py describes inputs and cases).

  * FullJacobianCompareWithCpu (ut_sfmaligner.cpp:235-327): RunStep vs the reference's host loop over DenseSfm, per entry (tests/helpers.py),
    both MFMA modes, single and batch -- and the reference's own criterion (inliers equal, |dJtJ| <= 1e-1) on top;
  * degenerate depths: depth 0, < 0, +-inf, NaN, 1e-30, 65.535 m, and a depth map DECODED ON THE GPU through a zero crossing of the proximity
    (UpdateDepth -> prx <= 0): the inlier SET of RunStep (valid0), EvaluateError, SE3 RunStep, Warp (rendered mask) and of the device
    tracker equals the reference's pixel for pixel;
  * FullJacobianFiniteDiff (ut_sfmaligner.cpp:329-487): Jtr of the GPU against finite differences of the GPU's OWN residual, per pose and
    per code entry, with the reference's step sizes and tolerances (:397-399, :418, :431, :474, :483) and a tighter central-difference form."""
import numpy as np
import pytest
import torch

import sfm_fixture as fx
from helpers import assert_item_close, block_errors, format_block_errors

pytestmark = pytest.mark.gpu

REL = {"raw": 1e-4, "mixed": 1e-4, "decoded": 1e-3}     # 'decoded': see tests/test_golden_sfm_fixture.py
MODES = ("f32", "bf16x3")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _ctx(dfx, mode):
    from deepfactors_amd import _lib
    ctx = dfx.Context(0)
    ctx.set_mfma_mode(_lib.DFX_MFMA_BF16X3 if mode == "bf16x3" else _lib.DFX_MFMA_F32_CHAIN)
    return ctx


@pytest.fixture(scope="module")
def scene(dfx, oracle):
    inp, z = fx.load()
    g = {k: _t(inp[k]) for k in ("img0", "img1", "prx_orig", "prx_jac")}
    g["grad1"] = torch.empty((fx.H, fx.W, 2), dtype=torch.float32, device="cuda")
    dfx.SobelGradients(g["img1"], g["grad1"])

    def gpu_update_depth(code, prx_orig, prx_jac, avg_dpt):     # the decoder under test, shaped like the oracle's for fx.depth_variant
        out = torch.empty((fx.H, fx.W), dtype=torch.float32, device="cuda")
        dfx.UpdateDepth(np.asarray(code, np.float32), g["prx_orig"], g["prx_jac"], avg_dpt, out)
        torch.cuda.synchronize()
        return out.cpu().numpy()

    # the depth maps the reference outputs of the fixture were computed on (the oracle's decoder is bit-identical to the reference's,
    # tests/test_oracle_vs_ref.py), plus the GPU's own decode of the zero-crossing code
    dpt = {v: fx.depth_variant(inp, v, oracle.update_depth) for v in fx.DEPTH_VARIANTS}
    dpt["decoded_on_gpu"] = fx.depth_variant(inp, "decoded", gpu_update_depth)
    return inp, z, g, dpt


def test_gpu_decoder_through_a_zero_of_the_proximity(dfx, scene):
    """UpdateDepth driven to prx <= 0 (kernel_update_depth, cu_image_proc.cpp:248-264; dpt = a / prx - a, warping.h:30-35): negative depths
    where prx < 0, enormous ones where prx -> 0+.  a / prx - a amplifies the summation-order difference of the 32-term dot product without
    bound near the zero, so the decoder is compared where it is linear -- in the proximity domain prx = a / (dpt + a) -- and through what the
    step makes of its output: the inlier set of RunStep on the GPU-decoded depth is the reference's up to the pixels whose proximity
    changed sign."""
    inp, z, g, dpt = scene
    want, got = dpt["decoded"], dpt["decoded_on_gpu"]
    assert (want < 0).mean() > 0.10 and (want > 100).sum() > 500
    a = np.float32(fx.AVG_DPT)
    assert np.array_equal(np.isfinite(got), np.isfinite(want))
    assert np.abs(a / (got + a) - a / (want + a)).max() <= 2e-6      # |prx| reaches 4 with this code: a few ulp(4) of summation-order difference
    sign_flips = int(((got < 0) != (want < 0)).sum())
    assert sign_flips <= 2
    rot, trs, huber = fx.CASES["fwd"]
    al = dfx.SfmAligner(dfx.SfmAlignerParams(dfx.DenseSfmParams(huber_delta=huber, avg_dpt=fx.AVG_DPT)), code_size=fx.CS)
    valid0 = torch.zeros((fx.H, fx.W), dtype=torch.float32, device="cuda")
    al.RunStep(fx.IDENTITY, fx.pose_inverse_of(rot, trs), inp["code_neg"], inp["cam"], g["img0"], g["img1"], _t(got), None, valid0, g["prx_jac"], g["grad1"])
    assert int(((valid0.cpu().numpy() != 0) != fx.expected(z, "fwd", "decoded")["valid0"]).sum()) <= sign_flips + 2


@pytest.mark.parametrize("variant", fx.DEPTH_VARIANTS)
@pytest.mark.parametrize("case", list(fx.CASES))
def test_hip_reproduces_the_reference_outputs(dfx, scene, case, variant):
    inp, z, g, dpt = scene
    rot, trs, huber = fx.CASES[case]
    pose0, pose1 = fx.IDENTITY, fx.pose_inverse_of(rot, trs)
    want = fx.expected(z, case, variant)
    d = _t(dpt[variant])
    flips_ok = 0                                     # inlier sets are compared EXACTLY
    for mode in MODES:
        ctx = _ctx(dfx, mode)
        al = dfx.SfmAligner(dfx.SfmAlignerParams(dfx.DenseSfmParams(huber_delta=huber, avg_dpt=fx.AVG_DPT)), code_size=fx.CS, ctx=ctx)
        valid0 = torch.zeros((fx.H, fx.W), dtype=torch.float32, device="cuda")
        got = al.RunStep(pose0, pose1, inp["code"], inp["cam"], g["img0"], g["img1"], d, None, valid0, g["prx_jac"], g["grad1"])
        vm = valid0.cpu().numpy() != 0
        ndiff = int((vm != want["valid0"]).sum())
        assert ndiff <= flips_ok, f"{case}/{variant}/{mode}: RunStep inlier set differs from the reference's on {ndiff} px"
        assert abs(got.inliers - want["sfm"].inliers) <= flips_ok
        errs = assert_item_close(got, want["sfm"], fx.W, fx.H, rel=REL[variant], what=f"RunStep {case}/{variant}/{mode}")
        if case == "ut" and variant == "raw":      # the reference's own acceptance criterion, ut_sfmaligner.cpp:320-326
            assert got.inliers == want["sfm"].inliers != 0
            assert np.abs(np.asarray(got.JtJ, np.float64) - want["sfm"].JtJ).max() <= 1e-1, format_block_errors(errs)
        # batch forms of the same kernels
        arr = al.make_pairs([dict(pose0=pose0, pose1=pose1, cam=inp["cam"], img0=g["img0"], img1=g["img1"], dpt0=d, prx0_jac=g["prx_jac"], grad1=g["grad1"])] * 2)
        for it in al.RunStepBatch(arr):
            assert_item_close(it, want["sfm"], fx.W, fx.H, rel=REL[variant], what=f"RunStepBatch {case}/{variant}/{mode}")
        e = al.EvaluateError(pose0, pose1, inp["cam"], g["img0"], g["img1"], d, None, g["grad1"])
        assert abs(e.inliers - want["err"][1]) <= flips_ok and abs(e.residual - want["err"][0]) <= 1e-4 * want["err"][0]
        for eb in al.EvaluateErrorBatch(arr):
            assert abs(eb.inliers - want["err"][1]) <= flips_ok and abs(eb.residual - want["err"][0]) <= 1e-4 * want["err"][0]
    rel = fx.rel_pose_qt(pose0, pose1)
    se3 = dfx.SE3Aligner()
    se3.SetHuberDelta(huber)
    k = se3.RunStep(rel, inp["cam"], g["img0"], g["img1"], d, g["grad1"])
    assert abs(k.inliers - want["se3"].inliers) <= flips_ok
    assert_item_close(k, want["se3"], fx.W, fx.H, what=f"SE3 RunStep {case}/{variant}")
    sarr = se3.make_pairs([dict(se3=rel, cam=inp["cam"], img0=g["img0"], img1=g["img1"], dpt0=d, grad1=g["grad1"])] * 2)
    for kb in se3.RunStepBatch(sarr):
        assert_item_close(kb, want["se3"], fx.W, fx.H, what=f"SE3 RunStepBatch {case}/{variant}")
    img2 = torch.full((fx.H, fx.W), -1.0, dtype=torch.float32, device="cuda")
    w = se3.Warp(rel, inp["cam"], g["img0"], g["img1"], d, img2)
    wm = img2.cpu().numpy() != 0
    assert int((wm != want["warp_mask"]).sum()) <= flips_ok and abs(w.inliers - want["warp"][1]) <= flips_ok
    assert abs(w.residual - want["warp"][0]) <= 1e-4 * max(abs(want["warp"][0]), np.sqrt(w.inliers))


@pytest.mark.parametrize("variant", ["mixed", "decoded"])
def test_device_tracker_on_degenerate_depths(dfx, oracle, scene, variant):
    """dfx_track_frame (camera_tracker.cpp:42-71 on the device) with a keyframe depth full of 0 / < 0 / inf / NaN: the same pose, inlier
    fraction and error as the host loop over the oracle (whose per-step inlier sets the fixture test above pins to the reference's)."""
    inp, z, g, dpt = scene
    d = dpt[variant].copy()
    # the tracker starts at the identity, where a depth of 1e-30 is a VALID correspondence (q.z = 1e-30 > 0, warping.h:221-224) whose
    # projection Jacobian fx / q.z = 3e32 squares to +inf in JtJ: the reference's own 6x6 system is singular on such a keyframe.  That
    # rectangle becomes depth 0 here (q = t = 0 at the identity: no correspondence); 0, < 0, +-inf, NaN and 65.535 m stay.
    d[(d > 0) & (d < 1e-20)] = 0.0
    huber, iters = 0.1, 12
    qt = fx.IDENTITY.copy()
    g1 = oracle.sobel(inp["img1"])
    for _ in range(iters):
        r = oracle.se3_step(qt, inp["cam"], inp["img0"], inp["img1"], d, g1, huber)
        qt = oracle.se3_solve_update(r.JtJ, r.Jtr, qt)
    last = oracle.se3_step(qt, inp["cam"], inp["img0"], inp["img1"], d, g1, huber)
    tr = dfx.CameraTracker([inp["cam"]], dfx.TrackerConfig(1, (iters,), huber))
    tr.SetKeyframe([g["img0"]], [_t(d)])
    pose = tr.TrackFrame([g["img1"]], [g["grad1"]])
    assert tr.last_result_.solver_failures == 0 and np.isfinite(pose).all()
    assert np.linalg.norm(pose - qt) < 2e-4, (pose, qt)
    # GetInliers / GetError are those of the LAST step taken (camera_tracker.cpp:64-69), i.e. at the pose before the last update
    assert abs(tr.GetInliers() * fx.W * fx.H - r.inliers) <= 2
    assert abs(tr.GetError() - r.residual / r.inliers) <= 1e-3 * r.residual / r.inliers
    assert last.inliers > 0.5 * fx.W * fx.H


def test_full_jacobian_finite_diff_on_the_gpu(dfx, oracle, scene):
    """ut_sfmaligner.cpp:329-487 through the HIP path: Jtr_i against finite differences of the residual, where BOTH sides are GPU RunStep
    outputs (code perturbations go through the GPU UpdateDepth, :467-468) -- per pose entry and per CODE entry.

    The evaluation window is the interior 220 x 160 px (depth NaN elsewhere = no correspondence, warping.h:221-224): every correspondence
    stays well inside img1, so no boundary pixel enters or leaves under the perturbations and the residual is differentiable (on the whole
    frame the unaligned 0 -> 25 pair changes its inlier set by ~100 px per 2 mm, each worth r^2 ~ 0.1: that, not the Jacobian, is what a
    finite difference would measure).

    (1) The reference's criterion: forward difference, pose eps 1e-5 tol 2e1 verbatim, code eps 1e-3 tol 1.5e-2 (:397-399, :418, :431, :474,
        :483) -- the code tolerance widened by what a FLOAT residual of this size cannot resolve, ulp(residual) / eps for the difference of
        two rounded sums (0.03 here: the reference's 1.5e-2 presumes a residual of a few units), and by the stencil error of (2).
    (2) Central difference with fp32-sized steps against the entry's own Cauchy-Schwarz scale sqrt(JtJ_ii * residual): 1e-2.  Jtr uses the
        Sobel gradient of img1, the residual its bilinear interpolant, so they agree to the stencil error (3e-3 on the 25 x 25-blurred images,
        measured with the fp64 oracle), not to rounding."""
    inp, z, g, dpt = scene
    rot, trs, huber = fx.CASES["ut01"]                     # huber_delta 0.5 like the reference test: almost every residual is in the quadratic zone
    pose0, pose1 = fx.IDENTITY, fx.pose_inverse_of(rot, trs)
    al = dfx.SfmAligner(dfx.SfmAlignerParams(dfx.DenseSfmParams(huber_delta=huber, avg_dpt=fx.AVG_DPT)), code_size=fx.CS)
    dbuf = torch.empty((fx.H, fx.W), dtype=torch.float32, device="cuda")
    window = torch.full((fx.H, fx.W), float("nan"), dtype=torch.float32, device="cuda")
    window[40:200, 50:270] = 0.0

    def step(p0, p1, code):
        dfx.UpdateDepth(np.asarray(code, np.float32), g["prx_orig"], g["prx_jac"], fx.AVG_DPT, dbuf)
        dbuf.add_(window)
        return al.RunStep(p0, p1, code, inp["cam"], g["img0"], g["img1"], dbuf, None, None, g["prx_jac"], g["grad1"])

    code0 = np.zeros(fx.CS, np.float32)
    base = step(pose0, pose1, code0)
    assert base.inliers == 220 * 160
    jtr = np.asarray(base.Jtr, np.float64)
    scale = np.sqrt(np.diag(base.toDenseMatrix()).astype(np.float64) * base.residual)     # Cauchy-Schwarz bound of |Jtr_i|
    ulp = float(np.spacing(np.float32(base.residual)))

    def perturbed(i, eps):
        if i < 6:
            return oracle.perturb_pose(pose0, i, eps, np.float64).astype(np.float32), pose1, code0
        if i < 12:
            return pose0, oracle.perturb_pose(pose1, i - 6, eps, np.float64).astype(np.float32), code0
        c = code0.copy(); c[i - 12] += eps
        return pose0, pose1, c

    for i in range(12 + fx.CS):
        # (1) the reference's own test
        eps, tol = (1e-5, 2e1) if i < 12 else (1e-3, 1.5e-2 + ulp / 1e-3 + 3e-3 * scale[i])
        r = step(*perturbed(i, eps))
        assert r.inliers == base.inliers
        fd = 0.5 * (r.residual - base.residual) / eps
        assert abs(fd - jtr[i]) <= tol, f"reference criterion, parameter {i}: finite difference {fd} vs Jtr {jtr[i]} (tol {tol})"
        # (2) central difference
        eps = 1e-3 if i < 12 else 1e-2
        rp, rm = step(*perturbed(i, eps)), step(*perturbed(i, -eps))
        assert rp.inliers == rm.inliers == base.inliers
        fc = 0.25 * (rp.residual - rm.residual) / eps
        assert abs(fc - jtr[i]) <= 1e-2 * scale[i], f"parameter {i}: central difference {fc} vs Jtr {jtr[i]} (scale {scale[i]:.3g})"
