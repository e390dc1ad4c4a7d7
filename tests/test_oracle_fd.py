"""Finite-difference property tests of the oracle's analytic Jacobians -- the reference's own ut_warping.cpp:72-380,
ut_pinhole_camera.cpp:50-134 and ut_sfmaligner.cpp:329-487 re-expressed with fixed seeds (the reference seeds its
RandomMachine from std::random_device, random_machine.h:54-55).  They pin the conventions: left-multiplicative rotation
updates, R3 x SO3 tangent split, [tx ty tz wx wy wz] order, -hat(R p) without translation."""
import numpy as np
import pytest


def rand_pose(rng, oracle):
    w = rng.uniform(-0.6, 0.6, 3)
    t = rng.uniform(-1, 1, 3)
    R = oracle.so3_exp(w)
    from deepfactors_amd import synth
    return np.concatenate([synth.R_to_quat(R), t])


CAM = np.array([277.128, 289.706, 160.0, 120.0, 320.0, 240.0])


@pytest.mark.parametrize("seed", range(5))
def test_relative_pose_jacobians(oracle, seed):
    """ut_warping.cpp RelativePose: d(T_ab)/d(pose_a), d(T_ab)/d(pose_b) vs finite differences (eps 1e-6, tol 1e-5)."""
    from deepfactors_amd import synth
    rng = np.random.default_rng(100 + seed)
    a, b = rand_pose(rng, oracle), rand_pose(rng, oracle)
    R, t, Ja, Jb = oracle.relative_pose(a, b)

    def local(Rn, tn):   # gtsam_traits.h:61-68 Local(): (dt, log(Rn R^T))
        dR = Rn @ R.T
        ang = np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))
        wv = np.array([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]]) / 2
        if ang > 1e-12:
            wv = wv * ang / np.sin(ang)
        return np.concatenate([tn - t, wv])

    eps = 1e-6
    for which, J in (("a", Ja), ("b", Jb)):
        for i in range(6):
            pa = oracle.perturb_pose(a, i, eps) if which == "a" else a
            pb = oracle.perturb_pose(b, i, eps) if which == "b" else b
            Rn, tn, _, _ = oracle.relative_pose(pa, pb)
            fd = local(Rn, tn) / eps
            assert np.abs(fd - J[:, i]).max() < 1e-5, (which, i)


@pytest.mark.parametrize("seed", range(5))
def test_correspondence_jacobians(oracle, seed):
    """FindCorrespondenceJacobianPose / Depth / Prx vs finite differences (ut_warping.cpp:243-380; ut_pinhole_camera.cpp)."""
    rng = np.random.default_rng(200 + seed)
    qt = rand_pose(rng, oracle) * np.array([0.2, 0.2, 0.2, 1, 0.1, 0.1, 0.1])
    qt[:4] /= np.linalg.norm(qt[:4])
    x, y, d = int(rng.integers(40, 280)), int(rng.integers(40, 200)), float(rng.uniform(1.0, 4.0))
    c0 = oracle.correspondence(x, y, d, CAM, qt)
    eps = 1e-6
    for i in range(6):
        c1 = oracle.correspondence(x, y, d, CAM, oracle.perturb_pose(qt, i, eps))
        assert np.abs((c1["pix1"] - c0["pix1"]) / eps - c0["jac_pose"][:, i]).max() < 1e-2
    c1 = oracle.correspondence(x, y, d + eps, CAM, qt)
    assert np.abs((c1["pix1"] - c0["pix1"]) / eps - c0["jac_dpt"]).max() < 1e-3
    # proximity: d = a/prx - a
    a = 2.0
    prx = a / (a + d)
    c1 = oracle.correspondence(x, y, a / (prx + 1e-8) - a, CAM, qt)
    assert np.abs((c1["pix1"] - c0["pix1"]) / 1e-8 - c0["jac_prx"]).max() < 1e-2 * max(1.0, np.abs(c0["jac_prx"]).max())


def test_depth_jacobian_prx_and_huber(oracle):
    a = 2.0
    for d in (0.5, 1.7, 4.0):
        prx = oracle.depth_to_prox(d, a)
        fd = (oracle.prox_to_depth(prx + 1e-7, a) - oracle.prox_to_depth(prx, a)) / 1e-7
        assert abs(fd - oracle.depth_jacobian_prx(d, a)) < 1e-4 * abs(fd)
    # HuberWeight (m_estimators.h:50-56): w^2 r^2 equals the Huber loss 2*rho(r) on both branches
    delta = 0.1
    for r in (0.0, 0.05, 0.1, 0.25, -0.7):
        w = oracle.huber_weight(r, delta, np.float64)
        rho2 = r * r if abs(r) <= delta else delta * (2 * abs(r) - delta)
        assert abs((w * r) ** 2 - rho2) < 1e-12


def test_sfm_jtr_matches_finite_difference_of_residual(oracle):
    """ut_sfmaligner.cpp:329-487 FullJacobianFiniteDiff in double: Jtr_i ~ 0.5 d(residual)/d(param_i).
    Not an identity: Jtr uses the Sobel gradient of img1 while the residual uses its bilinear interpolant, so the two
    agree only up to the stencil error (the reference accepts tol_pose = 2e1 absolute, tol_code = 1.5e-2); a smooth
    texture (320 px wide) keeps that error at the few-percent level."""
    from deepfactors_amd import synth
    p = synth.make_pair(320, 240, 16, seed=11)
    n = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v) for k, v in synth.to_numpy(p).items()}
    pose1 = n["pose1"].copy(); pose1[4] += 0.004; pose1[5] -= 0.003
    code = n["code"].copy()

    def step(p0, p1, c):
        dpt = oracle.update_depth(c, n["prx_orig"], n["prx_jac"], 2.0)
        return oracle.sfm_step(p0, p1, n["cam"], n["img0"], n["img1"], dpt, n["prx_jac"], n["grad1"], huber_delta=1e3)

    base = step(n["pose0"], pose1, code)
    scale = np.sqrt(np.diag(base.dense()) * base.residual)
    eps = 1e-6
    for i in range(12):
        p0 = oracle.perturb_pose(n["pose0"], i, eps) if i < 6 else n["pose0"]
        p1 = oracle.perturb_pose(pose1, i - 6, eps) if i >= 6 else pose1
        r = step(p0, p1, code)
        if r.inliers != base.inliers:
            continue   # a boundary pixel flipped: the residual is discontinuous there
        fd = 0.5 * (r.residual - base.residual) / eps
        assert abs(fd - base.Jtr[i]) <= 5e-2 * scale[i] + 1e-6, (i, fd, base.Jtr[i])
    for k in (0, 5, 15):
        c = code.copy(); c[k] += 1e-5
        r = step(n["pose0"], pose1, c)
        if r.inliers != base.inliers:
            continue
        fd = 0.5 * (r.residual - base.residual) / 1e-5
        assert abs(fd - base.Jtr[12 + k]) <= 5e-2 * scale[12 + k] + 1e-6, (k, fd, base.Jtr[12 + k])


def test_se3_gauss_newton_recovers_synthetic_motion(oracle):
    """Coarse-to-fine tracker schedule of camera_tracker.cpp:42-71 on the synthetic pair converges to the GT twist."""
    from deepfactors_amd import synth
    n = synth.to_numpy(synth.make_pair(160, 120, 16, seed=12, with_decoder=False))
    qt = synth.IDENTITY.copy()
    for _ in range(15):
        r = oracle.se3_step(qt, n["cam"], n["img0"], n["img1"], n["dpt0"], n["grad1"], 0.1, accum_f64=False)
        qt = oracle.se3_solve_update(r.JtJ, r.Jtr, qt)
    gt = n["pose10_true"]
    assert np.linalg.norm(qt[4:] - gt[4:]) < 2e-3 and np.linalg.norm(qt[:4] - gt[:4]) < 1e-3


def test_pose_local_is_the_inverse_of_the_reference_retraction(oracle):
    """deepfactors_amd.factors.pose_local (gtsam_traits.h:66-72, used by PhotometricFactor's relinearisation cache) inverts the
    perturbation the Jacobians are defined against (gtsam_traits.h:48-58; oracle.perturb_pose): Local(p, Retract(p, d)) = d."""
    from deepfactors_amd.factors import pose_equals, pose_local
    rng = np.random.default_rng(9)
    for _ in range(5):
        p = rand_pose(rng, oracle)
        d = rng.normal(0, 1e-2, 6)
        q = p.copy()
        for i in range(3):          # translations add ...
            q = oracle.perturb_pose(q, i, d[i])
        from deepfactors_amd import synth
        R = oracle.so3_exp(d[3:]) @ synth.quat_to_R(q[:4])   # ... the rotation update multiplies from the left
        q = np.concatenate([synth.R_to_quat(R), q[4:]])
        assert np.abs(pose_local(p, q) - d).max() < 1e-9
        assert not pose_equals(p, q, 1e-6) and pose_equals(p, p, 1e-6)
    # single-coordinate perturbations of the oracle map to unit tangent directions
    p = rand_pose(rng, oracle)
    for i in range(6):
        e = pose_local(p, oracle.perturb_pose(p, i, 1e-4))
        ref = np.zeros(6); ref[i] = 1e-4
        assert np.abs(e - ref).max() < 1e-10
