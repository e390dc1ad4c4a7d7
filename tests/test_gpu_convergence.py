"""north_star: "converge to the same pose/code estimates on fixed inputs".  A joint Gauss-Newton over (pose1, code0) --
the unknowns PhotometricFactor exposes per keyframe pair (photometric_factor.cpp:105-161) with pose0 held as the gauge --
driven once by the HIP path (UpdateDepth + SfmAligner::RunStep through the C ABI) and once by the CPU oracle, same
solver, same retraction (left-multiplicative rotation, R3 x SO3 split)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _quat_to_R(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _retract(synth, qt, d):
    R = synth.so3_exp(np.asarray(d[3:], np.float64)) @ _quat_to_R(qt[:4])
    return np.concatenate([synth.R_to_quat(R), np.asarray(qt[4:], np.float64) + d[:3]]).astype(np.float32)


def _gauss_newton(synth, step, update_depth, pose0, pose1, code, cs, iters, lm=1e-4):
    hist = []
    for _ in range(iters):
        r = step(pose0, pose1, update_depth(code))
        H = np.asarray(r.dense() if hasattr(r, "dense") else r.toDenseMatrix(), np.float64)
        g = np.asarray(r.Jtr, np.float64)
        idx = list(range(6, 12 + cs))
        A = H[np.ix_(idx, idx)]
        A = A + lm * np.diag(np.diag(A))
        d = -np.linalg.solve(A, g[idx])
        pose1 = _retract(synth, pose1, d[:6])
        code = (code + d[6:]).astype(np.float32)
        hist.append((float(r.residual), int(r.inliers)))
    return pose1, code, hist


@pytest.mark.parametrize("w,h,cs,seed", [(160, 120, 16, 21), (320, 240, 32, 22), (640, 480, 32, 23)])
def test_sfm_gauss_newton_converges_like_the_oracle(dfx, oracle, w, h, cs, seed):
    from deepfactors_amd import synth
    p = synth.make_pair(w, h, cs, seed=seed, device="cpu")
    n, g = synth.to_numpy(p), synth.to_device(p, "cuda")
    pose1_0 = _retract(synth, n["pose1"], np.array([0.01, -0.008, 0.005, 0.004, -0.003, 0.002]))
    code_0 = (n["code"] * 0.5).astype(np.float32)
    iters = 10

    al = dfx.SfmAligner(code_size=cs)
    dpt_dev = torch.empty_like(g["img0"])

    def upd_gpu(code):
        dfx.UpdateDepth(code, g["prx_orig"], g["prx_jac"], 2.0, dpt_dev, al.ctx)
        return dpt_dev

    def step_gpu(p0, p1, dpt):
        return al.RunStep(p0, p1, None, n["cam"], g["img0"], g["img1"], dpt, None, None, g["prx_jac"], g["grad1"])

    def upd_cpu(code):
        return oracle.update_depth(code, n["prx_orig"], n["prx_jac"], 2.0)

    def step_cpu(p0, p1, dpt):
        return oracle.sfm_step(p0, p1, n["cam"], n["img0"], n["img1"], dpt, n["prx_jac"], n["grad1"])

    pg, cg, hg = _gauss_newton(synth, step_gpu, upd_gpu, n["pose0"], pose1_0, code_0, cs, iters)
    pc, cc, hc = _gauss_newton(synth, step_cpu, upd_cpu, n["pose0"], pose1_0, code_0, cs, iters)

    # both converge (the residual drops by > 10x and settles) ...
    assert hg[-1][0] < 0.1 * hg[0][0] and abs(hg[-1][0] - hg[-2][0]) < 1e-3 * hg[-1][0]
    # ... to the same estimates, at SURVEY 8c's stated bar: pose within 1e-4 (quaternion / metres) and code within 1e-4 of the
    # oracle after the same schedule (measured on MI355X: pose 3e-7, code 6e-6 with normal equations of condition 5e6-7e6 -- the
    # same distance the oracle's own fp32-accumulating mode keeps from its fp64 mode), residual within 1e-4 relative
    assert np.abs(pg - pc).max() < 1e-4, (pg, pc)
    assert np.abs(cg - cc).max() < 1e-4, np.abs(cg - cc).max()
    assert abs(hg[-1][0] - hc[-1][0]) < 1e-4 * hc[-1][0] and abs(hg[-1][1] - hc[-1][1]) <= max(2, 1e-5 * w * h)
    # ... which are the generating ones up to the model error of the synthetic pair (Sobel vs bilinear derivative)
    assert np.abs(pg - n["pose1"]).max() < 2e-3
    assert np.abs(cg - n["code"]).max() < 0.25 * np.abs(code_0 - n["code"]).max()


def test_sfm_gauss_newton_on_the_reference_images_converges_like_the_oracle(dfx, oracle):
    """The same joint Gauss-Newton on REAL data: the reference's SfM test pair 0.jpg -> 25.jpg with 0.png depth (its zero pixels kept) and the seeded code Jacobian of
    tests/sfm_fixture.py, huber_delta 0.1, from the scaled test pose of ut_sfmaligner.cpp:254-268.  Unlike the synthetic pairs the images are not consistent with ANY pose
    (25 frames apart, depth of frame 0 only), so the residual stays large (Huber active throughout), the inlier set changes from iteration to iteration and the code
    moves by O(1): what is asserted is that the HIP-driven and the oracle-driven loops follow the SAME trajectory -- pose and code within 1e-4 after 6 iterations,
    residual within 1e-4 relative, inlier counts within a handful of border pixels -- not that they converge to a truth.  The damping is heavy (lambda = 10 x diag) on
    purpose: with lambda <= 1 the trajectory on this inconsistent pair is itself unstable (the oracle accumulating in float instead of double ends 4e-3 away in pose and
    5e-2 in code after 6 iterations -- inlier flips feed back); at lambda = 10 that same pair of oracle runs agrees to 5e-7 / 6e-5, so 1e-4 measures the evaluation."""
    import sfm_fixture as fx
    from deepfactors_amd import synth
    inp, _ = fx.load()
    rot, trs, huber = fx.CASES["ut01_h01"]
    pose0, pose1_0 = fx.IDENTITY, fx.pose_inverse_of(rot, trs)
    cs, iters = fx.CS, 6
    g = {k: torch.from_numpy(np.ascontiguousarray(inp[k])).cuda() for k in ("img0", "img1", "prx_orig", "prx_jac")}
    g["grad1"] = torch.empty((fx.H, fx.W, 2), dtype=torch.float32, device="cuda")
    dfx.SobelGradients(g["img1"], g["grad1"])
    grad1 = oracle.sobel(inp["img1"])
    assert np.array_equal(g["grad1"].cpu().numpy(), grad1)
    al = dfx.SfmAligner(dfx.SfmAlignerParams(dfx.DenseSfmParams(huber_delta=huber, avg_dpt=fx.AVG_DPT)), code_size=cs)
    dpt_dev = torch.empty((fx.H, fx.W), dtype=torch.float32, device="cuda")

    def upd_gpu(code):
        dfx.UpdateDepth(code, g["prx_orig"], g["prx_jac"], fx.AVG_DPT, dpt_dev, al.ctx)
        return dpt_dev

    def step_gpu(p0, p1, dpt):
        return al.RunStep(p0, p1, None, inp["cam"], g["img0"], g["img1"], dpt, None, None, g["prx_jac"], g["grad1"])

    def upd_cpu(code):
        return oracle.update_depth(code, inp["prx_orig"], inp["prx_jac"], fx.AVG_DPT)

    def step_cpu(p0, p1, dpt):
        return oracle.sfm_step(p0, p1, inp["cam"], inp["img0"], inp["img1"], dpt, inp["prx_jac"], grad1, huber_delta=huber, avg_dpt=fx.AVG_DPT)

    code_0 = np.zeros(cs, np.float32)
    pg, cg, hg = _gauss_newton(synth, step_gpu, upd_gpu, pose0, pose1_0, code_0, cs, iters, lm=10.0)
    pc, cc, hc = _gauss_newton(synth, step_cpu, upd_cpu, pose0, pose1_0, code_0, cs, iters, lm=10.0)
    assert hg[-1][0] < 0.6 * hg[0][0]                             # the loop does reduce the (Huber) cost: 640 -> 312
    assert np.abs(pg - pc).max() < 1e-4, (pg, pc)
    assert np.abs(cg - cc).max() < 1e-4, np.abs(cg - cc).max()
    for (rg, ig), (rc, ic) in zip(hg, hc):
        assert abs(rg - rc) <= 1e-4 * rc and abs(ig - ic) <= max(2, int(1e-4 * fx.W * fx.H)), (hg, hc)
    assert np.abs(cg).max() > 1e-3 and np.abs(pg - pose1_0).max() > 1e-4   # both the code and the pose moved
