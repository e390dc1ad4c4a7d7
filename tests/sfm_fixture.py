"""The real-image SfM known-answer fixture (tests/golden/sfm_fixture_0_25.npz): loading, preprocessing and case table, shared by the
generator (tests/golden/make_sfm_fixture.py, build container only) and the tests that consume it (CPU: oracle; GPU: the HIP path).

It re-creates ut_sfmaligner::FullJacobianCompareWithCpu (/root/reference/tests/ut_sfmaligner.cpp:235-327) as far as this tree allows:

  * images data/testimg/0.jpg -> 25.jpg (ut_sfmaligner.cpp:42-43), grayscale / 255, cv::blur 25 x 25 (:96-98);
  * depth data/testimg/0.png (16-bit mm; its 124 ZERO pixels are kept), turned into the decoder's output form prx = a / (a + d)
    (warping.h:37-42) and decoded back with UpdateDepth at code 0 like the test does (:279-283);
  * camera GetSceneNetCam(320, 240) (testing_utils.h:34-40) -- the test's network camera comes from data/nets/scannet256_32.cfg, a
    download that is not in the tree, so the images stay at their native 320 x 240 intrinsics;
  * prx_jac: the network is a download too, so the Jacobian is a SEEDED smooth field: a 16 x 21 x 32 float32 grid stored in the fixture,
    upsampled bilinearly with float32 ufuncs only (bit-deterministic on any machine);
  * poses of :254-268: pose0 = I, pose1 = inverse(SE3(exp(0.1, 0.1, 0), (-0.5, -0.5, 0))); huber_delta 0.5 (:69); plus scaled / forward
    variants, depth maps with degenerate entries (0, < 0, +-inf, NaN, 1e-30, 65.535 m) and a depth map DECODED from a code that drives the
    proximity through zero (14 % of the pixels get a negative depth, ~1000 a depth beyond 100 m), see CASES / DEPTH_VARIANTS.

The expected outputs stored in the fixture come from oracle/_ref (the reference's own DenseSfm / LucasKanadeSE3 / kernel_warp_calculate)."""
import os

import numpy as np
from scipy import ndimage

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sfm_fixture_0_25.npz")
W, H, CS, AVG_DPT = 320, 240, 32, 2.0
GRID = 16   # px per cell of the stored Jacobian grid


def scenenet_cam(w=W, h=H):   # tests/testing_utils.h:34-40
    return np.array([np.float32(w // 2 / 0.5773502691896257), np.float32(h // 2 / 0.41421356237309503), w // 2, h // 2, w, h], np.float32)


def so3_exp(w):
    w = np.asarray(w, np.float64)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


def pose_inverse_of(rot, trs):
    """Sophus::SE3f(SO3f::exp(rot), trs).inverse() as (qx, qy, qz, qw, tx, ty, tz)  (ut_sfmaligner.cpp:254-268)."""
    from deepfactors_amd import synth
    R = so3_exp(rot)
    return synth.pose_qt(R.T, -R.T @ np.asarray(trs, np.float64))


def rel_pose_qt(pose0, pose1):
    """pose_10 = pose1^-1 * pose0 (warping.h:98-103)."""
    from deepfactors_amd import synth
    R0, R1 = synth.quat_to_R(pose0[:4]), synth.quat_to_R(pose1[:4])
    return synth.pose_qt(R1.T @ R0, R1.T @ (np.asarray(pose0[4:], np.float64) - np.asarray(pose1[4:], np.float64)))


IDENTITY = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)

# name -> (rot, trs, huber_delta): pose1 = inverse(SE3(exp(rot), trs)), pose0 = identity
CASES = {
    "ut": ((0.1, 0.1, 0.0), (-0.5, -0.5, 0.0), 0.5),             # exactly ut_sfmaligner.cpp:254-268 with huber 0.5 (:69)
    "ut01": ((0.01, 0.01, 0.0), (-0.05, -0.05, 0.0), 0.5),       # the same motion scaled by 0.1: > 90 % overlap
    "ut01_h01": ((0.01, 0.01, 0.0), (-0.05, -0.05, 0.0), 0.1),   # ... at the mapper's default huber_delta
    "fwd": ((0.004, -0.006, 0.003), (0.02, -0.01, 0.10), 0.1),   # pose_10 = SE3(exp(rot), trs) has t.z > 0, so depth-0 pixels (q = t) are VALID
                                                                 # correspondences in FindCorrespondence (warping.h:204-241)
}
DEPTH_VARIANTS = ("raw", "mixed", "decoded")
FAR_DEPTH = 65.535   # the largest value a 16-bit millimetre depth image holds


def upsample_jac(grid, w=W, h=H):
    """[gh][gw][cs] float32 grid -> prx_jac [h][w*cs]: separable linear interpolation, float32 multiply/add ufuncs only."""
    g = np.asarray(grid, np.float32)
    cs = g.shape[2]

    def axis_weights(n):
        u = np.arange(n, dtype=np.float32) / np.float32(GRID)
        i0 = np.floor(u).astype(np.int64)
        return i0, (u - i0.astype(np.float32)).astype(np.float32)

    iy, ty = axis_weights(h)
    ix, tx = axis_weights(w)
    assert iy.max() + 1 < g.shape[0] and ix.max() + 1 < g.shape[1]
    rows = g[iy] * (np.float32(1) - ty)[:, None, None] + g[iy + 1] * ty[:, None, None]           # [h][gw][cs]
    out = rows[:, ix] * (np.float32(1) - tx)[None, :, None] + rows[:, ix + 1] * tx[None, :, None]   # [h][w][cs]
    return np.ascontiguousarray(out.astype(np.float32).reshape(h, w * cs))


def degenerate_depth(dpt):
    """The 'mixed' variant: rectangles of depth 0, negative, +inf, NaN, -inf, denormal-small and 65.535 m inside the real depth map.
    (Farther than that the reference's own fp32 row is noise: dpix/dd = D R ray cancels to O(t / d), and a 10 km pixel moves the
    code-code block by 5e-5 between two evaluation orders of the same formulas -- measured oracle vs oracle/_ref -- so it pins nothing.)"""
    d = np.array(dpt, np.float32, copy=True)
    d[20:40, 30:60] = 0.0
    d[60:80, 100:140] = -1.5
    d[100:110, 200:260] = np.inf
    d[150:160, 40:80] = np.nan
    d[180:190, 150:200] = -np.inf
    d[200:210, 250:300] = 1e-30
    d[215:225, 20:60] = FAR_DEPTH
    return d


def load(path=GOLDEN):
    """Returns (inputs dict, raw npz): img0, img1 (blurred, [0,1]), dpt_raw (m, zeros kept), dpt1_raw, prx_orig, prx_jac, cam."""
    z = np.load(path)
    img0 = ndimage.uniform_filter(z["img0"].astype(np.float32) / np.float32(255), 25, mode="mirror")   # cv::blur == BORDER_REFLECT_101
    img1 = ndimage.uniform_filter(z["img1"].astype(np.float32) / np.float32(255), 25, mode="mirror")
    d0 = z["dpt0_mm"].astype(np.float32) / np.float32(1000)
    d1 = z["dpt1_mm"].astype(np.float32) / np.float32(1000)
    a = np.float32(AVG_DPT)
    prx_orig = (a / (a + d0)).astype(np.float32)                                                          # DepthToProx, warping.h:37-42
    return dict(img0=img0, img1=img1, dpt_raw=d0, dpt1_raw=d1, prx_orig=prx_orig, prx_jac=upsample_jac(z["jac_grid"]), cam=scenenet_cam(),
                code=np.zeros(CS, np.float32), code_neg=z["code_neg"].astype(np.float32)), z


def depth_variant(inp, variant, update_depth):
    """dpt0 as the reference test builds it -- UpdateDepth(code = 0) of the proximity image (ut_sfmaligner.cpp:279-283) -- then the
    degenerate rectangles for 'mixed'; 'decoded' = UpdateDepth(code_neg): prx = prx_orig + j . c crosses zero, so dpt = a / prx - a
    (warping.h:30-35) comes out negative (prx < 0) or enormous (prx -> 0+).  `update_depth(code, prx_orig, prx_jac, avg_dpt)` is the
    decoder under test or the oracle's."""
    d = update_depth(inp["code_neg"] if variant == "decoded" else inp["code"], inp["prx_orig"], inp["prx_jac"], AVG_DPT)
    return degenerate_depth(d) if variant == "mixed" else d


def expected(z, case, variant):
    """The reference's outputs for (case, depth variant) as plain objects with JtJ / Jtr / residual / inliers (+ masks)."""
    from helpers import Item
    k = f"{case}_{variant}_"
    sfm = Item(z[k + "sfm_JtJ"], z[k + "sfm_Jtr"], z[k + "sfm_residual"], z[k + "sfm_inliers"])
    se3 = Item(z[k + "se3_JtJ"], z[k + "se3_Jtr"], z[k + "se3_residual"], z[k + "se3_inliers"])
    valid0 = np.unpackbits(z[k + "sfm_valid0"])[: W * H].reshape(H, W).astype(bool)
    warp_mask = np.unpackbits(z[k + "warp_mask"])[: W * H].reshape(H, W).astype(bool)
    return dict(sfm=sfm, se3=se3, valid0=valid0, err=(float(z[k + "err_residual"]), int(z[k + "err_inliers"])),
                warp=(float(z[k + "warp_residual"]), int(z[k + "warp_inliers"])), warp_mask=warp_mask)
