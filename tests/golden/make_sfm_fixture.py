"""Generates tests/golden/sfm_fixture_0_25.npz: the reference's own SfM test images and the outputs of the REFERENCE'S OWN per-pixel code.

Run in the build container only (needs /root/reference for the images and to build oracle/_ref):
    python tests/golden/make_sfm_fixture.py

Inputs (stored RAW so the file stays small; tests/sfm_fixture.py redoes the preprocessing of ut_sfmaligner.cpp:84-98):
    img0, img1     data/testimg/0.jpg, 25.jpg   uint8 grayscale (ut_sfmaligner.cpp:42-43)
    dpt0_mm/dpt1_mm data/testimg/0.png, 25.png   uint16 millimetres; 0.png has 124 zero pixels, kept
    jac_grid       [16][21][32] float32 ~ N(0, 0.05), numpy default_rng(0x5F25): the seeded stand-in for the decoder's prx_jac
    code_neg       [32] float32 ~ N(0, 2): the code of the 'decoded' depth variant (drives the proximity through zero)
Outputs, per case of tests/sfm_fixture.py:CASES x DEPTH_VARIANTS, from oracle/_ref/libdfx_ref.so (the reference's unmodified
warping.h / dense_sfm.h / lucas_kanade_se3.h / pinhole_camera_impl.h / m_estimators.h and the kernel_warp_calculate body):
    sfm_{JtJ,Jtr,residual,inliers,valid0}   SfmAligner::RunStep       host loop of ut_sfmaligner.cpp:299-315
    err_{residual,inliers}                  SfmAligner::EvaluateError  dense_sfm.h:72-119
    se3_{JtJ,Jtr,residual,inliers}          SE3Aligner::RunStep        lucas_kanade_se3.h:35-77 at pose_10
    warp_{residual,inliers,mask}            SE3Aligner::Warp           cu_se3aligner.cpp:61-113 (mask = pixels it rendered)
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF_IMG = "/root/reference/data/testimg"


def gray(path):
    """cv::imread(IMREAD_GRAYSCALE) == libjpeg's luma plane; PIL's draft('L') asks libjpeg for the same plane."""
    im = Image.open(path)
    im.draft("L", im.size)
    return np.asarray(im.convert("L"), dtype=np.uint8)


def main():
    import sfm_fixture as fx
    from oracle import dfx_ref as ref
    if not os.path.isdir(REF_IMG):
        sys.exit("reference images not present; the fixture can only be regenerated in the build container")
    ref.build()
    rng = np.random.default_rng(0x5F25)
    raw = dict(img0=gray(os.path.join(REF_IMG, "0.jpg")), img1=gray(os.path.join(REF_IMG, "25.jpg")),
               dpt0_mm=np.asarray(Image.open(os.path.join(REF_IMG, "0.png"))).astype(np.uint16),
               dpt1_mm=np.asarray(Image.open(os.path.join(REF_IMG, "25.png"))).astype(np.uint16),
               jac_grid=(rng.standard_normal((fx.H // fx.GRID + 1, fx.W // fx.GRID + 1, fx.CS)) * 0.05).astype(np.float32),
               code_neg=(rng.standard_normal(fx.CS) * 2.0).astype(np.float32))
    assert raw["img0"].shape == raw["img1"].shape == raw["dpt0_mm"].shape == (fx.H, fx.W) and int((raw["dpt0_mm"] == 0).sum()) == 124
    tmp = os.path.join(HERE, "_tmp_inputs.npz")
    np.savez(tmp, **raw)
    inp, _ = fx.load(tmp)
    os.remove(tmp)
    out = dict(raw)
    out["sources"] = np.frombuffer(ref.lib().ref_sources(), dtype=np.uint8)
    for variant in fx.DEPTH_VARIANTS:
        dpt0 = fx.depth_variant(inp, variant, ref.update_depth)
        for case, (rot, trs, huber) in fx.CASES.items():
            pose0, pose1 = fx.IDENTITY, fx.pose_inverse_of(rot, trs)
            grad1 = ref.sobel_gradients(inp["img1"])
            valid0 = np.zeros((fx.H, fx.W), np.float32)
            s = ref.sfm_step(pose0, pose1, inp["cam"], inp["img0"], inp["img1"], dpt0, inp["prx_jac"], grad1, huber_delta=huber, avg_dpt=fx.AVG_DPT,
                             valid0=valid0)
            e_res, e_inl = ref.sfm_error(pose0, pose1, inp["cam"], inp["img0"], inp["img1"], dpt0, grad1, huber, fx.AVG_DPT)
            rel, _, _ = ref.relative_pose(pose1, pose0)
            k = ref.se3_step(rel, inp["cam"], inp["img0"], inp["img1"], dpt0, grad1, huber)
            img2, w_res, w_inl = ref.se3_warp(rel, inp["cam"], inp["img0"], inp["img1"], dpt0)
            mask = img2 != 0
            assert int(mask.sum()) == w_inl, "a rendered sample was exactly 0: the mask would miss it"
            assert s.inliers > 0 and int(valid0.sum()) == s.inliers
            assert np.isfinite(s.JtJ).all() and np.isfinite(s.Jtr).all() and np.isfinite(k.JtJ).all(), (case, variant)
            p = f"{case}_{variant}_"
            out.update({p + "pose1": pose1, p + "huber": np.float32(huber),
                        p + "sfm_JtJ": s.JtJ, p + "sfm_Jtr": s.Jtr, p + "sfm_residual": np.float64(s.residual), p + "sfm_inliers": np.int64(s.inliers),
                        p + "sfm_valid0": np.packbits(valid0.astype(bool)),
                        p + "err_residual": np.float64(e_res), p + "err_inliers": np.int64(e_inl),
                        p + "se3_JtJ": k.JtJ, p + "se3_Jtr": k.Jtr, p + "se3_residual": np.float64(k.residual), p + "se3_inliers": np.int64(k.inliers),
                        p + "warp_residual": np.float64(w_res), p + "warp_inliers": np.int64(w_inl), p + "warp_mask": np.packbits(mask)})
            M = s.dense()
            print(f"{case:9s} {variant:5s} sfm inliers {s.inliers:6d} ({s.inliers / (fx.W * fx.H):.2f}) residual {s.residual:9.3f} |G11| {np.abs(M[:6, :6]).max():.3g} "
                  f"|G33| {np.abs(M[12:, 12:]).max():.3g}  err {e_inl} se3 {k.inliers} warp {w_inl}")
    path = os.path.join(HERE, "sfm_fixture_0_25.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
