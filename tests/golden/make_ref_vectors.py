"""Generates tests/golden/ref_vectors.npz: inputs AND the outputs of the REFERENCE'S OWN per-pixel code on them.

Run in the build container only (needs /root/reference to build oracle/_ref):
    python tests/golden/make_ref_vectors.py

The outputs come from oracle/_ref/libdfx_ref.so, i.e. the reference's unmodified headers
    sources/common/algorithm/{warping,dense_sfm,lucas_kanade_se3,pinhole_camera_impl,m_estimators}.h
driven by oracle/ref_harness.cpp the way tests/ut_sfmaligner.cpp:299-315 (host loop over DenseSfm) and
tests/ut_se3aligner.cpp:173-211 drive them.  They are the committed known-answer vectors for
    SfmAligner::RunStep   (cu_sfmaligner.cpp:149-185)    -> sfm_{JtJ,Jtr,residual,inliers,valid0}
    SfmAligner::EvaluateError (cu_sfmaligner.cpp:120-147) -> err_{residual,inliers}
    SE3Aligner::RunStep   (cu_se3aligner.cpp:153-176)     -> se3_{JtJ,Jtr,residual,inliers}
    UpdateDepth           (cu_image_proc.cpp:248-277)     -> dpt
    RelativePose          (warping.h:98-137)              -> rel_{pose,J_a,J_b}
so that the oracle (CPU tests) and the HIP path (GPU tests) stay pinned to reference-derived numbers even where the
prebuilt oracle/_ref library is absent (it is git-ignored; /root/reference does not exist on the GPU box).

The inputs are stored too (float32, small sizes): deepfactors_amd.synth evaluates its fields with torch, whose
transcendental functions may differ in the last bit between builds, and a known-answer test must not depend on that.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "ref_vectors.npz")

# (name, w, h, cs, seed, huber_delta, motion_scale): two code sizes (NCB = 2 and 4 of the step kernel), odd huber / motion on the second
CASES = [("a", 64, 40, 32, 0x601D, 0.1, 1.0),
         ("b", 64, 24, 64, 0x601E, 0.05, 0.6)]


def main():
    from deepfactors_amd import synth
    from oracle import dfx_ref as ref
    if not os.path.isdir(os.path.join(ref.REFERENCE, "sources")):
        sys.exit("reference tree not present; the vectors can only be regenerated in the build container")
    ref.build()
    out = {"sources": np.frombuffer(ref.lib().ref_sources(), dtype=np.uint8)}
    for name, w, h, cs, seed, huber, motion in CASES:
        n = synth.to_numpy(synth.make_pair(w, h, cs, seed=seed, device="cpu", motion_scale=motion))
        pose1 = n["pose1"].copy()
        pose1[4] += 0.01                     # off the minimum, so Jtr is not ~0
        valid0 = np.zeros((h, w), np.float32)
        s = ref.sfm_step(n["pose0"], pose1, n["cam"], n["img0"], n["img1"], n["dpt0"], n["prx_jac"], n["grad1"], huber_delta=huber, valid0=valid0)
        e_res, e_inl = ref.sfm_error(n["pose0"], pose1, n["cam"], n["img0"], n["img1"], n["dpt0"], n["grad1"], huber)
        rel, ja, jb = ref.relative_pose(pose1, n["pose0"])
        k = ref.se3_step(rel, n["cam"], n["img0"], n["img1"], n["dpt0"], n["grad1"], huber)
        dpt = ref.update_depth(n["code"], n["prx_orig"], n["prx_jac"], 2.0)
        assert s.inliers > 0.5 * w * h and k.inliers > 0.5 * w * h
        for key in ("cam", "pose0", "code", "img0", "img1", "dpt0", "grad1", "prx_orig", "prx_jac"):
            out[f"{name}_{key}"] = np.asarray(n[key], np.float32)
        out.update({f"{name}_pose1": pose1.astype(np.float32), f"{name}_huber": np.float32(huber), f"{name}_cs": np.int32(cs),
                    f"{name}_sfm_JtJ": s.JtJ, f"{name}_sfm_Jtr": s.Jtr, f"{name}_sfm_residual": np.float64(s.residual), f"{name}_sfm_inliers": np.int64(s.inliers),
                    f"{name}_sfm_valid0": valid0.astype(np.uint8),
                    f"{name}_err_residual": np.float64(e_res), f"{name}_err_inliers": np.int64(e_inl),
                    f"{name}_rel_pose": rel, f"{name}_rel_Ja": ja, f"{name}_rel_Jb": jb,
                    f"{name}_se3_JtJ": k.JtJ, f"{name}_se3_Jtr": k.Jtr, f"{name}_se3_residual": np.float64(k.residual), f"{name}_se3_inliers": np.int64(k.inliers),
                    f"{name}_dpt": dpt})
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    make_f3(synth, ref)
    make_f1(synth, ref)


OUT_F3 = os.path.join(HERE, "ref_vectors_f3.npz")


def make_f3(synth, ref):
    """tests/golden/ref_vectors_f3.npz: SparseGeometricFactor<float,32>::linearize (core/gtsam/sparse_geometric_factor.cpp:147-275, compiled
    unmodified) and kernel_depthaligner_run_step (cuda/cu_depthaligner.cpp:32-72) of the reference on stored inputs (oracle/ref_harness_f3.cpp)."""
    from oracle import dfx_oracle as orc   # only for the Sobel gradient of kf1's depth, an INPUT of the factor (mapper.cpp:998-1000)
    w, h, cs, npts = 64, 40, 32, 240
    k0 = synth.to_numpy(synth.make_pair(w, h, cs, seed=0x6F30, device="cpu"))
    k1 = synth.to_numpy(synth.make_pair(w, h, cs, seed=0x6F31, device="cpu", motion_scale=0.7))
    rng = np.random.default_rng(0x6F3)
    pts = np.stack([rng.integers(0, w, npts), rng.integers(0, h, npts)], axis=1).astype(np.int32)
    dgrad1 = orc.sobel(k1["dpt0"]).astype(np.float32)
    pose1 = k0["pose1"].copy(); pose1[4] += 0.08; pose1[5] -= 0.03
    huber = 0.05
    rows = ref.sparse_geometric(k0["pose0"], pose1, k0["code"], k1["code"], k0["cam"], pts, k0["prx_orig"], k0["prx_jac"], k1["prx_orig"], k1["prx_jac"], dgrad1, huber)
    zero = ~rows.any(axis=1)
    assert 0 < zero.sum() < npts
    code = (k0["code"] + rng.normal(0, 0.05, cs)).astype(np.float32)
    tgt = (k0["dpt0"] + rng.normal(0, 0.02, k0["dpt0"].shape)).astype(np.float32)
    d = ref.depth_aligner_step(code, tgt, k0["prx_orig"], k0["prx_jac"])
    out = dict(cam=k0["cam"], pose0=k0["pose0"], pose1=pose1.astype(np.float32), code0=k0["code"], code1=k1["code"], points=pts, huber=np.float32(huber),
               prx0=k0["prx_orig"], jac0=k0["prx_jac"], prx1=k1["prx_orig"], jac1=k1["prx_jac"], dgrad1=dgrad1, sg_rows=rows,
               da_code=code, da_tgt=tgt, da_JtJ=d.JtJ, da_Jtr=d.Jtr, da_residual=np.float64(d.residual), da_inliers=np.int64(d.inliers),
               sources=np.frombuffer(b"core/gtsam/sparse_geometric_factor.cpp (unmodified, #included); cuda/cu_depthaligner.cpp:32-72 (kernel template, cut out at build time)",
                                     dtype=np.uint8))
    np.savez_compressed(OUT_F3, **out)
    print("wrote", OUT_F3, os.path.getsize(OUT_F3), "bytes")


OUT_F1 = os.path.join(HERE, "ref_vectors_f1.npz")


def make_f1(synth, ref):
    """tests/golden/ref_vectors_f1.npz: the reference's kernel bodies kernel_sobel_gradients / kernel_gaussian_blur_down / kernel_squared_error
    (cuda/cu_image_proc.cpp:57-206) and kernel_warp_calculate (cuda/cu_se3aligner.cpp:61-113), cut out at build time (oracle/ref_harness_f1.cpp),
    on stored inputs: an even and an odd image size."""
    out = {"sources": np.frombuffer(b"cuda/cu_image_proc.cpp:34-206 (SetSobelCoefficients, kernel_sobel_gradients, SetGaussCoefficients, kernel_gaussian_blur_down, "
                                    b"kernel_squared_error) + cuda/cu_se3aligner.cpp:61-113 (kernel_warp_calculate): kernel bodies cut out at build time", dtype=np.uint8)}
    for name, w, h, seed in (("a", 64, 40, 0x6F10), ("b", 53, 37, 0x6F11)):
        n = synth.to_numpy(synth.make_pair(w, h, 16, seed=seed, device="cpu", with_decoder=False))
        img2, r, k = ref.se3_warp(n["pose10_true"], n["cam"], n["img0"], n["img1"], n["dpt0"])
        assert k > 0.5 * w * h
        out.update({f"{name}_cam": n["cam"], f"{name}_pose10": n["pose10_true"], f"{name}_img0": n["img0"], f"{name}_img1": n["img1"], f"{name}_dpt0": n["dpt0"],
                    f"{name}_sobel": ref.sobel_gradients(n["img0"]), f"{name}_blur": ref.gaussian_blur_down(n["img0"]),
                    f"{name}_sqerr": np.float32(ref.squared_error(n["img0"], n["img1"])),
                    f"{name}_warp_img2": img2, f"{name}_warp_residual": np.float32(r), f"{name}_warp_inliers": np.int64(k)})
    np.savez_compressed(OUT_F1, **out)
    print("wrote", OUT_F1, os.path.getsize(OUT_F1), "bytes")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--f1-only":   # the other files are left as committed
        from deepfactors_amd import synth as _synth
        from oracle import dfx_ref as _ref
        _ref.build()
        make_f1(_synth, _ref)
    elif len(sys.argv) > 1 and sys.argv[1] == "--f3-only":   # the first file is left as committed
        from deepfactors_amd import synth as _synth
        from oracle import dfx_ref as _ref
        _ref.build()
        make_f3(_synth, _ref)
    else:
        main()
