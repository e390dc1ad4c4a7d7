"""ctypes binding of oracle/_ref/libdfx_ref.so: the REFERENCE'S OWN L0 headers compiled unmodified (oracle/ref_harness.cpp,
oracle/Makefile target `ref`).

TEST INFRASTRUCTURE ONLY: importable from tests/ -- it pins the oracle (oracle/dfx_oracle.cpp, our restatement) and, through
it, the HIP path to the reference's own per-pixel code.  Nothing under ``deepfactors_amd/`` may import this.

The library is built only where /root/reference exists (the build container); the built file travels to the GPU box.
``available()`` tells whether it is there."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libdfx_ref.so")
REFERENCE = os.environ.get("DFX_REFERENCE", "/root/reference")
_lib = None


def build():
    """(Re)build from the reference sources when they are present; otherwise keep the prebuilt library."""
    if os.path.isdir(os.path.join(REFERENCE, "sources", "common", "algorithm")):
        subprocess.check_call(["make", "-C", _HERE, "ref", f"REFERENCE={REFERENCE}"], stdout=subprocess.DEVNULL)
    return _SO


def available():
    return os.path.exists(_SO)


def lib():
    global _lib
    if _lib is None:
        if not available():
            build()
        _lib = C.CDLL(_SO)
        _lib.ref_huber_weight_f32.restype = C.c_float
        _lib.ref_huber_weight_f32.argtypes = [C.c_float, C.c_float]
        _lib.ref_depth_jacobian_prx_f32.restype = C.c_float
        _lib.ref_depth_jacobian_prx_f32.argtypes = [C.c_float, C.c_float]
        _lib.ref_sources.restype = C.c_char_p
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


class StepResult:
    def __init__(self, np_):
        self.np = np_
        self.JtJ = np.zeros(np_ * (np_ + 1) // 2, np.float64)
        self.Jtr = np.zeros(np_, np.float64)
        self.residual = 0.0
        self.inliers = 0

    def dense(self):
        M = np.zeros((self.np, self.np), np.float64)
        M[np.triu_indices(self.np)] = self.JtJ
        return M + np.triu(M, 1).T


def sfm_step(pose0_qt, pose1_qt, cam, img0, img1, dpt0, prx_jac, grad1, huber_delta=0.1, avg_dpt=2.0, min_dpt=0.0, valid_border=2, valid0=None):
    """The reference's host evaluation of SfmAligner::RunStep (tests/ut_sfmaligner.cpp:299-315): RelativePose, DenseSfm per pixel."""
    img0, img1, dpt0, prx_jac, grad1 = (_f32(a) for a in (img0, img1, dpt0, prx_jac, grad1))
    h, w = img0.shape
    jac2 = prx_jac.reshape(h, -1)
    cs = jac2.shape[1] // w
    res = StepResult(12 + cs)
    r, n = C.c_double(0), C.c_uint64(0)
    vptr = C.c_void_p(0)
    if valid0 is not None:
        assert valid0.dtype == np.float32 and valid0.flags["C_CONTIGUOUS"] and valid0.shape == img0.shape
        vptr = _p(valid0)
    rc = lib().ref_sfm_step_f32(_p(_f32(pose0_qt)), _p(_f32(pose1_qt)), _p(_f32(cam)), C.c_int(cs), _p(img0), _p(img1), _p(dpt0), C.c_void_p(0), vptr,
                                _p(jac2), _p(grad1), C.c_int(w), C.c_int(h), C.c_size_t(img0.strides[0]), C.c_size_t(jac2.strides[0]),
                                C.c_size_t(grad1.strides[0]), C.c_float(huber_delta), C.c_float(avg_dpt), C.c_float(min_dpt), C.c_int(valid_border),
                                _p(res.JtJ), _p(res.Jtr), C.byref(r), C.byref(n))
    if rc != 0:
        raise ValueError(f"code size {cs} is not instantiated in the reference harness")
    res.residual, res.inliers = float(r.value), int(n.value)
    return res


def sfm_error(pose0_qt, pose1_qt, cam, img0, img1, dpt0, grad1, huber_delta=0.1, avg_dpt=2.0):
    img0, img1, dpt0, grad1 = (_f32(a) for a in (img0, img1, dpt0, grad1))
    h, w = img0.shape
    r, n = C.c_double(0), C.c_uint64(0)
    lib().ref_sfm_error_f32(_p(_f32(pose0_qt)), _p(_f32(pose1_qt)), _p(_f32(cam)), _p(img0), _p(img1), _p(dpt0), _p(grad1), C.c_int(w), C.c_int(h),
                            C.c_size_t(img0.strides[0]), C.c_size_t(grad1.strides[0]), C.c_float(huber_delta), C.c_float(avg_dpt), C.byref(r), C.byref(n))
    return float(r.value), int(n.value)


def se3_step(pose_qt, cam, img0, img1, dpt0, grad1, huber_delta):
    img0, img1, dpt0, grad1 = (_f32(a) for a in (img0, img1, dpt0, grad1))
    h, w = img0.shape
    res = StepResult(6)
    r, n = C.c_double(0), C.c_uint64(0)
    lib().ref_se3_step_f32(_p(_f32(pose_qt)), _p(_f32(cam)), _p(img0), _p(img1), _p(dpt0), _p(grad1), C.c_int(w), C.c_int(h), C.c_size_t(img0.strides[0]),
                           C.c_size_t(grad1.strides[0]), C.c_float(huber_delta), _p(res.JtJ), _p(res.Jtr), C.byref(r), C.byref(n))
    res.residual, res.inliers = float(r.value), int(n.value)
    return res


def se3_solve_and_update(jtj21, jtr6, pose_qt):
    """df::SE3SolveAndUpdate (lucas_kanade_se3.h:85-95) in float; returns the updated pose."""
    qt = _f32(pose_qt).copy()
    lib().ref_se3_solve_and_update_f32(_p(_f32(jtj21)), _p(_f32(jtr6)), _p(qt))
    return qt


def update_depth(code, prx_orig, prx_jac, avg_dpt):
    prx_orig, prx_jac = _f32(prx_orig), _f32(prx_jac)
    h, w = prx_orig.shape
    jac2 = prx_jac.reshape(h, -1)
    cs = jac2.shape[1] // w
    out = np.empty_like(prx_orig)
    rc = lib().ref_update_depth_f32(C.c_int(cs), _p(_f32(code)), _p(prx_orig), _p(jac2), C.c_float(avg_dpt), _p(out), C.c_int(w), C.c_int(h),
                                    C.c_size_t(prx_orig.strides[0]), C.c_size_t(jac2.strides[0]))
    if rc != 0:
        raise ValueError(f"code size {cs} is not instantiated in the reference harness")
    return out


def relative_pose(a_qt, b_qt):
    out, ja, jb = np.zeros(7, np.float32), np.zeros(36, np.float32), np.zeros(36, np.float32)
    lib().ref_relative_pose_f32(_p(_f32(a_qt)), _p(_f32(b_qt)), _p(out), _p(ja), _p(jb))
    return out, ja.reshape(6, 6), jb.reshape(6, 6)


def camera_pyramid(cam, levels):
    out = np.zeros(levels * 6, np.float32)
    lib().ref_camera_pyramid_f32(_p(_f32(cam)), int(levels), _p(out))
    return out.reshape(levels, 6)


def huber_weight(x, delta):
    return float(lib().ref_huber_weight_f32(x, delta))


def depth_jacobian_prx(d, a):
    return float(lib().ref_depth_jacobian_prx_f32(d, a))


def sparse_geometric(pose0_qt, pose1_qt, code0, code1, cam, points, prx0, jac0, prx1, jac1, dpt_grad1, huber_delta):
    """The reference's SparseGeometricFactor<float,32>::linearize (core/gtsam/sparse_geometric_factor.cpp:147-275, compiled unmodified by
    oracle/ref_harness_f3.cpp): rows [N][12 + 64 + 1] = [pose0 | pose1 | code0 | code1 | err] of the JacobianFactor (doubles, as GTSAM holds
    them).  avg_dpt is 2.0 inside the reference."""
    prx0, jac0, prx1, jac1, dpt_grad1 = (_f32(a).copy() for a in (prx0, jac0, prx1, jac1, dpt_grad1))
    h, w = prx0.shape
    assert jac0.reshape(h, -1).shape[1] == 32 * w, "the harness instantiates the reference's code size 32"
    pts = np.ascontiguousarray(np.asarray(points, np.int32).reshape(-1, 2))
    rows = np.zeros((len(pts), 12 + 64 + 1), np.float64)
    L = lib()
    L.ref_sparse_geometric_cs32.restype = C.c_int
    rc = L.ref_sparse_geometric_cs32(_p(_f32(pose0_qt)), _p(_f32(pose1_qt)), _p(_f32(code0)), _p(_f32(code1)), _p(_f32(cam)), C.c_int(w), C.c_int(h), _p(pts),
                                     C.c_int(len(pts)), _p(prx0), _p(jac0), _p(prx1), _p(jac1), _p(dpt_grad1), C.c_float(huber_delta), _p(rows))
    if rc != 0:
        raise RuntimeError(f"ref_sparse_geometric_cs32 failed: {rc}")
    return rows


def depth_aligner_step(code, tgt_dpt, prx_orig, prx_jac):
    """The reference's kernel_depthaligner_run_step (cuda/cu_depthaligner.cpp:32-72, cut out at build time and run as a host loop over all
    pixels in order, float accumulation like one CUDA thread): JTJJrReductionItem<float,32>.  avg_dpt is 2.0 inside the reference."""
    tgt_dpt, prx_orig, prx_jac = (_f32(a).copy() for a in (tgt_dpt, prx_orig, prx_jac))
    h, w = prx_orig.shape
    assert prx_jac.reshape(h, -1).shape[1] == 32 * w
    res = StepResult(32)
    jtj, jtr = np.zeros(32 * 33 // 2, np.float32), np.zeros(32, np.float32)
    r, n = C.c_float(0), C.c_uint64(0)
    lib().ref_depth_aligner_step_cs32(_p(_f32(code)), _p(tgt_dpt), _p(prx_orig), _p(prx_jac), C.c_int(w), C.c_int(h), _p(jtj), _p(jtr), C.byref(r), C.byref(n))
    res.JtJ, res.Jtr, res.residual, res.inliers = jtj.astype(np.float64), jtr.astype(np.float64), float(r.value), int(n.value)
    return res


# ---- part 3 (oracle/ref_harness_f1.cpp): the reference's image-proc kernel bodies and SE3Aligner::Warp's, cut out at build time ----
def sobel_gradients(img):
    """kernel_sobel_gradients (cuda/cu_image_proc.cpp:57-92) over every pixel: [H][W][2] = (gx, gy)."""
    a = _f32(img).copy()
    h, w = a.shape
    out = np.zeros((h, w, 2), np.float32)
    lib().ref_sobel_gradients(_p(a), C.c_int(w), C.c_int(h), _p(out))
    return out


def gaussian_blur_down(img):
    """kernel_gaussian_blur_down (cu_image_proc.cpp:134-164): [H/2][W/2]."""
    a = _f32(img).copy()
    h, w = a.shape
    out = np.zeros((h // 2, w // 2), np.float32)
    lib().ref_gaussian_blur_down(_p(a), C.c_int(w), C.c_int(h), _p(out), C.c_int(w // 2), C.c_int(h // 2))
    return out


def squared_error(a, b):
    """kernel_squared_error (cu_image_proc.cpp:190-206), its float sum in pixel order."""
    a, b = _f32(a).copy(), _f32(b).copy()
    h, w = a.shape
    f = lib().ref_squared_error
    f.restype = C.c_float
    return float(f(_p(a), _p(b), C.c_int(w), C.c_int(h)))


def se3_warp(pose_qt, cam, img0, img1, dpt0):
    """kernel_warp_calculate (cuda/cu_se3aligner.cpp:61-113): (img2, signed residual sum, inliers)."""
    i0, i1, d0 = _f32(img0).copy(), _f32(img1).copy(), _f32(dpt0).copy()
    h, w = i0.shape
    img2 = np.zeros((h, w), np.float32)
    r = C.c_float(0)
    n = C.c_uint64(0)
    lib().ref_se3_warp(_p(_f32(pose_qt)), _p(_f32(cam)), _p(i0), _p(i1), _p(d0), C.c_int(w), C.c_int(h), _p(img2), C.byref(r), C.byref(n))
    return img2, float(r.value), int(n.value)
