"""ctypes binding of the CPU oracle (oracle/libdfx_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``deepfactors_amd/`` may import this.

All image arguments are C-contiguous numpy arrays; ``dtype`` float32 selects the ``_f32`` entry
points (the reference's device Scalar), float64 the ``_f64`` ones (the reference's ut_warping
tests run in double).  Poses are ``(qx, qy, qz, qw, tx, ty, tz)`` like ``Sophus::SE3``; cameras are
``(fx, fy, u0, v0, w, h)``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdfx_oracle.so")


def build(force=False):
    """Compile the oracle with g++ (see oracle/Makefile)."""
    src = os.path.join(_HERE, "dfx_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libdfx_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def build_native():
    """The timed CPU baseline of bench.py: the same source compiled for THIS machine (g++ -O3 -march=native -ffp-contract=off),
    at run time -- a -march=native object built elsewhere may not run here.  Switches this module to the native library."""
    global _lib, _SO
    native = os.path.join(_HERE, "libdfx_oracle_native.so")
    subprocess.check_call(["make", "-C", _HERE, "-B", "libdfx_oracle_native.so"], stdout=subprocess.DEVNULL)
    _SO, _lib = native, None
    return native


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_max_threads.restype = C.c_int
        for sfx, ct in (("f32", C.c_float), ("f64", C.c_double)):
            for name in ("huber_weight", "depth_jacobian_prx", "prox_to_depth", "depth_to_prox", "squared_error"):
                getattr(_lib, f"orc_{name}_{sfx}").restype = ct
            for name in ("huber_weight", "depth_jacobian_prx", "prox_to_depth", "depth_to_prox"):
                getattr(_lib, f"orc_{name}_{sfx}").argtypes = [ct, ct]
    return _lib


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32", C.c_float
    if dtype == np.float64:
        return "f64", C.c_double
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _arr(a, dtype):
    return np.ascontiguousarray(np.asarray(a, dtype=dtype))


def _fn(name, dtype):
    sfx, ct = _sfx(dtype)
    return getattr(lib(), f"orc_{name}_{sfx}"), ct


def max_threads():
    return int(lib().orc_max_threads())


class StepResult:
    """Mirror of JTJJrReductionItem (reduction_items.h:78-143): packed upper-tri JtJ, Jtr, residual, inliers."""

    def __init__(self, np_, dtype):
        self.np = np_
        self.JtJ = np.zeros(np_ * (np_ + 1) // 2, dtype)
        self.Jtr = np.zeros(np_, dtype)
        self.residual = 0.0
        self.inliers = 0

    def dense(self):
        """SquareUpperTriangularMatrix::toDenseMatrix()"""
        M = np.zeros((self.np, self.np), self.JtJ.dtype)
        iu = np.triu_indices(self.np)
        M[iu] = self.JtJ
        M = M + np.triu(M, 1).T
        return M


# ------------------------------------------------------------------------------------------------
# small math
# ------------------------------------------------------------------------------------------------
def relative_pose(a_qt, b_qt, dtype=np.float64):
    f, _ = _fn("relative_pose", dtype)
    a, b = _arr(a_qt, dtype), _arr(b_qt, dtype)
    R, t = np.zeros(9, dtype), np.zeros(3, dtype)
    ja, jb = np.zeros(36, dtype), np.zeros(36, dtype)
    f(_p(a), _p(b), _p(R), _p(t), _p(ja), _p(jb))
    return R.reshape(3, 3), t, ja.reshape(6, 6), jb.reshape(6, 6)


def quat_to_R(q, dtype=np.float64):
    f, _ = _fn("quat_to_R", dtype)
    q = _arr(q, dtype)
    R = np.zeros(9, dtype)
    f(_p(q), _p(R))
    return R.reshape(3, 3)


def so3_exp(w, dtype=np.float64):
    f, _ = _fn("so3_exp", dtype)
    w = _arr(w, dtype)
    R = np.zeros(9, dtype)
    f(_p(w), _p(R))
    return R.reshape(3, 3)


def perturb_pose(qt, idx, eps, dtype=np.float64):
    f, ct = _fn("perturb_pose", dtype)
    qt = _arr(qt, dtype)
    out = np.zeros(7, dtype)
    f(_p(qt), C.c_int(idx), ct(eps), _p(out))
    return out


def se3_solve_update(jtj21, jtr6, qt, dtype=np.float32):
    f, _ = _fn("se3_solve_update", dtype)
    qt = _arr(qt, dtype).copy()
    rc = f(_p(_arr(jtj21, dtype)), _p(_arr(jtr6, dtype)), _p(qt))
    if rc:
        raise ArithmeticError("singular 6x6 system")
    return qt


def ldlt_solve(upper_packed, b, dtype=np.float64):
    f, _ = _fn("ldlt_solve", dtype)
    b = _arr(b, dtype)
    x = np.zeros_like(b)
    rc = f(C.c_int(len(b)), _p(_arr(upper_packed, dtype)), _p(b), _p(x))
    if rc:
        raise ArithmeticError("singular system")
    return x


def correspondence(x, y, dpt, cam, qt, border=1, min_dpt=0.0, avg_dpt=2.0, dtype=np.float64):
    f, ct = _fn("correspondence", dtype)
    cam, qt = _arr(cam, dtype), _arr(qt, dtype)
    pix1, tpt = np.zeros(2, dtype), np.zeros(3, dtype)
    jp, jd, jx = np.zeros(12, dtype), np.zeros(2, dtype), np.zeros(2, dtype)
    f.restype = C.c_int
    ok = f(C.c_int(x), C.c_int(y), ct(dpt), _p(cam), _p(qt), C.c_int(border), ct(min_dpt), ct(avg_dpt),
           _p(pix1), _p(tpt), _p(jp), _p(jd), _p(jx))
    return dict(valid=bool(ok), pix1=pix1, tpt=tpt, jac_pose=jp.reshape(2, 6), jac_dpt=jd, jac_prx=jx)


def huber_weight(x, delta, dtype=np.float32):
    f, ct = _fn("huber_weight", dtype)
    return float(f(ct(x), ct(delta)))


def depth_jacobian_prx(d, a, dtype=np.float64):
    f, ct = _fn("depth_jacobian_prx", dtype)
    return float(f(ct(d), ct(a)))


def prox_to_depth(p, a, dtype=np.float64):
    f, ct = _fn("prox_to_depth", dtype)
    return float(f(ct(p), ct(a)))


def depth_to_prox(d, a, dtype=np.float64):
    f, ct = _fn("depth_to_prox", dtype)
    return float(f(ct(d), ct(a)))


def bilinear(img, u, v):
    img = np.ascontiguousarray(img)
    f, ct = _fn("bilinear", img.dtype)
    nch = 1 if img.ndim == 2 else img.shape[2]
    h, w = img.shape[:2]
    out = np.zeros(nch, img.dtype)
    f(_p(img), C.c_int(w), C.c_int(h), C.c_size_t(img.strides[0]), C.c_int(nch), ct(u), ct(v), _p(out))
    return out


# ------------------------------------------------------------------------------------------------
# the hot path
# ------------------------------------------------------------------------------------------------
def _img(a, dtype):
    a = np.asarray(a)
    if a.dtype != dtype or not a.flags["C_CONTIGUOUS"]:
        a = np.ascontiguousarray(a, dtype=dtype)
    return a


def se3_step(pose_qt, cam, img0, img1, dpt0, grad1, huber_delta, accum_f64=True, threads=1):
    """SE3Aligner::RunStep (cu_se3aligner.cpp:153-176)."""
    dt = np.asarray(img0).dtype
    f, ct = _fn("se3_step", dt)
    img0, img1, dpt0, grad1 = (_img(a, dt) for a in (img0, img1, dpt0, grad1))
    h, w = img0.shape
    res = StepResult(6, dt)
    r = ct(0)
    n = C.c_uint64(0)
    f(_p(_arr(pose_qt, dt)), _p(_arr(cam, dt)), _p(img0), _p(img1), _p(dpt0), _p(grad1), C.c_int(w), C.c_int(h),
      C.c_size_t(img0.strides[0]), C.c_size_t(grad1.strides[0]), ct(huber_delta), C.c_int(int(accum_f64)),
      C.c_int(threads), _p(res.JtJ), _p(res.Jtr), C.byref(r), C.byref(n))
    res.residual, res.inliers = float(r.value), int(n.value)
    return res


def se3_warp(pose_qt, cam, img0, img1, dpt0, accum_f64=True):
    """SE3Aligner::Warp (cu_se3aligner.cpp:125-151). Returns (img2, residual_signed_sum, inliers)."""
    dt = np.asarray(img0).dtype
    f, ct = _fn("se3_warp", dt)
    img0, img1, dpt0 = (_img(a, dt) for a in (img0, img1, dpt0))
    h, w = img0.shape
    img2 = np.empty_like(img0)
    r = ct(0)
    n = C.c_uint64(0)
    f(_p(_arr(pose_qt, dt)), _p(_arr(cam, dt)), _p(img0), _p(img1), _p(dpt0), _p(img2), C.c_int(w), C.c_int(h),
      C.c_size_t(img0.strides[0]), C.c_int(int(accum_f64)), C.byref(r), C.byref(n))
    return img2, float(r.value), int(n.value)


def sfm_step(pose0_qt, pose1_qt, cam, img0, img1, dpt0, prx_jac, grad1, huber_delta=0.1, avg_dpt=2.0, min_dpt=0.0,
             valid_border=2, valid0=None, accum_f64=True, threads=1):
    """SfmAligner::RunStep (cu_sfmaligner.cpp:149-185). prx_jac is [H][W*CS]."""
    dt = np.asarray(img0).dtype
    f, ct = _fn("sfm_step", dt)
    img0, img1, dpt0, prx_jac, grad1 = (_img(a, dt) for a in (img0, img1, dpt0, prx_jac, grad1))
    h, w = img0.shape
    jac2 = prx_jac.reshape(h, -1)
    cs = jac2.shape[1] // w
    res = StepResult(12 + cs, dt)
    r = ct(0)
    n = C.c_uint64(0)
    vptr = C.c_void_p(0)
    if valid0 is not None:
        assert valid0.dtype == dt and valid0.flags["C_CONTIGUOUS"] and valid0.shape == img0.shape
        vptr = _p(valid0)
    f(_p(_arr(pose0_qt, dt)), _p(_arr(pose1_qt, dt)), _p(_arr(cam, dt)), C.c_int(cs), _p(img0), _p(img1), _p(dpt0),
      _p(jac2), _p(grad1), vptr, C.c_int(w), C.c_int(h), C.c_size_t(img0.strides[0]), C.c_size_t(jac2.strides[0]),
      C.c_size_t(grad1.strides[0]), ct(huber_delta), ct(avg_dpt), ct(min_dpt), C.c_int(valid_border),
      C.c_int(int(accum_f64)), C.c_int(threads), _p(res.JtJ), _p(res.Jtr), C.byref(r), C.byref(n))
    res.residual, res.inliers = float(r.value), int(n.value)
    return res


def sfm_error(pose0_qt, pose1_qt, cam, img0, img1, dpt0, huber_delta=0.1, accum_f64=True):
    """SfmAligner::EvaluateError (cu_sfmaligner.cpp:120-147). Returns (residual, inliers)."""
    dt = np.asarray(img0).dtype
    f, ct = _fn("sfm_error", dt)
    img0, img1, dpt0 = (_img(a, dt) for a in (img0, img1, dpt0))
    h, w = img0.shape
    r = ct(0)
    n = C.c_uint64(0)
    f(_p(_arr(pose0_qt, dt)), _p(_arr(pose1_qt, dt)), _p(_arr(cam, dt)), _p(img0), _p(img1), _p(dpt0), C.c_int(w),
      C.c_int(h), C.c_size_t(img0.strides[0]), ct(huber_delta), C.c_int(int(accum_f64)), C.byref(r), C.byref(n))
    return float(r.value), int(n.value)


def update_depth(code, prx_orig, prx_jac, avg_dpt):
    """df::UpdateDepth (cu_image_proc.cpp:248-277)."""
    dt = np.asarray(prx_orig).dtype
    f, ct = _fn("update_depth", dt)
    prx_orig, prx_jac = _img(prx_orig, dt), _img(prx_jac, dt)
    h, w = prx_orig.shape
    jac2 = prx_jac.reshape(h, -1)
    cs = jac2.shape[1] // w
    code = _arr(code, dt)
    assert code.shape == (cs,)
    out = np.empty_like(prx_orig)
    f(C.c_int(cs), _p(code), _p(prx_orig), _p(jac2), ct(avg_dpt), _p(out), C.c_int(w), C.c_int(h),
      C.c_size_t(prx_orig.strides[0]), C.c_size_t(jac2.strides[0]))
    return out


def depth_aligner_step(code, tgt_dpt, prx_orig, prx_jac, avg_dpt=2.0, accum_f64=True):
    """DepthAligner::RunStep (cu_depthaligner.cpp:32-110)."""
    dt = np.asarray(prx_orig).dtype
    f, ct = _fn("depth_aligner_step", dt)
    tgt_dpt, prx_orig, prx_jac = _img(tgt_dpt, dt), _img(prx_orig, dt), _img(prx_jac, dt)
    h, w = prx_orig.shape
    jac2 = prx_jac.reshape(h, -1)
    cs = jac2.shape[1] // w
    res = StepResult(cs, dt)
    r = ct(0)
    n = C.c_uint64(0)
    f(C.c_int(cs), _p(_arr(code, dt)), _p(tgt_dpt), _p(prx_orig), _p(jac2), ct(avg_dpt), C.c_int(w), C.c_int(h),
      C.c_size_t(prx_orig.strides[0]), C.c_size_t(jac2.strides[0]), C.c_int(int(accum_f64)), _p(res.JtJ), _p(res.Jtr),
      C.byref(r), C.byref(n))
    res.residual, res.inliers = float(r.value), int(n.value)
    return res


def sparse_geometric(pose0_qt, pose1_qt, code0, code1, cam, points, prx0, jac0, prx1, jac1, dpt_grad1, huber_delta, avg_dpt=2.0):
    """SparseGeometricFactor::linearize (sparse_geometric_factor.cpp:147-275): rows [N][12 + 2 CS + 1]."""
    dt = np.asarray(prx0).dtype
    f, ct = _fn("sparse_geometric", dt)
    prx0, jac0, prx1, jac1, dpt_grad1 = (_img(a, dt) for a in (prx0, jac0, prx1, jac1, dpt_grad1))
    h, w = prx0.shape
    j0, j1 = jac0.reshape(h, -1), jac1.reshape(h, -1)
    cs = j0.shape[1] // w
    pts = np.ascontiguousarray(np.asarray(points, np.int32).reshape(-1, 2))
    rows = np.zeros((len(pts), 12 + 2 * cs + 1), dt)
    f(_p(_arr(pose0_qt, dt)), _p(_arr(pose1_qt, dt)), _p(_arr(code0, dt)), _p(_arr(code1, dt)), _p(_arr(cam, dt)), C.c_int(cs), _p(pts),
      C.c_int(len(pts)), _p(prx0), _p(j0), _p(prx1), _p(j1), _p(dpt_grad1), C.c_int(w), C.c_int(h), C.c_size_t(prx0.strides[0]),
      C.c_size_t(j0.strides[0]), C.c_size_t(dpt_grad1.strides[0]), ct(huber_delta), ct(avg_dpt), _p(rows))
    return rows


def sobel(img):
    """df::SobelGradients (cu_image_proc.cpp:57-112): returns [H][W][2] = (gx, gy)/8, clamped borders."""
    img = np.ascontiguousarray(img)
    f, _ = _fn("sobel", img.dtype)
    h, w = img.shape
    g = np.empty((h, w, 2), img.dtype)
    f(_p(img), _p(g), C.c_int(w), C.c_int(h), C.c_size_t(img.strides[0]), C.c_size_t(g.strides[0]))
    return g


def blur_down(img):
    """df::GaussianBlurDown (cu_image_proc.cpp:134-186): 5x5 binomial, decimate by 2."""
    img = np.ascontiguousarray(img)
    f, _ = _fn("blur_down", img.dtype)
    h, w = img.shape
    out = np.empty((h // 2, w // 2), img.dtype)
    f(_p(img), _p(out), C.c_int(w), C.c_int(h), C.c_size_t(img.strides[0]), C.c_int(w // 2), C.c_int(h // 2),
      C.c_size_t(out.strides[0]))
    return out


def squared_error(a, b, accum_f64=True):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b, dtype=a.dtype)
    f, _ = _fn("squared_error", a.dtype)
    h, w = a.shape
    return float(f(_p(a), _p(b), C.c_int(w), C.c_int(h), C.c_size_t(a.strides[0]), C.c_int(int(accum_f64))))
