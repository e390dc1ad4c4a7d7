// dfx_oracle.cpp -- CPU ORACLE for the DeepFactors dense-alignment hot path.
//
// *** TEST INFRASTRUCTURE, NOT PRODUCT CODE. ***
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
// and only as the checker / the timed CPU baseline.  The shipped path (libdfx.so, HIP) never
// links, loads or falls back to anything in oracle/.
//
// What it is: a dependency-free C++17 restatement (no Eigen / Sophus / VisionCore -- none of them
// is available in the reference tree, all submodules are empty) of the reference's per-pixel math.
// Every function cites the reference file:line whose behaviour it follows.  Paths are relative to
// the reference repository root (jczarnowski/DeepFactors).
//
// Parity pin status: PINNED to the reference's own code.  oracle/_ref (oracle/Makefile `make ref`, oracle/ref_harness.cpp) compiles the
// reference's unmodified L0 headers -- common/algorithm/{warping,dense_sfm,lucas_kanade_se3,pinhole_camera_impl,m_estimators,
// camera_pyramid}.h, from where they lie under /root/reference -- against stand-in Eigen / Sophus / VisionCore headers
// (oracle/standins/), and tests/test_oracle_vs_ref.py checks this restatement against it per pixel item and per reduced system; the
// outputs of that library on stored inputs are committed as known-answer vectors (tests/golden/ref_vectors.npz,
// tests/test_golden_ref_vectors.py), so the pin also holds where /root/reference is absent.  The same for the rows beside the step
// kernel: SparseGeometricFactor::linearize (the reference's .cpp #included whole, ref_harness_f3.cpp) and the kernel bodies of DepthAligner,
// SobelGradients, GaussianBlurDown, SquaredError and SE3Aligner::Warp, cut out of cuda/cu_depthaligner.cpp, cu_image_proc.cpp and
// cu_se3aligner.cpp at build time (oracle/Makefile, ref_harness_f1.cpp; vectors tests/golden/ref_vectors_f3.npz, ref_vectors_f1.npz).
// Beside it, the reference's own test criteria are kept as known-answer tests:
//   - tests/ut_se3aligner.cpp:173-211  ImageAlignmentTest on data/testimg/1047->1052
//     (residual/inliers <= 1e-3 after 40 Gauss-Newton iterations)        -> tests/test_oracle_kat.py
//   - tests/ut_warping.cpp:72-380, tests/ut_pinhole_camera.cpp:50-134   finite-difference checks
//   - tests/ut_sfmaligner.cpp:329-487  Jtr vs finite difference of the residual
//   - tests/ut_decoder.cpp:161-199     decoder linearity
//   - tests/ut_cuda_utils.cpp:73-144   Sobel / blur-down conventions (vs scipy.ndimage here)
// What stays "parity unpinned": the conventions of the un-vendored VisionCore / Sophus pieces (bilinear sampling, packed
// upper-triangular order, quaternion storage) are not in the reference tree; the stand-ins fix them by spec and
// tests/test_oracle_kat.py shows they are the universal ones (scipy / torch agree).
//
// Scalar type: every entry point exists for float (the reference's device Scalar) and double
// (the reference's ut_warping tests run in double).  Reductions accumulate in double ("truth")
// or in the Scalar type sequentially ("reference-like"), selected by `accum_f64`.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off -fopenmp -shared -fPIC).

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc {

// ---------------------------------------------------------------------------------------------
// small fixed-size algebra
// ---------------------------------------------------------------------------------------------
template <typename S> struct Vec3 { S x, y, z; };

template <typename S> struct Cam { S fx, fy, u0, v0, w, h; };   // pinhole_camera.h (6 scalars)

template <typename S> struct Rigid {   // a Sophus::SE3 as rotation matrix (row-major) + translation
  S R[9];
  S t[3];
};

// Rotation matrix of q = (x, y, z, w) as stored by Sophus::SO3 (Eigen::Quaternion::toRotationMatrix convention).
// Once-per-pair host algebra is done in double and rounded once to the Scalar type (the reference does it in
// fp32 through Sophus' quaternion product; exact bits of that are unpinned -- see header).  libdfx's host code
// (deepfactors_amd/csrc/dfx_api.cpp) uses the same formulas in the same order, so both sides feed identical
// fp32 rotation/translation/Jacobian constants to the per-pixel math.
template <typename S>
static void quat_to_R(const S* q, S* R) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  const double n = std::sqrt(x * x + y * y + z * z + w * w);
  if (n > 0) { x /= n; y /= n; z /= n; w /= n; }
  const double Rd[9] = { 1 - 2 * (y * y + z * z), 2 * (x * y - z * w),     2 * (x * z + y * w),
                         2 * (x * y + z * w),     1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w),     2 * (y * z + x * w),     1 - 2 * (x * x + y * y) };
  for (int i = 0; i < 9; ++i) R[i] = S(Rd[i]);
}

template <typename S>
static Rigid<S> rigid_from_qt(const S* qt) {   // qt = qx qy qz qw tx ty tz
  Rigid<S> T;
  quat_to_R(qt, T.R);
  T.t[0] = qt[4]; T.t[1] = qt[5]; T.t[2] = qt[6];
  return T;
}

template <typename S>
static void quat_to_Rd(const S* q, double* R) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  const double n = std::sqrt(x * x + y * y + z * z + w * w);
  if (n > 0) { x /= n; y /= n; z /= n; w /= n; }
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

template <typename S>
static inline Vec3<S> rot(const S* R, const Vec3<S>& p) {
  return { R[0] * p.x + R[1] * p.y + R[2] * p.z,
           R[3] * p.x + R[4] * p.y + R[5] * p.z,
           R[6] * p.x + R[7] * p.y + R[8] * p.z };
}

// SO3::hat
template <typename S>
static inline void hat(const Vec3<S>& v, S* H) {
  H[0] = 0;    H[1] = -v.z; H[2] = v.y;
  H[3] = v.z;  H[4] = 0;    H[5] = -v.x;
  H[6] = -v.y; H[7] = v.x;  H[8] = 0;
}

// Rodrigues, Sophus SO3::exp
template <typename S>
static void so3_exp(const S* w, S* R) {
  const S th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const S th = std::sqrt(th2);
  S A, B;   // R = I + A*hat(w) + B*hat(w)^2
  if (th < S(1e-6)) { A = 1 - th2 / 6; B = S(0.5) - th2 / 24; }
  else { A = std::sin(th) / th; B = (1 - std::cos(th)) / th2; }
  S H[9]; hat(Vec3<S>{w[0], w[1], w[2]}, H);
  S H2[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    S s = 0; for (int k = 0; k < 3; ++k) s += H[i * 3 + k] * H[k * 3 + j];
    H2[i * 3 + j] = s;
  }
  for (int i = 0; i < 9; ++i) R[i] = A * H[i] + B * H2[i];
  R[0] += 1; R[4] += 1; R[8] += 1;
}

// ---------------------------------------------------------------------------------------------
// warping.h:98-137  RelativePose(pose_a, pose_b, jac_a, jac_b):  T_ab = a^-1 * b
//   jac_a = [[-Ra^T, -hat(Ra^T (ta - tb)) Ra^T], [0, -Ra^T]],  jac_b = blkdiag(Ra^T, Ra^T)
// Tangent order (tx,ty,tz,wx,wy,wz); 6x6 row-major.
// ---------------------------------------------------------------------------------------------
// a_qt, b_qt are the (qx qy qz qw tx ty tz) poses; all algebra in double, results rounded once.
template <typename S>
static Rigid<S> relative_pose(const S* a_qt, const S* b_qt, S* jac_a, S* jac_b) {
  double Ra[9], Rb[9];
  quat_to_Rd(a_qt, Ra);
  quat_to_Rd(b_qt, Rb);
  double Rat[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rat[i * 3 + j] = Ra[j * 3 + i];
  Rigid<S> ab;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    double s = 0; for (int k = 0; k < 3; ++k) s += Rat[i * 3 + k] * Rb[k * 3 + j];
    ab.R[i * 3 + j] = S(s);
  }
  const double d_ba[3] = { double(b_qt[4]) - double(a_qt[4]), double(b_qt[5]) - double(a_qt[5]), double(b_qt[6]) - double(a_qt[6]) };
  double tab[3];
  for (int i = 0; i < 3; ++i) {
    tab[i] = Rat[i * 3] * d_ba[0] + Rat[i * 3 + 1] * d_ba[1] + Rat[i * 3 + 2] * d_ba[2];
    ab.t[i] = S(tab[i]);
  }
  if (jac_a && jac_b) {
    const double v[3] = { -tab[0], -tab[1], -tab[2] };   // Ra^T (ta - tb)
    const double H[9] = { 0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0 };
    double HR[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
      double s = 0; for (int k = 0; k < 3; ++k) s += H[i * 3 + k] * Rat[k * 3 + j];
      HR[i * 3 + j] = s;
    }
    std::fill(jac_a, jac_a + 36, S(0));
    std::fill(jac_b, jac_b + 36, S(0));
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
      jac_a[i * 6 + j] = -S(Rat[i * 3 + j]);
      jac_a[i * 6 + 3 + j] = -S(HR[i * 3 + j]);
      jac_a[(3 + i) * 6 + 3 + j] = -S(Rat[i * 3 + j]);
      jac_b[i * 6 + j] = S(Rat[i * 3 + j]);
      jac_b[(3 + i) * 6 + 3 + j] = S(Rat[i * 3 + j]);
    }
  }
  return ab;
}

// ---------------------------------------------------------------------------------------------
// pitched image views (VisionCore Buffer2DView semantics: ptr + y*pitch_bytes, then [x])
// ---------------------------------------------------------------------------------------------
template <typename S>
struct Img {
  const S* ptr; std::size_t pitch; int w, h;   // pitch in bytes
  inline const S* row(int y) const { return reinterpret_cast<const S*>(reinterpret_cast<const char*>(ptr) + std::size_t(y) * pitch); }
  inline S at(int x, int y) const { return row(y)[x]; }
};
template <typename S>
struct ImgW {
  S* ptr; std::size_t pitch; int w, h;
  inline S* row(int y) const { return reinterpret_cast<S*>(reinterpret_cast<char*>(ptr) + std::size_t(y) * pitch); }
};

// VisionCore getBilinear (not in the reference tree -> convention fixed by spec):
//   ix=floor(u), iy=floor(v), lerp(lerp(I[iy][ix],I[iy][ix+1],fx), lerp(I[iy+1][ix],I[iy+1][ix+1],fx), fy)
//   with lerp(a,b,t)=a+t(b-a).  `nch` interleaved channels (1 = image, 2 = gradient).
template <typename S, int NCH>
static inline void bilinear(const Img<S>& I, S u, S v, S* out) {
  const S fu = std::floor(u), fv = std::floor(v);
  const int ix = int(fu), iy = int(fv);
  const S ax = u - fu, ay = v - fv;
  const S* r0 = I.row(iy) + std::size_t(ix) * NCH;
  const S* r1 = I.row(iy + 1) + std::size_t(ix) * NCH;
  for (int c = 0; c < NCH; ++c) {
    const S top = r0[c] + ax * (r0[NCH + c] - r0[c]);
    const S bot = r1[c] + ax * (r1[NCH + c] - r1[c]);
    out[c] = top + ay * (bot - top);
  }
}

// m_estimators.h:50-56  HuberWeight: sqrt-weight applied to both J and r
template <typename S>
static inline S huber_weight(S x, S delta) {
  const S aa = std::fabs(x);
  return aa <= delta ? S(1) : std::sqrt(delta * (2 * aa - delta)) / aa;
}

// warping.h:30-50
template <typename S> static inline S prox_to_depth(S prx, S a) { return a / prx - a; }
template <typename S> static inline S depth_to_prox(S d, S a) { return a / (a + d); }
template <typename S> static inline S depth_jacobian_prx(S d, S a) {
  const S prx = a / (a + d);
  return -a / (prx * prx);
}

// ---------------------------------------------------------------------------------------------
// warping.h:204-241 FindCorrespondence + pinhole_camera_impl.h:41-56,105-108
// ---------------------------------------------------------------------------------------------
template <typename S>
struct Corresp {
  Vec3<S> ray;    // ((x-u0)/fx, (y-v0)/fy, 1)   = ReprojectDepthJacobian
  Vec3<S> pt;     // ray * dpt
  Vec3<S> tpt;    // R pt + t
  S u, v;         // pix1
  bool valid;
};

template <typename S>
static inline Corresp<S> find_correspondence(int x, int y, S dpt, const Cam<S>& cam, const Rigid<S>& T,
                                             int border, S min_dpt) {
  Corresp<S> c;
  c.valid = false;
  c.ray = { (S(x) - cam.u0) / cam.fx, (S(y) - cam.v0) / cam.fy, S(1) };
  c.pt = { c.ray.x * dpt, c.ray.y * dpt, c.ray.z * dpt };
  const Vec3<S> rp = rot(T.R, c.pt);
  c.tpt = { rp.x + T.t[0], rp.y + T.t[1], rp.z + T.t[2] };
  c.u = c.v = 0;
  if (c.tpt.z > min_dpt) {
    c.u = cam.fx * c.tpt.x / c.tpt.z + cam.u0;
    c.v = cam.fy * c.tpt.y / c.tpt.z + cam.v0;
    const S b = S(border);
    c.valid = (c.u >= b) && (c.u < cam.w - b) && (c.v >= b) && (c.v < cam.h - b);
  }
  return c;
}

// warping.h:247-257 FindCorrespondenceJacobianPose = ProjectPointJacobian(tpt) * [I | -hat(R pt)]
// (note: R pt WITHOUT the translation, warping.h:162).  2x6 row-major.
template <typename S>
static inline void corresp_jac_pose(const Corresp<S>& c, const Cam<S>& cam, const Rigid<S>& T, S* C /*12*/, S* D /*6*/) {
  const S X = c.tpt.x, Y = c.tpt.y, Z = c.tpt.z;
  D[0] = cam.fx / Z; D[1] = 0;          D[2] = -(cam.fx * X) / Z / Z;
  D[3] = 0;          D[4] = cam.fy / Z; D[5] = -(cam.fy * Y) / Z / Z;
  const Vec3<S> rp = rot(T.R, c.pt);
  S H[9]; hat(rp, H);
  for (int r = 0; r < 2; ++r) {
    for (int j = 0; j < 3; ++j) C[r * 6 + j] = D[r * 3 + j];
    for (int j = 0; j < 3; ++j) {
      S s = 0; for (int k = 0; k < 3; ++k) s += D[r * 3 + k] * (-H[k * 3 + j]);
      C[r * 6 + 3 + j] = s;
    }
  }
}

// warping.h:259-291: d pix1 / d prx = D * R * ray * (-a/prx^2)
template <typename S>
static inline void corresp_jac_prx(const Corresp<S>& c, const S* D, const Rigid<S>& T, S dpt, S avg_dpt, S* out2) {
  const Vec3<S> rr = rot(T.R, c.ray);
  const S dprx = depth_jacobian_prx(dpt, avg_dpt);
  out2[0] = (D[0] * rr.x + D[1] * rr.y + D[2] * rr.z) * dprx;
  out2[1] = (D[3] * rr.x + D[4] * rr.y + D[5] * rr.z) * dprx;
}

// ---------------------------------------------------------------------------------------------
// reduction payloads (reduction_items.h:35-143).  JtJ is the packed upper triangle, row-major
// ((0,0),(0,1)..(0,N-1),(1,1)..) -- VisionCore SquareUpperTriangularMatrix order (by spec).
// ---------------------------------------------------------------------------------------------
template <typename A>
struct Accum {
  int np;
  std::vector<A> jtj, jtr;
  A residual = 0;
  std::uint64_t inliers = 0;
  explicit Accum(int n) : np(n), jtj(std::size_t(n) * (n + 1) / 2, A(0)), jtr(n, A(0)) {}
  template <typename S>
  inline void add(const S* J, S r) {
    inliers += 1;
    residual += A(r * r);
    std::size_t k = 0;
    for (int i = 0; i < np; ++i) {
      jtr[i] += A(J[i] * r);
      for (int j = i; j < np; ++j) jtj[k++] += A(J[i] * J[j]);
    }
  }
  void merge(const Accum& o) {
    for (std::size_t i = 0; i < jtj.size(); ++i) jtj[i] += o.jtj[i];
    for (int i = 0; i < np; ++i) jtr[i] += o.jtr[i];
    residual += o.residual; inliers += o.inliers;
  }
};

// lucas_kanade_se3.h:41-77 per-pixel item; returns false when the pixel is not an inlier.
template <typename S>
static inline bool se3_item(int x, int y, const Rigid<S>& T, const Cam<S>& cam, const Img<S>& img0, const Img<S>& img1,
                            const Img<S>& dpt0, const Img<S>& grad1, S huber_delta, S* J /*6*/, S* r) {
  const S d = dpt0.at(x, y);
  const Corresp<S> c = find_correspondence(x, y, d, cam, T, 1, S(0));
  if (!c.valid) return false;
  S C[12], D[6];
  corresp_jac_pose(c, cam, T, C, D);
  S g[2]; bilinear<S, 2>(grad1, c.u, c.v, g);
  for (int j = 0; j < 6; ++j) J[j] = (-g[0]) * C[j] + (-g[1]) * C[6 + j];
  S samp; bilinear<S, 1>(img1, c.u, c.v, &samp);
  S diff = img0.at(x, y) - samp;
  const S w = huber_weight(diff, huber_delta);
  diff *= w;
  for (int j = 0; j < 6; ++j) J[j] *= w;
  *r = diff;
  return true;
}

// dense_sfm.h:133-201 per-pixel item.  J = [dE/dpose0 (6), dE/dpose1 (6), dE/dcode0 (CS)]
template <typename S>
static inline bool sfm_item(int x, int y, int cs, const Rigid<S>& T10, const S* J0 /*pose10_J_pose0*/, const S* J1,
                            const Cam<S>& cam, const Img<S>& img0, const Img<S>& img1, const Img<S>& dpt0,
                            const Img<S>& jac, const Img<S>& grad1, S huber_delta, S avg_dpt, S min_dpt, int border,
                            S* J, S* r) {
  const S d = dpt0.at(x, y);
  const Corresp<S> c = find_correspondence(x, y, d, cam, T10, border, min_dpt);
  if (!c.valid) return false;
  S C[12], D[6];
  corresp_jac_pose(c, cam, T10, C, D);
  S g[2]; bilinear<S, 2>(grad1, c.u, c.v, g);
  S gC[6];
  for (int j = 0; j < 6; ++j) gC[j] = (-g[0]) * C[j] + (-g[1]) * C[6 + j];
  for (int j = 0; j < 6; ++j) {
    S s0 = 0, s1 = 0;
    for (int k = 0; k < 6; ++k) { s0 += gC[k] * J0[k * 6 + j]; s1 += gC[k] * J1[k * 6 + j]; }
    J[j] = s0; J[6 + j] = s1;
  }
  S pj[2]; corresp_jac_prx(c, D, T10, d, avg_dpt, pj);
  const S e = -(g[0] * pj[0] + g[1] * pj[1]);
  const S* jrow = jac.row(y) + std::size_t(x) * cs;
  for (int k = 0; k < cs; ++k) J[12 + k] = e * jrow[k];
  S samp; bilinear<S, 1>(img1, c.u, c.v, &samp);
  S diff = img0.at(x, y) - samp;
  // dense_sfm.h:58-67: the uncertainty weight is computed and then discarded (returns 1.0)
  const S w = huber_weight(diff, huber_delta) * S(1);
  for (int k = 0; k < 12 + cs; ++k) J[k] *= w;
  *r = diff * w;
  return true;
}

// dense_sfm.h:79-119 error item (border 1, min_dpt 0: FindCorrespondence defaults)
template <typename S>
static inline bool sfm_error_item(int x, int y, const Rigid<S>& T10, const Cam<S>& cam, const Img<S>& img0,
                                  const Img<S>& img1, const Img<S>& dpt0, S huber_delta, S* r) {
  const Corresp<S> c = find_correspondence(x, y, dpt0.at(x, y), cam, T10, 1, S(0));
  if (!c.valid) return false;
  S samp; bilinear<S, 1>(img1, c.u, c.v, &samp);
  S diff = img0.at(x, y) - samp;
  diff *= huber_weight(diff, huber_delta);
  *r = diff;
  return true;
}

template <typename S, typename A>
static void export_accum(const Accum<A>& acc, S* jtj, S* jtr, S* residual, std::uint64_t* inliers) {
  for (std::size_t i = 0; i < acc.jtj.size(); ++i) jtj[i] = S(acc.jtj[i]);
  for (int i = 0; i < acc.np; ++i) jtr[i] = S(acc.jtr[i]);
  *residual = S(acc.residual);
  *inliers = acc.inliers;
}

// Generic row-parallel reduction driver: f(x, y, J, &r) -> bool
template <typename S, typename A, typename F>
static void reduce_rows(int w, int h, int np, int threads, F&& f, Accum<A>& total) {
#ifdef _OPENMP
  if (threads > 1) {
    std::vector<Accum<A>> parts(threads, Accum<A>(np));
#pragma omp parallel num_threads(threads)
    {
      const int tid = omp_get_thread_num();
      std::vector<S> J(np);
      S r;
#pragma omp for schedule(static)
      for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
          if (f(x, y, J.data(), &r)) parts[tid].add(J.data(), r);
    }
    for (auto& p : parts) total.merge(p);
    return;
  }
#endif
  std::vector<S> J(np);
  S r;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      if (f(x, y, J.data(), &r)) total.add(J.data(), r);
}

template <typename S>
static Cam<S> cam_from(const S* c) { return { c[0], c[1], c[2], c[3], c[4], c[5] }; }

// cu_se3aligner.cpp:153-176
template <typename S>
static void se3_step(const S* pose_qt, const S* camv, const S* img0, const S* img1, const S* dpt0, const S* grad1,
                     int w, int h, std::size_t pitch, std::size_t gpitch, S huber_delta, int accum_f64, int threads,
                     S* jtj, S* jtr, S* residual, std::uint64_t* inliers) {
  const Rigid<S> T = rigid_from_qt(pose_qt);
  const Cam<S> cam = cam_from(camv);
  const Img<S> I0{ img0, pitch, w, h }, I1{ img1, pitch, w, h }, D0{ dpt0, pitch, w, h }, G1{ grad1, gpitch, w, h };
  auto f = [&](int x, int y, S* J, S* r) { return se3_item(x, y, T, cam, I0, I1, D0, G1, huber_delta, J, r); };
  if (accum_f64) { Accum<double> a(6); reduce_rows<S, double>(w, h, 6, threads, f, a); export_accum(a, jtj, jtr, residual, inliers); }
  else { Accum<S> a(6); reduce_rows<S, S>(w, h, 6, threads, f, a); export_accum(a, jtj, jtr, residual, inliers); }
}

// cu_sfmaligner.cpp:149-185 (host RelativePose(pose1, pose0, J1, J0) then the per-pixel sweep)
template <typename S>
static void sfm_step(const S* pose0_qt, const S* pose1_qt, const S* camv, int cs, const S* img0, const S* img1,
                     const S* dpt0, const S* jac, const S* grad1, S* valid0, int w, int h, std::size_t pitch,
                     std::size_t jpitch, std::size_t gpitch, S huber_delta, S avg_dpt, S min_dpt, int border,
                     int accum_f64, int threads, S* jtj, S* jtr, S* residual, std::uint64_t* inliers) {
  S J1[36], J0[36];
  const Rigid<S> T10 = relative_pose(pose1_qt, pose0_qt, J1, J0);   // jac_a -> pose1, jac_b -> pose0
  const Cam<S> cam = cam_from(camv);
  const int np = 12 + cs;
  const Img<S> I0{ img0, pitch, w, h }, I1{ img1, pitch, w, h }, D0{ dpt0, pitch, w, h };
  const Img<S> JAC{ jac, jpitch, w * cs, h }, G1{ grad1, gpitch, w, h };
  ImgW<S> V{ valid0, pitch, w, h };
  auto f = [&](int x, int y, S* J, S* r) {
    const bool ok = sfm_item(x, y, cs, T10, J0, J1, cam, I0, I1, D0, JAC, G1, huber_delta, avg_dpt, min_dpt, border, J, r);
    if (ok && valid0) V.row(y)[x] = S(1);   // dense_sfm.h:161: only ever set, never cleared
    return ok;
  };
  if (accum_f64) { Accum<double> a(np); reduce_rows<S, double>(w, h, np, threads, f, a); export_accum(a, jtj, jtr, residual, inliers); }
  else { Accum<S> a(np); reduce_rows<S, S>(w, h, np, threads, f, a); export_accum(a, jtj, jtr, residual, inliers); }
}

// cu_sfmaligner.cpp:120-147
template <typename S>
static void sfm_error(const S* pose0_qt, const S* pose1_qt, const S* camv, const S* img0, const S* img1, const S* dpt0,
                      int w, int h, std::size_t pitch, S huber_delta, int accum_f64, S* residual, std::uint64_t* inliers) {
  const Rigid<S> T10 = relative_pose<S>(pose1_qt, pose0_qt, nullptr, nullptr);
  const Cam<S> cam = cam_from(camv);
  const Img<S> I0{ img0, pitch, w, h }, I1{ img1, pitch, w, h }, D0{ dpt0, pitch, w, h };
  double accd = 0; S accs = 0; std::uint64_t n = 0;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      S r;
      if (sfm_error_item(x, y, T10, cam, I0, I1, D0, huber_delta, &r)) { accd += double(r * r); accs += r * r; ++n; }
    }
  *residual = accum_f64 ? S(accd) : accs;
  *inliers = n;
}

// cu_se3aligner.cpp:61-113 kernel_warp_calculate: renders img1 into frame 0, SIGNED residual sum
template <typename S>
static void se3_warp(const S* pose_qt, const S* camv, const S* img0, const S* img1, const S* dpt0, S* img2, int w, int h,
                     std::size_t pitch, int accum_f64, S* residual, std::uint64_t* inliers) {
  const Rigid<S> T = rigid_from_qt(pose_qt);
  const Cam<S> cam = cam_from(camv);
  const Img<S> I0{ img0, pitch, w, h }, I1{ img1, pitch, w, h }, D0{ dpt0, pitch, w, h };
  ImgW<S> O{ img2, pitch, w, h };
  double accd = 0; S accs = 0; std::uint64_t n = 0;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      O.row(y)[x] = 0;
      const Vec3<S> ray{ (S(x) - cam.u0) / cam.fx, (S(y) - cam.v0) / cam.fy, S(1) };
      const S d = D0.at(x, y);
      const Vec3<S> pt{ ray.x * d, ray.y * d, ray.z * d };
      const Vec3<S> rp = rot(T.R, pt);
      const Vec3<S> q{ rp.x + T.t[0], rp.y + T.t[1], rp.z + T.t[2] };
      if (!(q.z > 0)) continue;   // `if (depth <= 0) return;`  (NaN falls through in the reference; see note)
      const S u = cam.fx * q.x / q.z + cam.u0, v = cam.fy * q.y / q.z + cam.v0;
      if (u >= 1 && u < cam.w - 1 && v >= 1 && v < cam.h - 1) {
        S samp; bilinear<S, 1>(I1, u, v, &samp);
        O.row(y)[x] = samp;
        const S e = I0.at(x, y) - samp;
        accd += double(e); accs += e; ++n;
      }
    }
  *residual = accum_f64 ? S(accd) : accs;
  *inliers = n;
}

// cu_image_proc.cpp:248-277 + warping.h:52-69: dpt = a/(prx0 + j.c) - a.  Sequential dot product.
template <typename S>
static void update_depth(int cs, const S* code, const S* prx_orig, const S* jac, S avg_dpt, S* dpt_out, int w, int h,
                         std::size_t pitch, std::size_t jpitch) {
  const Img<S> P{ prx_orig, pitch, w, h }, JAC{ jac, jpitch, w * cs, h };
  ImgW<S> O{ dpt_out, pitch, w, h };
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const S* j = JAC.row(y) + std::size_t(x) * cs;
      S dot = 0;
      for (int k = 0; k < cs; ++k) dot += j[k] * code[k];
      O.row(y)[x] = prox_to_depth(P.at(x, y) + dot, avg_dpt);
    }
}

// cu_depthaligner.cpp:32-72 (avg_dpt hard-coded to 2 there; parameter here, callers pass 2)
template <typename S>
static void depth_aligner_step(int cs, const S* code, const S* tgt, const S* prx_orig, const S* jac, S avg_dpt, int w, int h,
                               std::size_t pitch, std::size_t jpitch, int accum_f64, S* jtj, S* jtr, S* residual,
                               std::uint64_t* inliers) {
  const Img<S> T{ tgt, pitch, w, h }, P{ prx_orig, pitch, w, h }, JAC{ jac, jpitch, w * cs, h };
  auto f = [&](int x, int y, S* J, S* r) {
    const S* j = JAC.row(y) + std::size_t(x) * cs;
    S dot = 0;
    for (int k = 0; k < cs; ++k) dot += j[k] * code[k];
    const S dpt = prox_to_depth(P.at(x, y) + dot, avg_dpt);
    const S diff = T.at(x, y) - dpt;
    const S s = -2 * std::fabs(diff) * depth_jacobian_prx(dpt, avg_dpt);
    for (int k = 0; k < cs; ++k) J[k] = s * j[k];
    *r = diff;
    return true;
  };
  if (accum_f64) { Accum<double> a(cs); reduce_rows<S, double>(w, h, cs, 1, f, a); export_accum(a, jtj, jtr, residual, inliers); }
  else { Accum<S> a(cs); reduce_rows<S, S>(w, h, cs, 1, f, a); export_accum(a, jtj, jtr, residual, inliers); }
}

// sparse_geometric_factor.cpp:147-275 SparseGeometricFactor::linearize -- one Jacobian row per sampled point:
//   row = w * [err_J_pose0 (6) | err_J_pose1 (6) | err_J_cde0 (CS) | err_J_cde1 (CS) | err],  err = dpt1 - (R p + t).z
// with dpt1 decoded at the NEAREST-NEIGHBOUR pixel of the projection (truncation, :207) and kf1's depth gradient sampled
// there (:220).  Invalid points (FindCorrespondence defaults: border 1, min_dpt 0) give an all-zero row (:190-198).
// The signs are the reference's: every Jacobian block is the negative of d(err)/dx, i.e. the Jacobian of
// (dpt1_p - dpt1), and the last column is b = err -- consistent with a gtsam::JacobianFactor ||A x - b||^2.
template <typename S>
static void sparse_geometric(const S* pose0_qt, const S* pose1_qt, const S* code0, const S* code1, const S* camv, int cs,
                             const int* pts_xy, int npts, const S* prx0, const S* jac0, const S* prx1, const S* jac1,
                             const S* dgrad1, int w, int h, std::size_t pitch, std::size_t jpitch, std::size_t gpitch, S huber_delta,
                             S avg_dpt, S* rows /* [npts][12 + 2cs + 1] */) {
  S J1[36], J0[36];
  const Rigid<S> T = relative_pose(pose1_qt, pose0_qt, J1, J0);
  const Cam<S> cam = cam_from(camv);
  const Img<S> P0{ prx0, pitch, w, h }, P1{ prx1, pitch, w, h }, JA0{ jac0, jpitch, w * cs, h }, JA1{ jac1, jpitch, w * cs, h };
  const Img<S> DG{ dgrad1, gpitch, w, h };
  const int nc = 12 + 2 * cs + 1;
  for (int i = 0; i < npts; ++i) {
    S* row = rows + std::size_t(i) * nc;
    for (int k = 0; k < nc; ++k) row[k] = 0;
    const int x = pts_xy[2 * i], y = pts_xy[2 * i + 1];
    const S* j0 = JA0.row(y) + std::size_t(x) * cs;
    S dot0 = 0;
    for (int k = 0; k < cs; ++k) dot0 += j0[k] * code0[k];
    const S d0 = prox_to_depth(P0.at(x, y) + dot0, avg_dpt);
    const Corresp<S> c = find_correspondence(x, y, d0, cam, T, 1, S(0));
    if (!c.valid) continue;
    const int nx = int(c.u), ny = int(c.v);   // cast<int>: truncation
    const S* j1 = JA1.row(ny) + std::size_t(nx) * cs;
    S dot1 = 0;
    for (int k = 0; k < cs; ++k) dot1 += j1[k] * code1[k];
    const S d1 = prox_to_depth(P1.at(nx, ny) + dot1, avg_dpt);
    S err = d1 - c.tpt.z;
    const S gx = DG.row(ny)[2 * nx], gy = DG.row(ny)[2 * nx + 1];
    S C[12], D[6];
    corresp_jac_pose(c, cam, T, C, D);
    // third row of [I | -hat(R p)]
    const Vec3<S> rp = rot(T.R, c.pt);
    const S dz[6] = { 0, 0, 1, -rp.y, rp.x, 0 };   // -hat(v) row 2 = (v.y, -v.x, 0) ... see below
    // -hat(v) = [[0, v.z, -v.y], [-v.z, 0, v.x], [v.y, -v.x, 0]]
    S dzr[6] = { 0, 0, 1, rp.y, -rp.x, 0 };
    (void)dz;
    S a10[6];   // dpt1p_J_pose10 - dpt_grad * corr_J_pose10   (1x6, w.r.t. pose10)
    for (int k = 0; k < 6; ++k) a10[k] = dzr[k] - (gx * C[k] + gy * C[6 + k]);
    S e0[6], e1[6];
    for (int j = 0; j < 6; ++j) {
      S s0 = 0, s1 = 0;
      for (int k = 0; k < 6; ++k) { s0 += a10[k] * J0[k * 6 + j]; s1 += a10[k] * J1[k * 6 + j]; }
      e0[j] = s0; e1[j] = s1;
    }
    const Vec3<S> rr = rot(T.R, c.ray);
    const S dprx0 = depth_jacobian_prx(d0, avg_dpt);
    const S pj0 = D[0] * rr.x + D[1] * rr.y + D[2] * rr.z, pj1 = D[3] * rr.x + D[4] * rr.y + D[5] * rr.z;
    const S sc0 = (rr.z - (gx * pj0 + gy * pj1)) * dprx0;
    const S sc1 = -depth_jacobian_prx(d1, avg_dpt);
    const S wgt = huber_weight(err, huber_delta);
    for (int j = 0; j < 6; ++j) { row[j] = e0[j] * wgt; row[6 + j] = e1[j] * wgt; }
    for (int k = 0; k < cs; ++k) { row[12 + k] = sc0 * j0[k] * wgt; row[12 + cs + k] = sc1 * j1[k] * wgt; }
    row[12 + 2 * cs] = err * wgt;
  }
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// cu_image_proc.cpp:34-92 Sobel /8, clamped borders; out = (gx,gy) interleaved
template <typename S>
static void sobel(const S* img, S* grad, int w, int h, std::size_t pitch, std::size_t gpitch) {
  static const S KX[3][3] = { { -1, 0, 1 }, { -2, 0, 2 }, { -1, 0, 1 } };
  static const S KY[3][3] = { { -1, -2, -1 }, { 0, 0, 0 }, { 1, 2, 1 } };
  const Img<S> I{ img, pitch, w, h };
  ImgW<S> G{ grad, gpitch, w, h };
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      S sx = 0, sy = 0;
      for (int py = -1; py <= 1; ++py)
        for (int px = -1; px <= 1; ++px) {
          const S pix = I.at(clampi(x + px, 0, w - 1), clampi(y + py, 0, h - 1));
          sx += pix * KX[py + 1][px + 1];
          sy += pix * KY[py + 1][px + 1];
        }
      G.row(y)[2 * x] = sx / 8;
      G.row(y)[2 * x + 1] = sy / 8;
    }
}

// cu_image_proc.cpp:119-164 5x5 binomial blur + decimate by 2, taps clamped, normalised by sum of weights
template <typename S>
static void blur_down(const S* in, S* out, int w, int h, std::size_t pitch, int ow, int oh, std::size_t opitch) {
  static const S B[5] = { 1, 4, 6, 4, 1 };
  const Img<S> I{ in, pitch, w, h };
  ImgW<S> O{ out, opitch, ow, oh };
  for (int y = 0; y < oh; ++y)
    for (int x = 0; x < ow; ++x) {
      S sum = 0, wall = 0;
      for (int py = 0; py < 5; ++py)
        for (int px = 0; px < 5; ++px) {
          const int nx = clampi(2 * x + px - 2, 0, w - 1), ny = clampi(2 * y + py - 2, 0, h - 1);
          const S k = B[px] * B[py];
          sum += I.at(nx, ny) * k;
          wall += k;
        }
      O.row(y)[x] = sum / wall;
    }
}

// cu_image_proc.cpp:190-227
template <typename S>
static S squared_error(const S* a, const S* b, int w, int h, std::size_t pitch, int accum_f64) {
  const Img<S> A{ a, pitch, w, h }, Bm{ b, pitch, w, h };
  double accd = 0; S accs = 0;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) { const S d = A.at(x, y) - Bm.at(x, y); accd += double(d * d); accs += d * d; }
  return accum_f64 ? S(accd) : accs;
}

// lucas_kanade_se3.h:85-95 / camera_tracker.cpp:59-63: update = -LDLT(JtJ)^-1 Jtr; t += dt; R = exp(dw) R
// JtJ is the packed upper triangle (21).  Pose in/out as qt (quaternion renormalised from the matrix).
template <typename S>
static void R_to_quat(const S* R, S* q) {
  const S tr = R[0] + R[4] + R[8];
  S x, y, z, w;
  if (tr > 0) { S s = std::sqrt(tr + 1) * 2; w = s / 4; x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s; }
  else if (R[0] > R[4] && R[0] > R[8]) { S s = std::sqrt(1 + R[0] - R[4] - R[8]) * 2; w = (R[7] - R[5]) / s; x = s / 4; y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s; }
  else if (R[4] > R[8]) { S s = std::sqrt(1 + R[4] - R[0] - R[8]) * 2; w = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; y = s / 4; z = (R[5] + R[7]) / s; }
  else { S s = std::sqrt(1 + R[8] - R[0] - R[4]) * 2; w = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; z = s / 4; }
  const S n = std::sqrt(x * x + y * y + z * z + w * w);
  q[0] = x / n; q[1] = y / n; q[2] = z / n; q[3] = w / n;
}

template <typename S>
static int ldlt_solve(int n, const S* Aupper_packed, const S* b, S* xout) {
  std::vector<double> A(std::size_t(n) * n), L(std::size_t(n) * n, 0.0), Dg(n), yv(n), zv(n);
  std::size_t k = 0;
  for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) { A[i * n + j] = A[j * n + i] = double(Aupper_packed[k++]); }
  for (int j = 0; j < n; ++j) {
    double d = A[j * n + j];
    for (int p = 0; p < j; ++p) d -= L[j * n + p] * L[j * n + p] * Dg[p];
    Dg[j] = d;
    if (d == 0.0) return 1;
    L[j * n + j] = 1.0;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i * n + j];
      for (int p = 0; p < j; ++p) s -= L[i * n + p] * L[j * n + p] * Dg[p];
      L[i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; ++i) { double s = double(b[i]); for (int p = 0; p < i; ++p) s -= L[i * n + p] * yv[p]; yv[i] = s; }
  for (int i = 0; i < n; ++i) zv[i] = yv[i] / Dg[i];
  for (int i = n - 1; i >= 0; --i) { double s = zv[i]; for (int p = i + 1; p < n; ++p) s -= L[p * n + i] * xout[p]; xout[i] = S(s); }
  return 0;
}

template <typename S>
static int se3_solve_update(const S* jtj21, const S* jtr6, S* pose_qt) {
  S upd[6];
  if (ldlt_solve(6, jtj21, jtr6, upd)) return 1;
  for (int i = 0; i < 6; ++i) upd[i] = -upd[i];
  Rigid<S> T = rigid_from_qt(pose_qt);
  S E[9]; so3_exp(upd + 3, E);
  S Rn[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    S s = 0; for (int k2 = 0; k2 < 3; ++k2) s += E[i * 3 + k2] * T.R[k2 * 3 + j];
    Rn[i * 3 + j] = s;
  }
  R_to_quat(Rn, pose_qt);
  pose_qt[4] += upd[0]; pose_qt[5] += upd[1]; pose_qt[6] += upd[2];
  return 0;
}

// testing_utils.h:73-88 GetPerturbedPose: translation additive, rotation left-multiplied
template <typename S>
static void perturb_pose(const S* qt_in, int idx, S eps, S* qt_out) {
  std::memcpy(qt_out, qt_in, 7 * sizeof(S));
  if (idx < 3) { qt_out[4 + idx] += eps; return; }
  S w[3] = { 0, 0, 0 }; w[idx - 3] = eps;
  S E[9]; so3_exp(w, E);
  Rigid<S> T = rigid_from_qt(qt_in);
  S Rn[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    S s = 0; for (int k = 0; k < 3; ++k) s += E[i * 3 + k] * T.R[k * 3 + j];
    Rn[i * 3 + j] = s;
  }
  R_to_quat(Rn, qt_out);
}

}  // namespace orc

// ---------------------------------------------------------------------------------------------
// C ABI (ctypes).  Suffix _f32 / _f64 selects the Scalar.
// ---------------------------------------------------------------------------------------------
#define ORC_API extern "C" __attribute__((visibility("default")))

#define ORC_DEFINE(SFX, S)                                                                                              \
  ORC_API void orc_relative_pose_##SFX(const S* a_qt, const S* b_qt, S* R9, S* t3, S* jac_a36, S* jac_b36) {            \
    const orc::Rigid<S> AB = orc::relative_pose(a_qt, b_qt, jac_a36, jac_b36);                                          \
    std::memcpy(R9, AB.R, sizeof(AB.R)); std::memcpy(t3, AB.t, sizeof(AB.t));                                           \
  }                                                                                                                     \
  ORC_API void orc_quat_to_R_##SFX(const S* q, S* R9) { orc::quat_to_R(q, R9); }                                        \
  ORC_API void orc_so3_exp_##SFX(const S* w3, S* R9) { orc::so3_exp(w3, R9); }                                          \
  ORC_API void orc_perturb_pose_##SFX(const S* qt, int idx, S eps, S* out) { orc::perturb_pose(qt, idx, eps, out); }    \
  ORC_API int orc_se3_solve_update_##SFX(const S* jtj21, const S* jtr6, S* qt) { return orc::se3_solve_update(jtj21, jtr6, qt); } \
  ORC_API int orc_ldlt_solve_##SFX(int n, const S* upper, const S* b, S* x) { return orc::ldlt_solve(n, upper, b, x); } \
  /* single-pixel probes used by the finite-difference tests (ut_warping / ut_pinhole_camera) */                        \
  ORC_API int orc_correspondence_##SFX(int x, int y, S dpt, const S* camv, const S* qt, int border, S min_dpt, S avg_dpt,\
                                       S* pix1 /*2*/, S* tpt /*3*/, S* jac_pose /*12*/, S* jac_dpt /*2*/, S* jac_prx /*2*/) { \
    const orc::Rigid<S> T = orc::rigid_from_qt(qt);                                                                     \
    const orc::Cam<S> cam = orc::cam_from(camv);                                                                        \
    const orc::Corresp<S> c = orc::find_correspondence(x, y, dpt, cam, T, border, min_dpt);                             \
    pix1[0] = c.u; pix1[1] = c.v; tpt[0] = c.tpt.x; tpt[1] = c.tpt.y; tpt[2] = c.tpt.z;                                 \
    S D[6];                                                                                                             \
    orc::corresp_jac_pose(c, cam, T, jac_pose, D);                                                                      \
    orc::corresp_jac_prx(c, D, T, dpt, avg_dpt, jac_prx);                                                               \
    const S dprx = orc::depth_jacobian_prx(dpt, avg_dpt);                                                               \
    jac_dpt[0] = jac_prx[0] / dprx; jac_dpt[1] = jac_prx[1] / dprx;                                                     \
    return c.valid ? 1 : 0;                                                                                             \
  }                                                                                                                     \
  ORC_API S orc_huber_weight_##SFX(S x, S d) { return orc::huber_weight(x, d); }                                        \
  ORC_API S orc_depth_jacobian_prx_##SFX(S d, S a) { return orc::depth_jacobian_prx(d, a); }                            \
  ORC_API S orc_prox_to_depth_##SFX(S p, S a) { return orc::prox_to_depth(p, a); }                                      \
  ORC_API S orc_depth_to_prox_##SFX(S d, S a) { return orc::depth_to_prox(d, a); }                                      \
  ORC_API void orc_bilinear_##SFX(const S* img, int w, int h, size_t pitch, int nch, S u, S v, S* out) {                \
    const orc::Img<S> I{ img, pitch, w, h };                                                                            \
    if (nch == 2) orc::bilinear<S, 2>(I, u, v, out); else orc::bilinear<S, 1>(I, u, v, out);                            \
  }                                                                                                                     \
  ORC_API void orc_se3_step_##SFX(const S* qt, const S* cam, const S* img0, const S* img1, const S* dpt0, const S* grad1,\
                                  int w, int h, size_t pitch, size_t gpitch, S huber, int accum_f64, int threads,       \
                                  S* jtj, S* jtr, S* residual, uint64_t* inliers) {                                     \
    orc::se3_step(qt, cam, img0, img1, dpt0, grad1, w, h, pitch, gpitch, huber, accum_f64, threads, jtj, jtr, residual, inliers); \
  }                                                                                                                     \
  ORC_API void orc_se3_warp_##SFX(const S* qt, const S* cam, const S* img0, const S* img1, const S* dpt0, S* img2,      \
                                  int w, int h, size_t pitch, int accum_f64, S* residual, uint64_t* inliers) {          \
    orc::se3_warp(qt, cam, img0, img1, dpt0, img2, w, h, pitch, accum_f64, residual, inliers);                          \
  }                                                                                                                     \
  ORC_API void orc_sfm_step_##SFX(const S* p0, const S* p1, const S* cam, int cs, const S* img0, const S* img1,         \
                                  const S* dpt0, const S* jac, const S* grad1, S* valid0, int w, int h, size_t pitch,   \
                                  size_t jpitch, size_t gpitch, S huber, S avg_dpt, S min_dpt, int border,              \
                                  int accum_f64, int threads, S* jtj, S* jtr, S* residual, uint64_t* inliers) {         \
    orc::sfm_step(p0, p1, cam, cs, img0, img1, dpt0, jac, grad1, valid0, w, h, pitch, jpitch, gpitch, huber, avg_dpt,   \
                  min_dpt, border, accum_f64, threads, jtj, jtr, residual, inliers);                                    \
  }                                                                                                                     \
  ORC_API void orc_sfm_error_##SFX(const S* p0, const S* p1, const S* cam, const S* img0, const S* img1, const S* dpt0, \
                                   int w, int h, size_t pitch, S huber, int accum_f64, S* residual, uint64_t* inliers) {\
    orc::sfm_error(p0, p1, cam, img0, img1, dpt0, w, h, pitch, huber, accum_f64, residual, inliers);                    \
  }                                                                                                                     \
  ORC_API void orc_update_depth_##SFX(int cs, const S* code, const S* prx, const S* jac, S avg_dpt, S* out, int w, int h,\
                                      size_t pitch, size_t jpitch) {                                                    \
    orc::update_depth(cs, code, prx, jac, avg_dpt, out, w, h, pitch, jpitch);                                           \
  }                                                                                                                     \
  ORC_API void orc_depth_aligner_step_##SFX(int cs, const S* code, const S* tgt, const S* prx, const S* jac, S avg_dpt, \
                                            int w, int h, size_t pitch, size_t jpitch, int accum_f64, S* jtj, S* jtr,   \
                                            S* residual, uint64_t* inliers) {                                           \
    orc::depth_aligner_step(cs, code, tgt, prx, jac, avg_dpt, w, h, pitch, jpitch, accum_f64, jtj, jtr, residual, inliers); \
  }                                                                                                                     \
  ORC_API void orc_sparse_geometric_##SFX(const S* p0, const S* p1, const S* c0, const S* c1, const S* cam, int cs, const int* pts,  \
                                          int npts, const S* prx0, const S* jac0, const S* prx1, const S* jac1, const S* dgrad1,      \
                                          int w, int h, size_t pitch, size_t jpitch, size_t gpitch, S huber, S avg_dpt, S* rows) {    \
    orc::sparse_geometric(p0, p1, c0, c1, cam, cs, pts, npts, prx0, jac0, prx1, jac1, dgrad1, w, h, pitch, jpitch, gpitch, huber,     \
                          avg_dpt, rows);                                                                                             \
  }                                                                                                                     \
  ORC_API void orc_sobel_##SFX(const S* img, S* grad, int w, int h, size_t pitch, size_t gpitch) {                      \
    orc::sobel(img, grad, w, h, pitch, gpitch);                                                                         \
  }                                                                                                                     \
  ORC_API void orc_blur_down_##SFX(const S* in, S* out, int w, int h, size_t pitch, int ow, int oh, size_t opitch) {    \
    orc::blur_down(in, out, w, h, pitch, ow, oh, opitch);                                                               \
  }                                                                                                                     \
  ORC_API S orc_squared_error_##SFX(const S* a, const S* b, int w, int h, size_t pitch, int accum_f64) {                \
    return orc::squared_error(a, b, w, h, pitch, accum_f64);                                                            \
  }

ORC_DEFINE(f32, float)
ORC_DEFINE(f64, double)

ORC_API int orc_max_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
