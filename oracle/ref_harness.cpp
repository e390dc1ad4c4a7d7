// ref_harness.cpp -- oracle/_ref: the REFERENCE'S OWN per-pixel code, compiled from where it lies.
//
// TEST INFRASTRUCTURE ONLY (tests/ and the oracle's own validation): nothing under deepfactors_amd/ may link or load this.
//
// This translation unit includes, UNMODIFIED and from /root/reference (passed with -I by oracle/Makefile, never copied):
//     sources/common/algorithm/warping.h            RelativePose + Jacobians, FindCorrespondence, warp Jacobians, DepthFromCode
//     sources/common/algorithm/dense_sfm.h           DenseSfm, DenseSfm_EvaluateError, DenseSfmParams
//     sources/common/algorithm/lucas_kanade_se3.h    LucasKanadeSE3, SE3SolveAndUpdate
//     sources/common/algorithm/pinhole_camera{,_impl}.h, m_estimators.h
//     sources/cuda/reduction_items.h, kernel_utils.h (their host parts)
// against the stand-in headers under oracle/standins/ for the un-vendored Eigen / Sophus / VisionCore / OpenCV (the
// reference's thirdparty/ submodules are empty).  What is reference-derived here: every formula of the path (warp, projection,
// all Jacobians, Huber weight, the 44-column row, the accumulated products).  What is NOT (stated in DESIGN.md): the small
// linear algebra (stand-in Eigen), the quaternion arithmetic (stand-in Sophus, restating Sophus' formulas), bilinear sampling and
// the packed upper-triangular order (stand-in VisionCore).
//
// Host loops mirror the reference's own CPU evaluation, tests/ut_sfmaligner.cpp:299-315 (RelativePose, then DenseSfm per pixel
// with vc::TargetHost views) -- per pixel into a fresh item, summed here in double so that the comparison with the oracle does
// not depend on a float summation order.
#include <cstdint>
#include <cstring>
#include <vector>

#include "dense_sfm.h"
#include "lucas_kanade_se3.h"
#include <ostream>
#include "camera_pyramid.h"

#define REF_API extern "C" __attribute__((visibility("default")))

namespace {

typedef vc::Image2DView<float, vc::TargetHost> ImgF;
typedef Eigen::Matrix<float, 1, 2> GradT;
typedef vc::Image2DView<GradT, vc::TargetHost> ImgG;

// lucas_kanade_se3.h:74 writes `ReductionItem::HessianType(J.transpose())` without `typename`: nvcc accepts that, g++ and clang do
// not (a dependent name is a non-type unless marked).  LucasKanadeSE3 takes its item type as a template parameter, so the header
// compiles unmodified with an item whose `HessianType` IS a non-type: a static function that builds the very same packed matrix.
struct Se3Item : df::JTJJrReductionItem<float, 6> {
  template <typename V>
  static vc::types::SquareUpperTriangularMatrix<float, 6> HessianType(const V& v) { return vc::types::SquareUpperTriangularMatrix<float, 6>(v); }
};

Sophus::SE3f pose_from_qt(const float* qt) {   // (qx qy qz qw tx ty tz), like Sophus::SE3f(Quaternion, translation)
  return Sophus::SE3f(Sophus::SO3f(qt[0], qt[1], qt[2], qt[3]), Eigen::Matrix<float, 3, 1>(qt[4], qt[5], qt[6]));
}
df::PinholeCamera<float> cam_from(const float* c) { return df::PinholeCamera<float>(c[0], c[1], c[2], c[3], c[4], c[5]); }

template <int CS>
void sfm_step_t(const float* p0, const float* p1, const float* camv, const float* img0, const float* img1, const float* dpt0, const float* std0,
                float* valid0, const float* jac, const float* grad1, int w, int h, size_t pitch, size_t jpitch, size_t gpitch, float huber, float avg_dpt,
                float min_dpt, int border, double* JtJ, double* Jtr, double* residual, uint64_t* inliers) {
  constexpr int NP = 12 + CS;
  typedef df::JTJJrReductionItem<float, NP> Item;
  const Sophus::SE3f pose0 = pose_from_qt(p0), pose1 = pose_from_qt(p1);
  Eigen::Matrix<float, 6, 6> J0, J1;
  const Sophus::SE3f pose_10 = df::RelativePose(pose1, pose0, J1, J0);   // cu_sfmaligner.cpp:166
  const df::PinholeCamera<float> cam = cam_from(camv);
  df::DenseSfmParams prm;
  prm.huber_delta = huber; prm.avg_dpt = avg_dpt; prm.min_dpt = min_dpt; prm.valid_border = border;
  std::vector<float> zeros((size_t)w, 0.0f), vscratch;
  ImgF I0(const_cast<float*>(img0), w, h, pitch), I1(const_cast<float*>(img1), w, h, pitch), D0(const_cast<float*>(dpt0), w, h, pitch);
  ImgF S0 = std0 ? ImgF(const_cast<float*>(std0), w, h, pitch) : ImgF(zeros.data(), w, h, 0);
  if (!valid0) { vscratch.assign((size_t)w * h, 0.0f); valid0 = vscratch.data(); }
  ImgF V0(valid0, w, h, valid0 == vscratch.data() ? (size_t)w * sizeof(float) : pitch);
  ImgF JC(const_cast<float*>(jac), (size_t)w * CS, h, jpitch);
  ImgG G1(reinterpret_cast<GradT*>(const_cast<float*>(grad1)), w, h, gpitch);
  const Eigen::Matrix<float, CS, 1> code = Eigen::Matrix<float, CS, 1>::Zero();   // unused by DenseSfm (depth is already decoded)
  const int NT = NP * (NP + 1) / 2;
  for (int k = 0; k < NT; ++k) JtJ[k] = 0.0;
  for (int k = 0; k < NP; ++k) Jtr[k] = 0.0;
  double res = 0.0;
  uint64_t inl = 0;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      Item it;
      df::DenseSfm<float, CS, vc::TargetHost>(x, y, pose_10, J0, J1, code, cam, I0, I1, D0, S0, V0, JC, G1, prm, it);
      if (!it.inliers) continue;
      inl += it.inliers;
      res += (double)it.residual;
      for (int k = 0; k < NT; ++k) JtJ[k] += (double)it.JtJ.coeff()(k);
      for (int k = 0; k < NP; ++k) Jtr[k] += (double)it.Jtr(k);
    }
  *residual = res;
  *inliers = inl;
}

}  // namespace

// SfmAligner::RunStep as the reference's own host evaluation (ut_sfmaligner.cpp:299-315).  Same argument order as orc_sfm_step_f32,
// double outputs.  valid0 / std0 may be null.  Returns 0, or -1 for a code size that is not instantiated here.
REF_API int ref_sfm_step_f32(const float* p0, const float* p1, const float* cam, int cs, const float* img0, const float* img1, const float* dpt0,
                             const float* std0, float* valid0, const float* jac, const float* grad1, int w, int h, size_t pitch, size_t jpitch, size_t gpitch,
                             float huber, float avg_dpt, float min_dpt, int border, double* JtJ, double* Jtr, double* residual, uint64_t* inliers) {
  switch (cs) {
    case 16: sfm_step_t<16>(p0, p1, cam, img0, img1, dpt0, std0, valid0, jac, grad1, w, h, pitch, jpitch, gpitch, huber, avg_dpt, min_dpt, border, JtJ, Jtr, residual, inliers); return 0;
    case 32: sfm_step_t<32>(p0, p1, cam, img0, img1, dpt0, std0, valid0, jac, grad1, w, h, pitch, jpitch, gpitch, huber, avg_dpt, min_dpt, border, JtJ, Jtr, residual, inliers); return 0;
    case 64: sfm_step_t<64>(p0, p1, cam, img0, img1, dpt0, std0, valid0, jac, grad1, w, h, pitch, jpitch, gpitch, huber, avg_dpt, min_dpt, border, JtJ, Jtr, residual, inliers); return 0;
    default: return -1;
  }
}

// SfmAligner::EvaluateError (DenseSfm_EvaluateError per pixel, dense_sfm.h:79-119)
REF_API void ref_sfm_error_f32(const float* p0, const float* p1, const float* camv, const float* img0, const float* img1, const float* dpt0, const float* grad1,
                               int w, int h, size_t pitch, size_t gpitch, float huber, float avg_dpt, double* residual, uint64_t* inliers) {
  const Sophus::SE3f pose_10 = df::RelativePose(pose_from_qt(p1), pose_from_qt(p0));
  const df::PinholeCamera<float> cam = cam_from(camv);
  df::DenseSfmParams prm;
  prm.huber_delta = huber; prm.avg_dpt = avg_dpt;
  std::vector<float> zeros((size_t)w, 0.0f);
  ImgF I0(const_cast<float*>(img0), w, h, pitch), I1(const_cast<float*>(img1), w, h, pitch), D0(const_cast<float*>(dpt0), w, h, pitch), S0(zeros.data(), w, h, 0);
  ImgG G1(reinterpret_cast<GradT*>(const_cast<float*>(grad1)), w, h, gpitch);
  double res = 0.0;
  uint64_t inl = 0;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      df::CorrespondenceReductionItem<float> it;
      df::DenseSfm_EvaluateError<float, 32, vc::TargetHost>(x, y, pose_10, cam, I0, I1, D0, S0, G1, prm, it);
      res += (double)it.residual; inl += it.inliers;
    }
  *residual = res; *inliers = inl;
}

// SE3Aligner::RunStep (LucasKanadeSE3 per pixel, lucas_kanade_se3.h:41-77); pose = pose_10 directly
REF_API void ref_se3_step_f32(const float* qt, const float* camv, const float* img0, const float* img1, const float* dpt0, const float* grad1, int w, int h,
                              size_t pitch, size_t gpitch, float huber, double* JtJ21, double* Jtr6, double* residual, uint64_t* inliers) {
  const Sophus::SE3f se3 = pose_from_qt(qt);
  const df::PinholeCamera<float> cam = cam_from(camv);
  ImgF I0(const_cast<float*>(img0), w, h, pitch), I1(const_cast<float*>(img1), w, h, pitch), D0(const_cast<float*>(dpt0), w, h, pitch);
  ImgG G1(reinterpret_cast<GradT*>(const_cast<float*>(grad1)), w, h, gpitch);
  for (int k = 0; k < 21; ++k) JtJ21[k] = 0.0;
  for (int k = 0; k < 6; ++k) Jtr6[k] = 0.0;
  double res = 0.0;
  uint64_t inl = 0;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const Se3Item it = df::LucasKanadeSE3<float, vc::TargetHost, Sophus::SE3f, df::PinholeCamera<float>, Se3Item>(x, y, se3, cam, I0, I1, D0, G1, huber);
      if (!it.inliers) continue;
      inl += it.inliers; res += (double)it.residual;
      for (int k = 0; k < 21; ++k) JtJ21[k] += (double)it.JtJ.coeff()(k);
      for (int k = 0; k < 6; ++k) Jtr6[k] += (double)it.Jtr(k);
    }
  *residual = res; *inliers = inl;
}

// The reference's Gauss-Newton update of the tracker / of ut_se3aligner's ImageAlignmentTest (lucas_kanade_se3.h:85-95): float JtJ (packed
// upper triangle, 21) and Jtr (6) in, pose (qx qy qz qw tx ty tz) updated in place.
REF_API void ref_se3_solve_and_update_f32(const float* JtJ21, const float* Jtr6, float* qt) {
  Eigen::Matrix<float, 6, 6> H;
  int k = 0;
  for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { H(r, c) = JtJ21[k]; H(c, r) = JtJ21[k]; ++k; }
  Eigen::Matrix<float, 6, 1> g;
  for (int i = 0; i < 6; ++i) g(i) = Jtr6[i];
  Sophus::SE3f est = pose_from_qt(qt);
  df::SE3SolveAndUpdate(H, g, est);
  const auto q = est.unit_quaternion();
  qt[0] = q.x(); qt[1] = q.y(); qt[2] = q.z(); qt[3] = q.w();
  qt[4] = est.translation()(0); qt[5] = est.translation()(1); qt[6] = est.translation()(2);
}

// UpdateDepth's per-pixel body (cu_image_proc.cpp:258-262): DepthFromCode(code, prx_J_cde, prx_orig(x,y), avg_dpt)
template <int CS>
static void update_depth_t(const float* code, const float* prx, const float* jac, float avg_dpt, float* out, int w, int h, size_t pitch, size_t jpitch) {
  Eigen::Matrix<float, CS, 1> c;
  for (int i = 0; i < CS; ++i) c(i) = code[i];
  ImgF P(const_cast<float*>(prx), w, h, pitch), JC(const_cast<float*>(jac), (size_t)w * CS, h, jpitch), O(out, w, h, pitch);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      Eigen::Map<const Eigen::Matrix<float, 1, CS>> tmp(&JC((size_t)x * CS, y));
      const Eigen::Matrix<float, 1, CS> prx_J_cde(tmp);
      O(x, y) = df::DepthFromCode(c, prx_J_cde, P(x, y), avg_dpt);
    }
}
REF_API int ref_update_depth_f32(int cs, const float* code, const float* prx, const float* jac, float avg_dpt, float* out, int w, int h, size_t pitch, size_t jpitch) {
  switch (cs) {
    case 16: update_depth_t<16>(code, prx, jac, avg_dpt, out, w, h, pitch, jpitch); return 0;
    case 32: update_depth_t<32>(code, prx, jac, avg_dpt, out, w, h, pitch, jpitch); return 0;
    case 64: update_depth_t<64>(code, prx, jac, avg_dpt, out, w, h, pitch, jpitch); return 0;
    default: return -1;
  }
}

// RelativePose with both Jacobians (warping.h:105-137), row-major 6x6 out
REF_API void ref_relative_pose_f32(const float* a_qt, const float* b_qt, float* out_qt, float* jac_a36, float* jac_b36) {
  Eigen::Matrix<float, 6, 6> Ja, Jb;
  const Sophus::SE3f ab = df::RelativePose(pose_from_qt(a_qt), pose_from_qt(b_qt), Ja, Jb);
  const auto q = ab.unit_quaternion();
  out_qt[0] = q.x(); out_qt[1] = q.y(); out_qt[2] = q.z(); out_qt[3] = q.w();
  for (int i = 0; i < 3; ++i) out_qt[4 + i] = ab.translation()(i);
  for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { jac_a36[r * 6 + c] = Ja(r, c); jac_b36[r * 6 + c] = Jb(r, c); }
}

// CameraPyramid (camera_pyramid.h:35-48): out = levels x (fx, fy, u0, v0, w, h)
REF_API void ref_camera_pyramid_f32(const float* cam6, int levels, float* out) {
  const df::PinholeCamera<float> cam(cam6[0], cam6[1], cam6[2], cam6[3], cam6[4], cam6[5]);
  const df::CameraPyramid<float> pyr(cam, (std::size_t)levels);
  for (int i = 0; i < levels; ++i) {
    out[i * 6 + 0] = pyr[i].fx(); out[i * 6 + 1] = pyr[i].fy(); out[i * 6 + 2] = pyr[i].u0(); out[i * 6 + 3] = pyr[i].v0();
    out[i * 6 + 4] = pyr[i].width(); out[i * 6 + 5] = pyr[i].height();
  }
}

REF_API float ref_huber_weight_f32(float x, float delta) { return df::HuberWeight(x, delta); }
REF_API float ref_depth_jacobian_prx_f32(float d, float a) { return df::DepthJacobianPrx(d, a); }
REF_API const char* ref_sources() {
  return "sources/common/algorithm/{warping,dense_sfm,lucas_kanade_se3,pinhole_camera,pinhole_camera_impl,m_estimators,camera_pyramid}.h + sources/cuda/{reduction_items,kernel_utils}.h, unmodified";
}
