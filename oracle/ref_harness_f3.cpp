// ref_harness_f3.cpp -- oracle/_ref, part 2: the REFERENCE'S OWN code for the two "next" operators of SURVEY section 8f-3, compiled from where it lies.
//
// TEST INFRASTRUCTURE ONLY (tests/ and the oracle's own validation): nothing under deepfactors_amd/ may link or load this.
//
//  * SparseGeometricFactor<float,32>::linearize -- /root/reference/sources/core/gtsam/sparse_geometric_factor.cpp (the whole file, with
//    uniform_sampler.cpp beside it) is #included UNMODIFIED.  What is stood in (oracle/standins_f3/): the GTSAM surface it touches
//    (Values, NonlinearFactor, VerticalBlockMatrix, JacobianFactor, boost::shared_ptr: data carriers, no arithmetic), and three of the
//    reference's own headers that cannot be compiled here and are shadowed by carriers of the same name: keyframe.h (CUDA-synced
//    pyramids -> host views), cu_image_proc.h and gtsam_traits.h (included by the .cpp, unused by linearize).  Every formula of the
//    factor -- RelativePose, DepthFromCode, FindCorrespondence and its Jacobians, the nearest-neighbour lookup, the Huber weight, the
//    row layout [pose0 | pose1 | code0 | code1 | err] -- is the reference's.
//  * kernel_depthaligner_run_step -- the kernel template is cut out of /root/reference/sources/cuda/cu_depthaligner.cpp at BUILD time
//    (oracle/Makefile: a sed line range into oracle/_ref/depthaligner_kernel.inc; that file also holds CUDA launch syntax no host compiler
//    takes, and nothing of it is committed) and compiled here as a host function: __global__ / __device__ are defined away,
//    vc::runReductions runs the per-pixel lambda over all pixels in order, vc::finalizeReduction stores the thread's sum.  The item type
//    exposes HessianType as a static function (the kernel writes `Item::HessianType(J.transpose())` without `typename`, which nvcc accepts
//    and g++ does not -- same workaround as for lucas_kanade_se3.h:74 in ref_harness.cpp).
#include <math.h>
#include <stdlib.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define DF_CODE_SIZE 32
#define DF_GTSAM_TRAITS_H_   // core/gtsam/gtsam_traits.h sits beside the .cpp, so a quoted include finds it before any stand-in: its include guard
                             // is set instead (gtsam::traits for Sophus types need the real GTSAM; linearize() uses none of it)
#include "sparse_geometric_factor.cpp"   // + its headers: warping.h, dense_sfm.h, pinhole_camera.h (reference), gtsam/... keyframe.h (stand-ins)
#include "uniform_sampler.cpp"

#include "reduction_items.h"

#define REF_API extern "C" __attribute__((visibility("default")))

namespace {
Sophus::SE3f pose_from_qt3(const float* qt) {
  return Sophus::SE3f(Sophus::SO3f(qt[0], qt[1], qt[2], qt[3]), Eigen::Matrix<float, 3, 1>(qt[4], qt[5], qt[6]));
}
}  // namespace

// rows_out: [npts][12 + 2 * 32 + 1] row-major doubles = the VerticalBlockMatrix of the JacobianFactor the reference returns
REF_API int ref_sparse_geometric_cs32(const float* pose0_qt, const float* pose1_qt, const float* code0, const float* code1, const float* camv, int w, int h,
                                      const int* pts_xy, int npts, float* prx0, float* jac0, float* prx1, float* jac1, float* dgrad1, float huber_delta,
                                      double* rows_out) {
  constexpr int CS = 32;
  typedef df::SparseGeometricFactor<float, CS> Factor;
  typedef df::Keyframe<float> KF;
  auto kf0 = std::make_shared<KF>(), kf1 = std::make_shared<KF>();
  kf0->id = 0; kf1->id = 1;
  kf0->pyr_prx_orig.level0 = vc::Image2DView<float, vc::TargetHost>(prx0, w, h, (size_t)w * 4);
  kf0->pyr_jac.level0 = vc::Image2DView<float, vc::TargetHost>(jac0, (size_t)w * CS, h, (size_t)w * CS * 4);
  kf1->pyr_prx_orig.level0 = vc::Image2DView<float, vc::TargetHost>(prx1, w, h, (size_t)w * 4);
  kf1->pyr_jac.level0 = vc::Image2DView<float, vc::TargetHost>(jac1, (size_t)w * CS, h, (size_t)w * CS * 4);
  kf1->dpt_grad = vc::Image2DView<KF::GradT, vc::TargetHost>(reinterpret_cast<KF::GradT*>(dgrad1), w, h, (size_t)w * 8);
  std::vector<df::Point> pts((size_t)npts);
  for (int i = 0; i < npts; ++i) { pts[(size_t)i].x = pts_xy[2 * i]; pts[(size_t)i].y = pts_xy[2 * i + 1]; }
  const df::PinholeCamera<float> cam(camv[0], camv[1], camv[2], camv[3], camv[4], camv[5]);
  const gtsam::Key kp0 = 1, kp1 = 2, kc0 = 3, kc1 = 4;
  Factor factor(cam, pts, kf0, kf1, kp0, kp1, kc0, kc1, huber_delta, false);
  gtsam::Values vals;
  vals.insert(kp0, pose_from_qt3(pose0_qt));
  vals.insert(kp1, pose_from_qt3(pose1_qt));
  vals.insert(kc0, gtsam::Vector(std::vector<double>(code0, code0 + CS)));
  vals.insert(kc1, gtsam::Vector(std::vector<double>(code1, code1 + CS)));
  const auto gf = factor.linearize(vals);
  const auto* jf = dynamic_cast<const gtsam::JacobianFactor*>(gf.get());
  if (!jf) return -1;
  const gtsam::VerticalBlockMatrix& Ab = jf->matrixObject();
  if (Ab.rows() != npts || Ab.cols() != 12 + 2 * CS + 1) return -2;
  std::memcpy(rows_out, Ab.data().data(), sizeof(double) * Ab.data().size());
  return 0;
}

// ---- DepthAligner kernel (cu_depthaligner.cpp:32-72) as a host function ------------------------------------------------------------------
#define __global__
#define __device__
namespace vc {
template <typename T, typename Target> struct Buffer1DView {
  T* p;
  T* ptr() const { return p; }
};
template <typename F> void runReductions(std::size_t n, F f) { for (std::size_t i = 0; i < n; ++i) f((unsigned int)i); }
template <typename Item, typename W> void finalizeReduction(Item* out, Item* sum, W, Item) { *out = *sum; }
}  // namespace vc
namespace df {
template <int CS>
struct DepthItem : df::JTJJrReductionItem<float, CS> {
  template <typename V>
  static vc::types::SquareUpperTriangularMatrix<float, CS> HessianType(const V& v) { return vc::types::SquareUpperTriangularMatrix<float, CS>(v); }
  static void WarpReduceSum(DepthItem&) {}   // named by the kernel's finalizeReduction call; a device-only member of the reference item
};
template <typename Scalar, int CS>
struct DepthAligner {   // the typedefs of cu_depthaligner.h the kernel template names
  typedef Eigen::Matrix<Scalar, CS, 1> CodeT;
  typedef vc::Image2DView<Scalar, vc::TargetHost> ImageBuffer;
  typedef DepthItem<CS> ReductionItem;
};
#include "depthaligner_kernel.inc"
}  // namespace df

REF_API void ref_depth_aligner_step_cs32(const float* code, float* tgt_dpt, float* prx_orig, float* prx_jac, int w, int h, float* JtJ, float* Jtr, float* residual,
                                         std::uint64_t* inliers) {
  constexpr int CS = 32;
  typedef df::DepthAligner<float, CS> DA;
  DA::CodeT c;
  for (int i = 0; i < CS; ++i) c(i) = code[i];
  DA::ImageBuffer T(tgt_dpt, w, h, (size_t)w * 4), P(prx_orig, w, h, (size_t)w * 4), J(prx_jac, (size_t)w * CS, h, (size_t)w * CS * 4);
  DA::ReductionItem out;
  vc::Buffer1DView<DA::ReductionItem, vc::TargetDeviceCUDA> scratch{ &out };
  df::kernel_depthaligner_run_step<float, CS>(c, T, P, J, scratch);
  const int NT = CS * (CS + 1) / 2;
  for (int k = 0; k < NT; ++k) JtJ[k] = out.JtJ.coeff()(k);
  for (int k = 0; k < CS; ++k) Jtr[k] = out.Jtr(k);
  *residual = out.residual;
  *inliers = out.inliers;
}
