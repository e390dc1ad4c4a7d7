// ref_harness_f1.cpp -- oracle/_ref, part 3: the REFERENCE'S OWN kernel bodies of the pyramid construction and of
// SE3Aligner::Warp (8a-4), compiled from where they lie.
//
// TEST INFRASTRUCTURE ONLY (tests/ and the oracle's own validation): nothing under deepfactors_amd/ may link or load this.
//
// The four kernels live in .cpp files that are CUDA launch syntax around them, so -- like the DepthAligner kernel of part 2 -- they are CUT OUT
// at build time by oracle/Makefile (sed line ranges into oracle/_ref/*.inc; nothing of them is committed) and compiled here as host functions:
//   cuda/cu_image_proc.cpp:34-48    SetSobelCoefficients            (the tap table, written through Eigen's comma initialiser)
//                         :50-55    struct SobelCoeffs, __constant__ SC
//                         :57-92    kernel_sobel_gradients
//                         :119-130  SetGaussCoefficients
//                         :134-164  kernel_gaussian_blur_down
//                         :190-206  kernel_squared_error
//   cuda/cu_se3aligner.cpp:61-113   kernel_warp_calculate
// What is stood in: __global__ / __device__ / __constant__ are defined away; blockIdx / blockDim / threadIdx are host variables that a loop
// over the grid sets per pixel (one thread per block, one block per pixel: the kernels' own index arithmetic runs unchanged);
// vc::runReductions runs the per-pixel lambda over all pixels in order and vc::finalizeReduction stores the sum (as in part 2); the VisionCore
// view gains inBounds / getWithClampedRange (oracle/standins: clamp of the coordinates to the image).  Every tap, weight, normalisation,
// validity rule and sum is the reference's.
#include <math.h>
#include <stdlib.h>

#include <cstddef>
#include <cstdint>
#include <cstring>

#include <Eigen/Core>
#include <sophus/se3.hpp>
#include <VisionCore/Buffers/Image2D.hpp>

#include "pinhole_camera.h"
#include "reduction_items.h"

#define REF_API extern "C" __attribute__((visibility("default")))
#define __global__
#define __device__
#define __constant__

namespace {
struct Dim3 { unsigned x = 0, y = 0, z = 0; };
Dim3 blockIdx, blockDim, threadIdx;
inline int clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
}  // namespace

namespace vc {
template <typename T, typename Target> struct Buffer1DView {
  T* p;
  T* ptr() const { return p; }
};
template <typename F> void runReductions(std::size_t n, F f) { for (std::size_t i = 0; i < n; ++i) f((unsigned int)i); }
template <typename Item, typename W> void finalizeReduction(Item* out, Item* sum, W, Item) { *out = *sum; }
}  // namespace vc

namespace df {
template <typename T> struct reduction_traits { static void WarpReduceSum(T&) {} };   // the scalar specialisation lives under __CUDACC__ (kernel_utils.h)

template <typename Derived>
#include "sobel_coeffs_fn.inc"
#include "sobel_coeffs_struct.inc"
SobelCoeffs SC;
template <typename PixelT, typename TG>
#include "kernel_sobel.inc"
template <typename Derived>
#include "gauss_coeffs_fn.inc"
float gauss_coeffs[25];
template <typename Scalar>
#include "kernel_blur_down.inc"
template <typename Scalar>
#include "kernel_squared_error.inc"

template <typename Scalar>
struct WarpItem : df::CorrespondenceReductionItem<Scalar> {
  static void WarpReduceSum(WarpItem&) {}   // named by the kernel's finalizeReduction call; a device-only member of the reference item
};
template <typename Scalar>
struct SE3Aligner {   // the typedefs of cu_se3aligner.h the kernel template names
  typedef vc::Image2DView<Scalar, vc::TargetHost> ImageBuffer;
  typedef WarpItem<Scalar> CorrespondenceItem;
};
template <typename Scalar, typename BaseT = SE3Aligner<Scalar>>
#include "kernel_warp.inc"
}  // namespace df

namespace {
template <typename K> void for_each_pixel(std::size_t w, std::size_t h, K k) {
  blockDim.x = blockDim.y = 1; threadIdx.x = threadIdx.y = 0;
  for (std::size_t y = 0; y < h; ++y) for (std::size_t x = 0; x < w; ++x) { blockIdx.x = (unsigned)x; blockIdx.y = (unsigned)y; k(); }
}
}  // namespace

REF_API void ref_sobel_gradients(float* img, int w, int h, float* grad_out) {
  typedef Eigen::Matrix<float, 1, 2> G;
  df::SobelCoeffs coeffs;
  Eigen::Map<Eigen::Matrix<float, 3, 3>> mx(coeffs.X), my(coeffs.Y);   // SobelGradients, cu_image_proc.cpp:98-103
  df::SetSobelCoefficients(mx, my);
  df::SC = coeffs;
  vc::Image2DView<float, vc::TargetDeviceCUDA> I(img, w, h, (size_t)w * 4);
  vc::Image2DView<G, vc::TargetDeviceCUDA> Gv(reinterpret_cast<G*>(grad_out), w, h, (size_t)w * 8);
  for_each_pixel(w, h, [&] { df::kernel_sobel_gradients<float, float>(I, Gv); });
}

REF_API void ref_gaussian_blur_down(float* in, int w, int h, float* out, int ow, int oh) {
  float coeffs[25];
  Eigen::Map<Eigen::Matrix<float, 5, 5>> gk(coeffs);                    // GaussianBlurDown, cu_image_proc.cpp:170-175
  df::SetGaussCoefficients(gk);
  std::memcpy(df::gauss_coeffs, coeffs, sizeof coeffs);
  vc::Image2DView<float, vc::TargetDeviceCUDA> I(in, w, h, (size_t)w * 4), O(out, ow, oh, (size_t)ow * 4);
  for_each_pixel(ow, oh, [&] { df::kernel_gaussian_blur_down<float>(I, O); });
}

REF_API float ref_squared_error(float* a, float* b, int w, int h) {
  vc::Image2DView<float, vc::TargetDeviceCUDA> A(a, w, h, (size_t)w * 4), B(b, w, h, (size_t)w * 4);
  float out = 0.f;
  vc::Buffer1DView<float, vc::TargetDeviceCUDA> scratch{ &out };
  df::kernel_squared_error<float>(A, B, scratch);
  return out;
}

REF_API void ref_se3_warp(const float* pose_qt, const float* camv, float* img0, float* img1, float* dpt0, int w, int h, float* img2_out, float* residual,
                          std::uint64_t* inliers) {
  const Sophus::SE3f se3(Sophus::SO3f(pose_qt[0], pose_qt[1], pose_qt[2], pose_qt[3]), Eigen::Matrix<float, 3, 1>(pose_qt[4], pose_qt[5], pose_qt[6]));
  const df::PinholeCamera<float> cam(camv[0], camv[1], camv[2], camv[3], camv[4], camv[5]);
  typedef df::SE3Aligner<float> A;
  A::ImageBuffer I0(img0, w, h, (size_t)w * 4), I1(img1, w, h, (size_t)w * 4), D0(dpt0, w, h, (size_t)w * 4), I2(img2_out, w, h, (size_t)w * 4);
  A::CorrespondenceItem out;
  vc::Buffer1DView<A::CorrespondenceItem, vc::TargetDeviceCUDA> scratch{ &out };
  df::kernel_warp_calculate<float>(se3, cam, I0, I1, D0, I2, scratch);
  *residual = out.residual;
  *inliers = out.inliers;
}
