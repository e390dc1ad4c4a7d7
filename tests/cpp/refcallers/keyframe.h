// Stand-in that SHADOWS the reference's core/mapping/keyframe.h + frame.h (they pull in CUDA-synced pyramids, OpenCV, DBoW2) for the callers'
// compile test: the members core/gtsam/photometric_factor.cpp and core/system/camera_tracker.cpp read, with the reference's names and shapes
// (frame.h:36-120, keyframe.h:34-100: pyr_jac is [H][W * CS]), as device-resident pyramids whose GetGpuLevel(i) is the view the aligners take.
// TEST INFRASTRUCTURE (tests/cpp/ref_callers_test.cpp): carriers only.
#pragma once
#include <cstddef>
#include <memory>
#include <string>

#include <Eigen/Dense>
#include <sophus/se3.hpp>
#include <glog/logging.h>
#include <VisionCore/Buffers/BufferPyramid.hpp>

namespace df {

template <typename T>
class SyncedBufferPyramid {   // cuda/synced_pyramid.h as its users see it; device side only
 public:
  SyncedBufferPyramid(std::size_t levels, std::size_t w, std::size_t h) : p_(new vc::RuntimeBufferPyramidManaged<T, vc::TargetDeviceCUDA>(levels, w, h)) {}
  // references, as cuda/synced_pyramid.h:118-126 returns them (callers bind them to `ImageBuf&` parameters)
  const vc::Image2DView<T, vc::TargetDeviceCUDA>& GetGpuLevel(int i) const { return (*p_)[(std::size_t)i]; }
  vc::Image2DView<T, vc::TargetDeviceCUDA>& GetGpuLevel(int i) { return (*p_)[(std::size_t)i]; }
 private:
  std::shared_ptr<vc::RuntimeBufferPyramidManaged<T, vc::TargetDeviceCUDA>> p_;
};

template <typename Scalar>
class Frame {
 public:
  typedef std::shared_ptr<Frame<Scalar>> Ptr;
  typedef Sophus::SE3<Scalar> SE3T;
  typedef Eigen::Matrix<Scalar, 1, 2> GradT;
  Frame(std::size_t pyrlevels, std::size_t w, std::size_t h) : pyr_img(pyrlevels, w, h), pyr_grad(pyrlevels, w, h), id(0) {}
  virtual ~Frame() {}
  SyncedBufferPyramid<Scalar> pyr_img;
  SyncedBufferPyramid<GradT> pyr_grad;
  SE3T pose_wk;
  std::size_t id;
};

template <typename Scalar>
class Keyframe : public Frame<Scalar> {
 public:
  typedef std::shared_ptr<Keyframe<Scalar>> Ptr;
  typedef Eigen::Matrix<Scalar, Eigen::Dynamic, 1> CodeT;
  Keyframe(std::size_t pyrlevels, std::size_t w, std::size_t h, std::size_t cs)
      : Frame<Scalar>(pyrlevels, w, h), pyr_dpt(pyrlevels, w, h), pyr_vld(pyrlevels, w, h), pyr_stdev(pyrlevels, w, h), pyr_prx_orig(pyrlevels, w, h),
        pyr_jac(pyrlevels, cs * w, h) {}
  SyncedBufferPyramid<Scalar> pyr_dpt, pyr_vld, pyr_stdev, pyr_prx_orig, pyr_jac;
  CodeT code;
};

}  // namespace df
