// Stand-in that SHADOWS the reference's core/mapping/keyframe.h for oracle/_ref (that header pulls in CUDA-synced pyramids, OpenCV and
// DBoW2): the three members core/gtsam/sparse_geometric_factor.cpp reads -- pyr_jac / pyr_prx_orig via GetCpuLevel(0), dpt_grad(x, y) --
// as views over caller-owned host arrays, plus `id`.  TEST INFRASTRUCTURE; layouts per mapping/keyframe.h:46-56,88-92.
#pragma once
#include <cstddef>
#include <memory>

#include <Eigen/Core>
#include <VisionCore/Buffers/Image2D.hpp>

namespace df {

template <typename Scalar>
struct HostPyramidView {
  vc::Image2DView<Scalar, vc::TargetHost> level0;
  const vc::Image2DView<Scalar, vc::TargetHost>& GetCpuLevel(int) const { return level0; }
};

template <typename Scalar>
class Keyframe {
 public:
  typedef std::shared_ptr<Keyframe<Scalar>> Ptr;
  typedef Eigen::Matrix<Scalar, 1, 2> GradT;
  std::size_t id = 0;
  HostPyramidView<Scalar> pyr_jac, pyr_prx_orig;
  vc::Image2DView<GradT, vc::TargetHost> dpt_grad;
};

}  // namespace df
