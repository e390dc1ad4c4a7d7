// vc::RuntimeBufferPyramidManaged as core/system/camera_tracker.h:45-46 names it: `levels` owning images, level i + 1 half the size of level i.
// TEST INFRASTRUCTURE.
#pragma once
#include <memory>
#include <vector>

#include <VisionCore/Buffers/Image2D.hpp>

namespace vc {
template <typename T, typename Target>
class RuntimeBufferPyramidManaged {
 public:
  RuntimeBufferPyramidManaged(std::size_t levels, std::size_t w, std::size_t h) {
    for (std::size_t i = 0; i < levels; ++i) { lv_.emplace_back(new Image2DManaged<T, Target>(w, h)); w /= 2; h /= 2; }
  }
  std::size_t getLevelCount() const { return lv_.size(); }
  Image2DManaged<T, Target>& operator[](std::size_t i) { return *lv_[i]; }
  const Image2DManaged<T, Target>& operator[](std::size_t i) const { return *lv_[i]; }
 private:
  std::vector<std::unique_ptr<Image2DManaged<T, Target>>> lv_;
};
}  // namespace vc
