// VisionCore image buffers as the reference's CALLERS of the hot path use them (core/system/camera_tracker.cpp:72-84): on top of the view of
// oracle/standins (element (x, y) at ptr + y * pitch + x * sizeof(T)) the OWNING images -- device memory from HIP instead of VisionCore's CUDA
// allocator ("SyncedBufferPyramid only needs its allocator switched"), copyFrom across host / device, getOpenCV().
// TEST INFRASTRUCTURE (tests/cpp/ref_callers_test.cpp): carriers, no arithmetic.
#pragma once
#include_next <VisionCore/Buffers/Image2D.hpp>

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <opencv2/opencv.hpp>

#define VC_HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::abort(); } } while (0)

namespace vc {

template <typename T>
class Image2DManaged<T, TargetDeviceCUDA> : public Image2DView<T, TargetDeviceCUDA> {
 public:
  typedef Image2DView<T, TargetDeviceCUDA> ViewT;
  Image2DManaged(std::size_t w, std::size_t h) : ViewT(alloc(w, h), w, h, w * sizeof(T)) {}
  ~Image2DManaged() { (void)hipFree(this->ptr()); }
  Image2DManaged(const Image2DManaged&) = delete;
  Image2DManaged& operator=(const Image2DManaged&) = delete;
  Image2DManaged(Image2DManaged&& o) : ViewT(o) { static_cast<ViewT&>(o) = ViewT(); }
  ViewT view() const { return *this; }
  void copyFrom(const Image2DView<T, TargetHost>& s) { VC_HIPOK(hipMemcpy2D(this->ptr(), this->pitch(), s.ptr(), s.pitch(), this->width() * sizeof(T), this->height(), hipMemcpyHostToDevice)); }
  void copyFrom(const Image2DView<T, TargetDeviceCUDA>& s) { VC_HIPOK(hipMemcpy2D(this->ptr(), this->pitch(), s.ptr(), s.pitch(), this->width() * sizeof(T), this->height(), hipMemcpyDeviceToDevice)); }
 private:
  static T* alloc(std::size_t w, std::size_t h) { T* p = nullptr; VC_HIPOK(hipMalloc((void**)&p, w * h * sizeof(T))); VC_HIPOK(hipMemset(p, 0, w * h * sizeof(T))); return p; }
};

template <typename T>
class Image2DManaged<T, TargetHost> : public Image2DView<T, TargetHost> {
 public:
  typedef Image2DView<T, TargetHost> ViewT;
  Image2DManaged(std::size_t w, std::size_t h) : ViewT(static_cast<T*>(std::calloc(w * h, sizeof(T))), w, h, w * sizeof(T)) {}
  ~Image2DManaged() { std::free(this->ptr()); }
  Image2DManaged(const Image2DManaged&) = delete;
  Image2DManaged& operator=(const Image2DManaged&) = delete;
  void copyFrom(const Image2DView<T, TargetDeviceCUDA>& s) { VC_HIPOK(hipMemcpy2D(this->ptr(), this->pitch(), s.ptr(), s.pitch(), this->width() * sizeof(T), this->height(), hipMemcpyDeviceToHost)); }
  void copyFrom(const Image2DView<T, TargetHost>& s) { for (std::size_t y = 0; y < this->height(); ++y) std::memcpy(this->rowPtr(y), s.rowPtr(y), this->width() * sizeof(T)); }
  cv::Mat getOpenCV() const {
    cv::Mat m((int)this->height(), (int)this->width());
    for (std::size_t y = 0; y < this->height(); ++y) for (std::size_t x = 0; x < this->width(); ++x) m.ptr()[y * this->width() + x] = static_cast<float>((*this)(x, y));
    return m;
  }
};

template <typename T, typename Target> using Buffer2DManaged = Image2DManaged<T, Target>;

}  // namespace vc
