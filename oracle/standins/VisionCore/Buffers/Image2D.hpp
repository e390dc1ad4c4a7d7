// Stand-in for VisionCore's 2-D image views (jczarnowski/vision_core, absent).  TEST INFRASTRUCTURE (oracle/_ref).
// Semantics as SURVEY.md appendix B restates them (not verifiable in the reference tree -- "parity unpinned" for exactly these
// two conventions): element (x, y) at (char*)ptr + y * pitch + x * sizeof(T); getBilinear: floor, lerp in x then in y,
// lerp(a, b, t) = a + t * (b - a), no bounds check.
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

#include <Eigen/Core>

namespace vc {

struct TargetHost {};
struct TargetDeviceCUDA {};

template <typename T, typename Target>
class Image2DView {
 public:
  typedef T ValueType;
  Image2DView() : ptr_(nullptr), pitch_(0), w_(0), h_(0) {}
  Image2DView(T* ptr, std::size_t w, std::size_t h, std::size_t pitch_bytes) : ptr_(ptr), pitch_(pitch_bytes), w_(w), h_(h) {}
  T* ptr() const { return ptr_; }
  std::size_t pitch() const { return pitch_; }
  std::size_t width() const { return w_; }
  std::size_t height() const { return h_; }
  std::size_t area() const { return w_ * h_; }
  T* rowPtr(std::size_t y) const { return reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(ptr_)) + y * pitch_); }
  T& operator()(std::size_t x, std::size_t y) const { return rowPtr(y)[x]; }
  // Buffer2DView::inBounds / getWithClampedRange (used by the image-proc kernels, cu_image_proc.cpp:71,84): inside the image / the nearest pixel inside
  template <typename I> bool inBounds(I x, I y) const { return x >= I(0) && y >= I(0) && (std::size_t)x < w_ && (std::size_t)y < h_; }
  const T& getWithClampedRange(int x, int y) const {
    const int cx = x < 0 ? 0 : (x >= (int)w_ ? (int)w_ - 1 : x), cy = y < 0 ? 0 : (y >= (int)h_ ? (int)h_ - 1 : y);
    return rowPtr((std::size_t)cy)[cx];
  }

  template <typename TR, typename S>
  TR getBilinear(S u, S v) const {
    using std::floor;
    const S ix = floor(u), iy = floor(v);
    const S fx = u - ix, fy = v - iy;
    const T* bl = rowPtr(static_cast<std::size_t>(iy)) + static_cast<std::size_t>(ix);
    const T* tl = rowPtr(static_cast<std::size_t>(iy) + 1) + static_cast<std::size_t>(ix);
    const TR a = lerp<TR>(bl[0], bl[1], fx), b = lerp<TR>(tl[0], tl[1], fx);
    return lerp<TR>(a, b, fy);
  }
  template <typename TR, typename S>
  TR getBilinear(const Eigen::Matrix<S, 2, 1>& p) const { return getBilinear<TR, S>(p(0), p(1)); }

 private:
  template <typename TR, typename A, typename S> static TR lerp(const A& a, const A& b, S t) { return TR(a + t * (b - a)); }
  T* ptr_;
  std::size_t pitch_, w_, h_;
};

// owning host image; only named by RenderDpt's default template argument (warping.h:71-91)
template <typename T, typename Target>
class Image2DManaged {
 public:
  typedef Image2DView<T, Target> ViewT;
  Image2DManaged(std::size_t w, std::size_t h) : d_(w * h), w_(w), h_(h) {}
  T& operator()(std::size_t x, std::size_t y) { return d_[y * w_ + x]; }
  operator ViewT() { return ViewT(d_.data(), w_, h_, w_ * sizeof(T)); }
 private:
  std::vector<T> d_;
  std::size_t w_, h_;
};

template <typename T, typename Target> using Buffer2DView = Image2DView<T, Target>;

}  // namespace vc
