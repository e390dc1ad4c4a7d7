// Stand-in for <opencv2/opencv.hpp> as core/system/camera_tracker.cpp:78-84 uses it (a float residual image: Mat - Mat, cv::abs, clone) and
// as pinhole_camera_impl.h names it (PinholeCamera::FromFile, never called).  TEST INFRASTRUCTURE.
#pragma once
#include <cmath>
#include <cstddef>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
namespace cv {
class Mat {
 public:
  Mat() : rows(0), cols(0) {}
  Mat(int r, int c) : rows(r), cols(c), d_(std::make_shared<std::vector<float>>((std::size_t)r * c)) {}
  int rows, cols;
  float* ptr() { return d_ ? d_->data() : nullptr; }
  const float* ptr() const { return d_ ? d_->data() : nullptr; }
  template <typename T> T at(int r, int c) const { return static_cast<T>((*d_)[(std::size_t)r * cols + c]); }
  Mat clone() const { Mat m(rows, cols); if (d_) *m.d_ = *d_; return m; }   // a Mat copy shares the pixels, clone() owns its own
  Mat operator-(const Mat& o) const { Mat m(rows, cols); for (std::size_t k = 0; k < d_->size(); ++k) (*m.d_)[k] = (*d_)[k] - (*o.d_)[k]; return m; }
 private:
  std::shared_ptr<std::vector<float>> d_;
};
inline Mat abs(const Mat& a) { Mat m = a.clone(); for (int k = 0; k < m.rows * m.cols; ++k) m.ptr()[k] = std::fabs(m.ptr()[k]); return m; }
struct FileNode {
  void operator>>(int&) const { throw std::runtime_error("cv::FileStorage stand-in"); }
  void operator>>(Mat&) const { throw std::runtime_error("cv::FileStorage stand-in"); }
};
struct FileStorage {
  enum { READ = 0 };
  FileStorage(const std::string&, int) {}
  bool isOpened() const { return false; }
  FileNode operator[](const char*) const { return FileNode(); }
};
}  // namespace cv
