// Stand-in for <opencv2/opencv.hpp>: pinhole_camera_impl.h includes it only for PinholeCamera::FromFile (:139-157), which the
// oracle never calls; these declarations let that member's body parse.  TEST INFRASTRUCTURE (oracle/_ref).
#pragma once
#include <stdexcept>
#include <string>
namespace cv {
struct Mat { template <typename T> T at(int, int) const { throw std::runtime_error("cv::Mat stand-in"); } };
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
