// Stand-in for the slice of GTSAM 4.0 (boost::shared_ptr era; thirdparty/gtsam of the reference is an empty submodule) that the reference's
// core/gtsam/photometric_factor.{h,cpp} and gtsam_traits.h touch.  TEST INFRASTRUCTURE (tests/cpp/ref_callers_test.cpp): data carriers only --
// Values holds poses and codes by key, NonlinearFactor is the interface the factor overrides, traits<> is the primary template the
// reference's own gtsam_traits.h specialises for Sophus::SE3 (compiled unmodified), boost::shared_ptr maps onto std.
#pragma once
#include <cmath>
#include <cstdint>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include <Eigen/Dense>
#include <sophus/se3.hpp>
#include <glog/logging.h>

namespace boost {
template <typename T> using shared_ptr = std::shared_ptr<T>;
template <typename T, typename... A> std::shared_ptr<T> make_shared(A&&... a) { return std::make_shared<T>(std::forward<A>(a)...); }
}  // namespace boost

namespace gtsam {

typedef std::uint64_t Key;
typedef Eigen::MatrixXd Matrix;
typedef Eigen::VectorXd Vector;
inline std::string DefaultKeyFormatter(Key k) { return std::to_string(k); }

template <typename T> struct traits;
// VectorSpace traits of gtsam::Vector: Equals = equal_with_abs_tol (gtsam/base/Vector.cpp: sizes equal and every |a_i - b_i| <= tol)
template <> struct traits<Vector> {
  static bool Equals(const Vector& a, const Vector& b, double tol) {
    if (a.size() != b.size()) return false;
    for (int i = 0; i < a.size(); ++i) if (std::isnan(a(i)) != std::isnan(b(i)) || std::fabs(a(i) - b(i)) > tol) return false;
    return true;
  }
};

class Values {
 public:
  void insert(Key k, const Sophus::SE3f& p) { poses_[k] = p; }
  void insert(Key k, const Vector& v) { codes_[k] = v; }
  template <typename T> const T& at(Key k) const { return get(k, static_cast<const T*>(nullptr)); }
 private:
  const Sophus::SE3f& get(Key k, const Sophus::SE3f*) const { return poses_.at(k); }
  const Vector& get(Key k, const Vector*) const { return codes_.at(k); }
  std::map<Key, Sophus::SE3f> poses_;
  std::map<Key, Vector> codes_;
};

template <int N> struct KeyList {
  std::vector<Key> keys;
  KeyList& operator()(Key k) { keys.push_back(k); return *this; }
};
template <int N> KeyList<N> cref_list_of(Key k) { KeyList<N> l; l.keys.push_back(k); return l; }

class GaussianFactor {
 public:
  virtual ~GaussianFactor() {}
};

class NonlinearFactor {
 public:
  typedef boost::shared_ptr<NonlinearFactor> shared_ptr;
  NonlinearFactor() {}
  template <int N> explicit NonlinearFactor(const KeyList<N>& l) : keys_(l.keys) {}
  virtual ~NonlinearFactor() {}
  virtual double error(const Values& c) const = 0;
  virtual boost::shared_ptr<GaussianFactor> linearize(const Values& c) const = 0;
  virtual size_t dim() const = 0;
  virtual shared_ptr clone() const = 0;
  virtual bool active(const Values&) const { return true; }
  const std::vector<Key>& keys() const { return keys_; }
 protected:
  std::vector<Key> keys_;
};

}  // namespace gtsam
