// Stand-in for the slice of GTSAM 4.0 (boost::shared_ptr era; the reference's thirdparty/gtsam submodule is empty) that
// core/gtsam/sparse_geometric_factor.{h,cpp} touch.  TEST INFRASTRUCTURE (oracle/_ref): data carriers only -- no arithmetic of the
// reference is restated here.  gtsam::Values holds poses (Sophus::SE3f) and codes (gtsam::Vector, double) by key; NonlinearFactor is the
// interface the factor overrides; boost::shared_ptr / make_shared map onto the std ones.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include <Eigen/Core>
#include <sophus/se3.hpp>

#ifndef EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#endif

namespace boost {
template <typename T> using shared_ptr = std::shared_ptr<T>;
template <typename T, typename... A> std::shared_ptr<T> make_shared(A&&... a) { return std::make_shared<T>(std::forward<A>(a)...); }
}  // namespace boost

namespace gtsam {

typedef std::uint64_t Key;
inline std::string DefaultKeyFormatter(Key k) { return std::to_string(k); }

// gtsam::Vector = Eigen::VectorXd: a dynamic double vector whose cast<S>() lands in the fixed-size matrices the factor assigns it to
class Vector {
 public:
  Vector() {}
  explicit Vector(const std::vector<double>& v) : v_(v) {}
  template <typename S> struct Cast {
    const std::vector<double>& v;
    template <int N> operator Eigen::Matrix<S, N, 1>() const { Eigen::Matrix<S, N, 1> m; for (int i = 0; i < N; ++i) m(i) = static_cast<S>(v[(size_t)i]); return m; }
  };
  template <typename S> Cast<S> cast() const { return Cast<S>{ v_ }; }
 private:
  std::vector<double> v_;
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
