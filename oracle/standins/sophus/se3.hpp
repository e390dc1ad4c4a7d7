// Minimal stand-in for Sophus::SO3 / Sophus::SE3 (strasdat/Sophus, the submodule the reference pins; SHA not recoverable).
// TEST INFRASTRUCTURE (oracle/_ref).  The arithmetic restates Sophus' published formulas in the template Scalar:
//   SO3 * point   p + w * uv + vec x uv,  uv = 2 (vec x p)                       (sophus/so3.hpp, operator*(Point))
//   SO3 * SO3     explicit quaternion product, then renormalised                   (operator*(SO3Base), SO3(Quaternion) ctor)
//   matrix()      Eigen's Quaternion::toRotationMatrix()
//   exp           quaternion (sin(theta/2)/theta * w, cos(theta/2)) with the Taylor branch below 1e-10 (so3.hpp expAndTheta)
//   SE3           T * p = R p + t;  inverse() = (R^-1, R^-1 * (-t));  A * B = (Ra Rb, ta + Ra tb)   (sophus/se3.hpp)
#pragma once
#include <Eigen/Core>

namespace Sophus {

template <typename T>
class SO3 {
 public:
  typedef T Scalar;
  typedef Eigen::Matrix<T, 3, 1> Point;
  typedef Eigen::Matrix<T, 3, 1> Tangent;
  typedef Eigen::Matrix<T, 3, 3> Transformation;
  static constexpr int DoF = 3;

  struct Quaternion {   // Eigen::Quaternion subset: x y z w accessors, coeffs(), vec()
    T x_, y_, z_, w_;
    T x() const { return x_; } T y() const { return y_; } T z() const { return z_; } T w() const { return w_; }
    Point vec() const { return Point(x_, y_, z_); }
    struct Coeffs { T v[4]; T operator()(int i) const { return v[i]; } T operator[](int i) const { return v[i]; } };
    Coeffs coeffs() const { return Coeffs{ { x_, y_, z_, w_ } }; }
  };

  SO3() : q_{ T(0), T(0), T(0), T(1) } {}
  // from quaternion components; normalises like Sophus' SO3(Quaternion) constructor
  SO3(T x, T y, T z, T w) : q_{ x, y, z, w } { normalize(); }
  static SO3 fromUnit(T x, T y, T z, T w) { SO3 r; r.q_ = Quaternion{ x, y, z, w }; return r; }

  const Quaternion& unit_quaternion() const { return q_; }

  void normalize() {
    using std::sqrt;
    const T n = sqrt(q_.x_ * q_.x_ + q_.y_ * q_.y_ + q_.z_ * q_.z_ + q_.w_ * q_.w_);
    q_.x_ /= n; q_.y_ /= n; q_.z_ /= n; q_.w_ /= n;
  }

  Transformation matrix() const {
    const T tx = T(2) * q_.x_, ty = T(2) * q_.y_, tz = T(2) * q_.z_;
    const T twx = tx * q_.w_, twy = ty * q_.w_, twz = tz * q_.w_;
    const T txx = tx * q_.x_, txy = ty * q_.x_, txz = tz * q_.x_;
    const T tyy = ty * q_.y_, tyz = tz * q_.y_, tzz = tz * q_.z_;
    Transformation R;
    R(0, 0) = T(1) - (tyy + tzz); R(0, 1) = txy - twz;          R(0, 2) = txz + twy;
    R(1, 0) = txy + twz;          R(1, 1) = T(1) - (txx + tzz); R(1, 2) = tyz - twx;
    R(2, 0) = txz - twy;          R(2, 1) = tyz + twx;          R(2, 2) = T(1) - (txx + tyy);
    return R;
  }

  SO3 inverse() const { return fromUnit(-q_.x_, -q_.y_, -q_.z_, q_.w_); }

  template <typename D>
  Point operator*(const Eigen::MatrixBase<D>& p_in) const {
    const Point p(p_in);
    const Point v = q_.vec();
    Point uv = v.cross(p);
    uv += uv;
    return p + uv * q_.w_ + v.cross(uv);
  }

  SO3 operator*(const SO3& o) const {
    const Quaternion &a = q_, &b = o.q_;
    return SO3(a.w_ * b.x_ + a.x_ * b.w_ + a.y_ * b.z_ - a.z_ * b.y_,
               a.w_ * b.y_ + a.y_ * b.w_ + a.z_ * b.x_ - a.x_ * b.z_,
               a.w_ * b.z_ + a.z_ * b.w_ + a.x_ * b.y_ - a.y_ * b.x_,
               a.w_ * b.w_ - a.x_ * b.x_ - a.y_ * b.y_ - a.z_ * b.z_);
  }

  static Transformation hat(const Tangent& o) {
    Transformation O;
    O(0, 0) = T(0);  O(0, 1) = -o(2); O(0, 2) = o(1);
    O(1, 0) = o(2);  O(1, 1) = T(0);  O(1, 2) = -o(0);
    O(2, 0) = -o(1); O(2, 1) = o(0);  O(2, 2) = T(0);
    return O;
  }

  static SO3 exp(const Tangent& omega) {
    using std::sqrt; using std::sin; using std::cos;
    const T theta_sq = omega.squaredNorm();
    T imag, real;
    if (theta_sq < T(1e-10) * T(1e-10)) {
      const T theta_po4 = theta_sq * theta_sq;
      imag = T(0.5) - T(1.0 / 48.0) * theta_sq + T(1.0 / 3840.0) * theta_po4;
      real = T(1) - T(1.0 / 8.0) * theta_sq + T(1.0 / 384.0) * theta_po4;
    } else {
      const T theta = sqrt(theta_sq), half = T(0.5) * theta;
      imag = sin(half) / theta;
      real = cos(half);
    }
    return fromUnit(imag * omega(0), imag * omega(1), imag * omega(2), real);
  }

  Tangent log() const {
    using std::sqrt; using std::atan; using std::abs;
    const T sn = q_.x_ * q_.x_ + q_.y_ * q_.y_ + q_.z_ * q_.z_;
    const T w = q_.w_;
    T two_atan_nbyw_by_n;
    if (sn < T(1e-10) * T(1e-10)) {
      two_atan_nbyw_by_n = T(2) / w - T(2.0 / 3.0) * sn / (w * w * w);
    } else {
      const T n = sqrt(sn);
      if (abs(w) < T(1e-10)) two_atan_nbyw_by_n = (w > T(0) ? T(M_PI) : -T(M_PI)) / n;
      else two_atan_nbyw_by_n = T(2) * atan(n / w) / n;
    }
    return Tangent(two_atan_nbyw_by_n * q_.x_, two_atan_nbyw_by_n * q_.y_, two_atan_nbyw_by_n * q_.z_);
  }

 private:
  Quaternion q_;
};

template <typename T>
class SE3 {
 public:
  typedef T Scalar;
  typedef SO3<T> SO3Type;
  typedef Eigen::Matrix<T, 3, 1> Point;
  typedef Eigen::Matrix<T, 3, 1> TranslationType;
  typedef Eigen::Matrix<T, 6, 1> Tangent;
  static constexpr int DoF = 6;

  SE3() {}
  SE3(const SO3Type& so3, const Point& t) : so3_(so3), t_(t) {}

  const SO3Type& so3() const { return so3_; }
  SO3Type& so3() { return so3_; }
  const TranslationType& translation() const { return t_; }
  TranslationType& translation() { return t_; }
  const typename SO3Type::Quaternion& unit_quaternion() const { return so3_.unit_quaternion(); }
  typename SO3Type::Transformation rotationMatrix() const { return so3_.matrix(); }

  SE3 inverse() const {
    const SO3Type invR = so3_.inverse();
    return SE3(invR, invR * (t_ * T(-1)));
  }
  SE3 operator*(const SE3& o) const { return SE3(so3_ * o.so3_, t_ + so3_ * o.t_); }
  // log: (V^-1 t, omega) with V^-1 = I - 1/2 Omega + (1 - theta cos(theta/2) / (2 sin(theta/2))) / theta^2 Omega^2   (sophus/se3.hpp)
  Tangent log() const {
    using std::sqrt; using std::sin; using std::cos; using std::abs;
    const typename SO3Type::Tangent w = so3_.log();
    const T theta = sqrt(w.squaredNorm());
    const typename SO3Type::Transformation Om = SO3Type::hat(w);
    typename SO3Type::Transformation Vinv = SO3Type::Transformation::Identity() - Om * T(0.5);
    if (abs(theta) < T(1e-10)) Vinv = Vinv + (Om * Om) * T(1.0 / 12.0);
    else { const T half = T(0.5) * theta; Vinv = Vinv + (Om * Om) * ((T(1) - theta * cos(half) / (T(2) * sin(half))) / (theta * theta)); }
    const Point u = Vinv * t_;
    Tangent r;
    r(0) = u(0); r(1) = u(1); r(2) = u(2); r(3) = w(0); r(4) = w(1); r(5) = w(2);
    return r;
  }
  template <typename D> Point operator*(const Eigen::MatrixBase<D>& p) const { return so3_ * p + t_; }

 private:
  SO3Type so3_;
  TranslationType t_;
};

typedef SE3<float> SE3f;
typedef SE3<double> SE3d;
typedef SO3<float> SO3f;

}  // namespace Sophus
