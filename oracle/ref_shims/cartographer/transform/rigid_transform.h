// Stand-ins for transform::Rigid2<T> and transform::Rigid3<T> over the stand-in Eigen types.
// Only the members the translation units built by `make ref` use; the algebra is that of the
// reference's class (rigid_transform.h:118-196): inverse through the conjugate, a product that
// renormalises its rotation, rigid * point = rotation * point + translation.
#ifndef ORACLE_REF_SHIMS_RIGID_TRANSFORM_H_
#define ORACLE_REF_SHIMS_RIGID_TRANSFORM_H_
#include <array>
#include <cmath>
#include "Eigen/Core"
#include "Eigen/Geometry"
#include "cartographer/common/lua_parameter_dictionary.h"   // reaches includers this way upstream

namespace cartographer {
namespace transform {

template <typename S>
class Rigid2 {
 public:
  typedef Eigen::Matrix<S, 2, 1> Vector;
  typedef Eigen::Rotation2D<S> Rotation2D;

  Rigid2() {}
  Rigid2(const Vector& t, const Rotation2D& r) : t_(t), r_(r) {}
  Rigid2(const Vector& t, double angle) : t_(t), r_(angle) {}

  static Rigid2 Identity() { return Rigid2(); }
  static Rigid2 Translation(const Vector& t) { return Rigid2(t, Rotation2D()); }

  const Vector& translation() const { return t_; }
  Rotation2D rotation() const { return r_; }

  // rigid_transform.h:70-78.
  double normalized_angle() const {
    double a = r_.angle();            // common::NormalizeAngleDifference (common/math.h:69-75)
    const double two_pi = 2. * M_PI;
    while (a > M_PI) a -= two_pi;
    while (a < -M_PI) a += two_pi;
    return a;
  }
  template <typename U>
  Rigid2<U> cast() const {
    return Rigid2<U>(t_.template cast<U>(), r_.template cast<U>());
  }
  Rigid2 inverse() const {
    const Rotation2D rotation = r_.inverse();
    const Vector translation = -(rotation * t_);
    return Rigid2(translation, rotation);
  }

 private:
  Vector t_;
  Rotation2D r_;
};
// rigid_transform.h:90-97.
template <typename S>
Rigid2<S> operator*(const Rigid2<S>& lhs, const Rigid2<S>& rhs) {
  return Rigid2<S>(lhs.rotation() * rhs.translation() + lhs.translation(),
                   lhs.rotation() * rhs.rotation());
}
typedef Rigid2<double> Rigid2d;
typedef Rigid2<float> Rigid2f;

template <typename S>
class Rigid3 {
 public:
  typedef Eigen::Matrix<S, 3, 1> Vector;
  typedef Eigen::Quaternion<S> Quaternion;
  typedef Eigen::AngleAxis<S> AngleAxis;

  Rigid3() : t_(Vector::Zero()) {}                       // Quaternion() is the identity
  Rigid3(const Vector& t, const Quaternion& q) : t_(t), q_(q) {}
  Rigid3(const Vector& t, const AngleAxis& aa) : t_(t), q_(aa) {}

  static Rigid3 Identity() { return Rigid3(); }
  static Rigid3 Translation(const Vector& t) { return Rigid3(t, Quaternion()); }
  static Rigid3 Rotation(const Quaternion& q) { return Rigid3(Vector::Zero(), q); }
  static Rigid3 Rotation(const AngleAxis& aa) { return Rigid3(Vector::Zero(), Quaternion(aa)); }
  // rigid_transform.h:141-147.
  static Rigid3 FromArrays(const std::array<S, 4>& rotation, const std::array<S, 3>& translation) {
    return Rigid3(Vector(translation[0], translation[1], translation[2]),
                  Quaternion(rotation[0], rotation[1], rotation[2], rotation[3]));
  }

  const Vector& translation() const { return t_; }
  const Quaternion& rotation() const { return q_; }

  template <typename U>
  Rigid3<U> cast() const {
    return Rigid3<U>(t_.template cast<U>(), q_.template cast<U>());
  }

  Rigid3 inverse() const {
    const Quaternion back = q_.conjugate();
    return Rigid3(-(back * t_), back);
  }

  // rigid * point and rigid * rigid.
  Vector operator*(const Vector& p) const { return q_ * p + t_; }
  Rigid3 operator*(const Rigid3& rhs) const {
    return Rigid3(q_ * rhs.t_ + t_, (q_ * rhs.q_).normalized());
  }

 private:
  Vector t_;
  Quaternion q_;
};
typedef Rigid3<double> Rigid3d;
typedef Rigid3<float> Rigid3f;

}  // namespace transform
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_RIGID_TRANSFORM_H_
