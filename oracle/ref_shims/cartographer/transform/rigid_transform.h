// Stand-ins for transform::Rigid2<T> and Rigid3<T> (same algebra as
// transform/rigid_transform.h:34-103 and :118-196) over the stand-in Eigen types.
#ifndef ORACLE_REF_SHIMS_RIGID_TRANSFORM_H_
#define ORACLE_REF_SHIMS_RIGID_TRANSFORM_H_
#include "Eigen/Core"
#include "Eigen/Geometry"
#include "cartographer/common/lua_parameter_dictionary.h"   // the real header declares FromDictionary
namespace cartographer {
namespace transform {
template <typename FloatType>
class Rigid2 {
 public:
  using Vector = Eigen::Vector2<FloatType>;
  using Rotation2D = Eigen::Rotation2D<FloatType>;
  Rigid2() : translation_(), rotation_() {}
  Rigid2(const Vector& translation, const Rotation2D& rotation)
      : translation_(translation), rotation_(rotation) {}
  Rigid2(const Vector& translation, const double rotation)
      : translation_(translation), rotation_(rotation) {}
  static Rigid2 Translation(const Vector& vector) { return Rigid2(vector, Rotation2D()); }
  static Rigid2 Identity() { return Rigid2(); }
  const Vector& translation() const { return translation_; }
  Rotation2D rotation() const { return rotation_; }
 private:
  Vector translation_;
  Rotation2D rotation_;
};
using Rigid2d = Rigid2<double>;
using Rigid2f = Rigid2<float>;

// Rigid3<T>: same algebra as transform/rigid_transform.h:118-196 -- inverse() through the
// conjugate, the product renormalises its rotation, rigid * point = rotation * point +
// translation.
template <typename FloatType>
class Rigid3 {
 public:
  using Vector = Eigen::Matrix<FloatType, 3, 1>;
  using Quaternion = Eigen::Quaternion<FloatType>;
  using AngleAxis = Eigen::AngleAxis<FloatType>;
  Rigid3() : translation_(Vector::Zero()), rotation_(Quaternion::Identity()) {}
  Rigid3(const Vector& translation, const Quaternion& rotation)
      : translation_(translation), rotation_(rotation) {}
  Rigid3(const Vector& translation, const AngleAxis& rotation)
      : translation_(translation), rotation_(rotation) {}
  static Rigid3 Rotation(const AngleAxis& angle_axis) {
    return Rigid3(Vector::Zero(), Quaternion(angle_axis));
  }
  static Rigid3 Rotation(const Quaternion& rotation) { return Rigid3(Vector::Zero(), rotation); }
  static Rigid3 Translation(const Vector& vector) {
    return Rigid3(vector, Quaternion::Identity());
  }
  static Rigid3 Identity() { return Rigid3(); }
  template <typename OtherType>
  Rigid3<OtherType> cast() const {
    return Rigid3<OtherType>(translation_.template cast<OtherType>(),
                             rotation_.template cast<OtherType>());
  }
  const Vector& translation() const { return translation_; }
  const Quaternion& rotation() const { return rotation_; }
  Rigid3 inverse() const {
    const Quaternion rotation = rotation_.conjugate();
    const Vector translation = -(rotation * translation_);
    return Rigid3(translation, rotation);
  }
 private:
  Vector translation_;
  Quaternion rotation_;
};
template <typename FloatType>
Rigid3<FloatType> operator*(const Rigid3<FloatType>& lhs, const Rigid3<FloatType>& rhs) {
  return Rigid3<FloatType>(lhs.rotation() * rhs.translation() + lhs.translation(),
                           (lhs.rotation() * rhs.rotation()).normalized());
}
template <typename FloatType>
typename Rigid3<FloatType>::Vector operator*(const Rigid3<FloatType>& rigid,
                                             const typename Rigid3<FloatType>::Vector& point) {
  return rigid.rotation() * point + rigid.translation();
}
using Rigid3d = Rigid3<double>;
using Rigid3f = Rigid3<float>;
}  // namespace transform
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_RIGID_TRANSFORM_H_
