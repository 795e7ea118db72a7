// Stand-ins for transform::Rigid2<T> (same algebra as transform/rigid_transform.h:34-103) and
// the one use of Rigid3f the 2D matchers make: Rigid3f::Rotation(AngleAxisf) applied to points.
#ifndef ORACLE_REF_SHIMS_RIGID_TRANSFORM_H_
#define ORACLE_REF_SHIMS_RIGID_TRANSFORM_H_
#include "Eigen/Core"
#include "Eigen/Geometry"
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

class Rigid3f {
 public:
  static Rigid3f Rotation(const Eigen::AngleAxisf& angle_axis) {
    Rigid3f r;
    r.rotation_ = Eigen::Quaternionf(angle_axis);
    return r;
  }
  // Rigid3 * point (rigid_transform.h:191-196): rotation * point + translation (zero here).
  Eigen::Vector3f operator*(const Eigen::Vector3f& point) const {
    const Eigen::Vector3f r = rotation_ * point;
    return Eigen::Vector3f(r.x() + 0.f, r.y() + 0.f, r.z() + 0.f);
  }
 private:
  Eigen::Quaternionf rotation_;
};
}  // namespace transform
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_RIGID_TRANSFORM_H_
