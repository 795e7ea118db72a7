// Shadows the reference's transform/transform.h for `make ref`: the files built there include
// it for Eigen types, the Rigid transforms and glog's CHECK macros; the real header needs
// Eigen's geometry module.
#ifndef ORACLE_REF_SHIMS_CARTOGRAPHER_TRANSFORM_TRANSFORM_H_
#define ORACLE_REF_SHIMS_CARTOGRAPHER_TRANSFORM_TRANSFORM_H_
#include <algorithm>
#include <cmath>
#include "Eigen/Core"
#include "Eigen/Geometry"
#include "cartographer/common/math.h"   // the reference's own (int64, Clamp, ...)
#include "cartographer/common/port.h"
#include "cartographer/transform/proto/transform.pb.h"
#include "cartographer/transform/rigid_transform.h"
#include "glog/logging.h"

namespace cartographer {
namespace transform {

// transform/transform.h:33-37.
template <typename FloatType>
FloatType GetAngle(const Rigid3<FloatType>& transform) {
  return FloatType(2) * std::atan2(transform.rotation().vec().norm(),
                                   std::abs(transform.rotation().w()));
}

// transform/transform.h:42-47: the direction the rotation sends UnitX to.
template <typename T>
T GetYaw(const Eigen::Quaternion<T>& rotation) {
  const Eigen::Matrix<T, 3, 1> direction = rotation * Eigen::Matrix<T, 3, 1>::UnitX();
  return std::atan2(direction.y(), direction.x());
}
template <typename T>
T GetYaw(const Rigid3<T>& transform) {
  return GetYaw(transform.rotation());
}

// transform/transform.h:85-99: sin/cos of `norm / 2.` are evaluated in double whatever T is
// (the literal 2. promotes), then narrowed to T.
template <typename T>
Eigen::Quaternion<T> AngleAxisVectorToRotationQuaternion(
    const Eigen::Matrix<T, 3, 1>& angle_axis) {
  T scale = T(0.5);
  T w = T(1.);
  constexpr double kCutoffAngle = 1e-8;   // linearised below this angle
  if (angle_axis.squaredNorm() > kCutoffAngle) {
    const T norm = angle_axis.norm();
    scale = std::sin(norm / 2.) / norm;
    w = std::cos(norm / 2.);
  }
  const Eigen::Matrix<T, 3, 1> quaternion_xyz = scale * angle_axis;
  return Eigen::Quaternion<T>(w, quaternion_xyz.x(), quaternion_xyz.y(), quaternion_xyz.z());
}

// transform/transform.cc:40-42,94-99 and :117-127: conversions the grid headers call.
inline Eigen::Vector2d ToEigen(const proto::Vector2d& vector) {
  return Eigen::Vector2d(vector.x(), vector.y());
}
inline proto::Vector2d ToProto(const Eigen::Vector2d& vector) {
  proto::Vector2d result;
  result.set_x(vector.x());
  result.set_y(vector.y());
  return result;
}
inline proto::Rigid3d ToProto(const Rigid3d& rigid) {
  proto::Rigid3d result;
  result.t[0] = rigid.translation().x(); result.t[1] = rigid.translation().y();
  result.t[2] = rigid.translation().z();
  result.q[0] = rigid.rotation().w(); result.q[1] = rigid.rotation().x();
  result.q[2] = rigid.rotation().y(); result.q[3] = rigid.rotation().z();
  return result;
}

}  // namespace transform
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_CARTOGRAPHER_TRANSFORM_TRANSFORM_H_
