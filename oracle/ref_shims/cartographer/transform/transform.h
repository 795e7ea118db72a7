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

// Rotation angle of a rigid transform (transform.h:33-37): 2 atan2(|q.vec|, |q.w|).
template <typename S>
S GetAngle(const Rigid3<S>& rigid) {
  const Eigen::Quaternion<S>& q = rigid.rotation();
  const S sine_half = q.vec().norm();
  return S(2) * std::atan2(sine_half, std::abs(q.w()));
}

// Yaw of a rotation (transform.h:42-47): heading of the rotated x axis in the xy plane.
template <typename S>
S GetYaw(const Eigen::Quaternion<S>& q) {
  const Eigen::Matrix<S, 3, 1> heading = q * Eigen::Matrix<S, 3, 1>::UnitX();
  return std::atan2(heading.y(), heading.x());
}
template <typename S>
S GetYaw(const Rigid3<S>& rigid) { return GetYaw(rigid.rotation()); }

// transform.h:101-115.
template <typename S>
Rigid2<S> Project2D(const Rigid3<S>& transform) {
  return Rigid2<S>(transform.translation().template head<2>(), GetYaw(transform));
}
template <typename S>
Rigid3<S> Embed3D(const Rigid2<S>& transform) {
  return Rigid3<S>(Eigen::Matrix<S, 3, 1>(transform.translation().x(),
                                          transform.translation().y(), S(0)),
                   Eigen::AngleAxis<S>(transform.rotation().angle(),
                                       Eigen::Matrix<S, 3, 1>::UnitZ()));
}

// Angle-axis vector -> quaternion (transform.h:85-99).  Below |v|^2 = 1e-8 the reference
// linearises (w = 1, xyz = v / 2); above, sin and cos of |v| / 2 are evaluated in DOUBLE
// whatever S is (its literal `2.` promotes) and narrowed to S afterwards.
template <typename S>
Eigen::Quaternion<S> AngleAxisVectorToRotationQuaternion(const Eigen::Matrix<S, 3, 1>& v) {
  S w = S(1), k = S(0.5);
  if (v.squaredNorm() > 1e-8) {
    const S length = v.norm();
    k = std::sin(length / 2.) / length;
    w = std::cos(length / 2.);
  }
  const Eigen::Matrix<S, 3, 1> xyz = k * v;
  return Eigen::Quaternion<S>(w, xyz.x(), xyz.y(), xyz.z());
}

// Proto conversions the grid headers call (transform.cc).
inline Eigen::Vector2d ToEigen(const proto::Vector2d& v) { return Eigen::Vector2d(v.x(), v.y()); }
inline proto::Vector2d ToProto(const Eigen::Vector2d& v) {
  proto::Vector2d out;
  out.set_x(v.x());
  out.set_y(v.y());
  return out;
}
inline proto::Rigid3d ToProto(const Rigid3d& rigid) {
  proto::Rigid3d out;
  const auto& t = rigid.translation();
  const auto& q = rigid.rotation();
  out.t[0] = t.x(); out.t[1] = t.y(); out.t[2] = t.z();
  out.q[0] = q.w(); out.q[1] = q.x(); out.q[2] = q.y(); out.q[3] = q.z();
  return out;
}

}  // namespace transform
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_CARTOGRAPHER_TRANSFORM_TRANSFORM_H_
