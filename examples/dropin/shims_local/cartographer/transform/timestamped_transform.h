// Stand-in for transform/timestamped_transform.h: the struct (timestamped_transform.h:27-30) and
// a FromProto over an empty stand-in message -- LocalTrajectoryBuilder3D::AddImuData converts
// options.initial_poses(), which this build leaves empty.
#ifndef DROPIN_SHIMS_LOCAL_TIMESTAMPED_TRANSFORM_H_
#define DROPIN_SHIMS_LOCAL_TIMESTAMPED_TRANSFORM_H_
#include "cartographer/common/time.h"
#include "cartographer/transform/rigid_transform.h"
namespace cartographer {
namespace transform {
namespace proto {
struct TimestampedTransform {};
}  // namespace proto
struct TimestampedTransform {
  common::Time time;
  transform::Rigid3d transform;
};
inline TimestampedTransform FromProto(const proto::TimestampedTransform&) {
  return TimestampedTransform{common::Time::min(), Rigid3d::Identity()};
}
}  // namespace transform
}  // namespace cartographer
#endif  // DROPIN_SHIMS_LOCAL_TIMESTAMPED_TRANSFORM_H_
