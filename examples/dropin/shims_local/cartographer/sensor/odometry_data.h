// Stand-in for sensor/odometry_data.h: the struct (odometry_data.h:27-30) without its proto
// conversions.
#ifndef DROPIN_SHIMS_LOCAL_ODOMETRY_DATA_H_
#define DROPIN_SHIMS_LOCAL_ODOMETRY_DATA_H_
#include "cartographer/common/time.h"
#include "cartographer/transform/rigid_transform.h"
namespace cartographer {
namespace sensor {
struct OdometryData {
  common::Time time;
  transform::Rigid3d pose;
};
}  // namespace sensor
}  // namespace cartographer
#endif  // DROPIN_SHIMS_LOCAL_ODOMETRY_DATA_H_
