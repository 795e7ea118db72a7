// Stand-in for sensor/imu_data.h: the struct (imu_data.h:27-31) without its proto conversions.
#ifndef DROPIN_SHIMS_LOCAL_IMU_DATA_H_
#define DROPIN_SHIMS_LOCAL_IMU_DATA_H_
#include "Eigen/Core"
#include "cartographer/common/time.h"
namespace cartographer {
namespace sensor {
namespace proto {
struct ImuData {};   // options.initial_imu_data() of the 3D builder: left empty in this build
}  // namespace proto
struct ImuData {
  common::Time time;
  Eigen::Vector3d linear_acceleration;
  Eigen::Vector3d angular_velocity;
};
inline ImuData FromProto(const proto::ImuData&) {
  return ImuData{common::Time::min(), Eigen::Vector3d::Zero(), Eigen::Vector3d::Zero()};
}
}  // namespace sensor
}  // namespace cartographer
#endif  // DROPIN_SHIMS_LOCAL_IMU_DATA_H_
