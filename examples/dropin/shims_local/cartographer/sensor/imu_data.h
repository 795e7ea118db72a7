// Stand-in for sensor/imu_data.h: the struct (imu_data.h:27-31) without its proto conversions.
#ifndef DROPIN_SHIMS_LOCAL_IMU_DATA_H_
#define DROPIN_SHIMS_LOCAL_IMU_DATA_H_
#include "Eigen/Core"
#include "cartographer/common/time.h"
namespace cartographer {
namespace sensor {
struct ImuData {
  common::Time time;
  Eigen::Vector3d linear_acceleration;
  Eigen::Vector3d angular_velocity;
};
}  // namespace sensor
}  // namespace cartographer
#endif  // DROPIN_SHIMS_LOCAL_IMU_DATA_H_
