// Stand-in for sensor/range_data.h: the struct (range_data.h:30-34) without its proto helpers.
#ifndef ORACLE_REF_SHIMS_RANGE_DATA_H_
#define ORACLE_REF_SHIMS_RANGE_DATA_H_
#include "Eigen/Core"
#include "cartographer/common/port.h"
#include "cartographer/sensor/point_cloud.h"
namespace cartographer {
namespace sensor {
struct RangeData {
  Eigen::Vector3f origin;
  PointCloud returns;
  PointCloud misses;
};
}  // namespace sensor
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_RANGE_DATA_H_
