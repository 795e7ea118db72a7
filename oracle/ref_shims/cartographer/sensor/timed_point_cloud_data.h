// Stand-in for sensor/timed_point_cloud_data.h + the timed half of rangefinder_point.h: the
// types voxel_filter.h names in its overloads.
#ifndef ORACLE_REF_SHIMS_TIMED_POINT_CLOUD_DATA_H_
#define ORACLE_REF_SHIMS_TIMED_POINT_CLOUD_DATA_H_
#include <cstddef>
#include <vector>
#include "Eigen/Core"
namespace cartographer {
namespace sensor {
struct TimedRangefinderPoint {
  Eigen::Vector3f position;
  float time;
};
using TimedPointCloud = std::vector<TimedRangefinderPoint>;
struct TimedPointCloudOriginData {
  struct RangeMeasurement {
    TimedRangefinderPoint point_time;
    float intensity;
    size_t origin_index;
  };
};
}  // namespace sensor
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_TIMED_POINT_CLOUD_DATA_H_
