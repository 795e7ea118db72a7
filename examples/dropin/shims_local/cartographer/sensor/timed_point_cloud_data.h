// Stand-in for sensor/timed_point_cloud_data.h (+ the timed half of rangefinder_point.h) in the
// local-trajectory-builder build: the two structs with all their members
// (sensor/timed_point_cloud_data.h:27-45), without the proto conversions.
#ifndef DROPIN_SHIMS_LOCAL_TIMED_POINT_CLOUD_DATA_H_
#define DROPIN_SHIMS_LOCAL_TIMED_POINT_CLOUD_DATA_H_
#include <cstddef>
#include <map>      // the reference's header reaches these through protobuf
#include <set>
#include <string>
#include <vector>
#include "Eigen/Core"
#include "cartographer/common/time.h"
namespace cartographer {
namespace sensor {
struct TimedRangefinderPoint {
  Eigen::Vector3f position;
  float time;
};
using TimedPointCloud = std::vector<TimedRangefinderPoint>;
struct TimedPointCloudData {
  common::Time time;
  Eigen::Vector3f origin;
  TimedPointCloud ranges;
  std::vector<float> intensities;   // as long as `ranges`, or empty
};
struct TimedPointCloudOriginData {
  struct RangeMeasurement {
    TimedRangefinderPoint point_time;
    float intensity;
    size_t origin_index;
  };
  common::Time time;
  std::vector<Eigen::Vector3f> origins;
  std::vector<RangeMeasurement> ranges;
};
}  // namespace sensor
}  // namespace cartographer
#endif  // DROPIN_SHIMS_LOCAL_TIMED_POINT_CLOUD_DATA_H_
