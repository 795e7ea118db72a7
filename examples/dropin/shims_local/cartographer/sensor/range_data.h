// Stand-in for sensor/range_data.h (+ the helpers of rangefinder_point.h and point_cloud.h the
// local trajectory builder names): RangeData, TransformRangeData, CropRangeData
// (sensor/range_data.cc:25-40), ToRangefinderPoint and rigid * point
// (sensor/rangefinder_point.h:52-83), CropPointCloud (sensor/point_cloud.cc:77-82).
#ifndef DROPIN_SHIMS_LOCAL_RANGE_DATA_H_
#define DROPIN_SHIMS_LOCAL_RANGE_DATA_H_
#include "Eigen/Core"
#include "cartographer/common/port.h"
#include "cartographer/sensor/point_cloud.h"
#include "cartographer/sensor/timed_point_cloud_data.h"
#include "cartographer/transform/rigid_transform.h"
namespace cartographer {
namespace sensor {
struct RangeData {
  Eigen::Vector3f origin;
  PointCloud returns;
  PointCloud misses;
};
inline RangefinderPoint ToRangefinderPoint(const TimedRangefinderPoint& timed) {
  return {timed.position};
}
template <class T>
inline RangefinderPoint operator*(const transform::Rigid3<T>& lhs, const RangefinderPoint& rhs) {
  RangefinderPoint result = rhs;
  result.position = lhs * rhs.position;
  return result;
}
inline PointCloud CropPointCloud(const PointCloud& point_cloud, const float min_z,
                                 const float max_z) {
  return point_cloud.copy_if([min_z, max_z](const RangefinderPoint& point) {
    return min_z <= point.position.z() && point.position.z() <= max_z;
  });
}
inline RangeData TransformRangeData(const RangeData& range_data,
                                    const transform::Rigid3f& transform) {
  return RangeData{transform * range_data.origin,
                   TransformPointCloud(range_data.returns, transform),
                   TransformPointCloud(range_data.misses, transform)};
}
inline RangeData CropRangeData(const RangeData& range_data, const float min_z,
                               const float max_z) {
  return RangeData{range_data.origin, CropPointCloud(range_data.returns, min_z, max_z),
                   CropPointCloud(range_data.misses, min_z, max_z)};
}
}  // namespace sensor
}  // namespace cartographer
#endif  // DROPIN_SHIMS_LOCAL_RANGE_DATA_H_
