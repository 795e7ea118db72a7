// Stand-in for sensor/range_data.h: one scan as the inserters take it -- where the sensor was
// and which points it saw (returns) or failed to see within range (misses), map frame.
#ifndef ORACLE_REF_SHIMS_RANGE_DATA_H_
#define ORACLE_REF_SHIMS_RANGE_DATA_H_
#include "Eigen/Core"
#include "cartographer/common/port.h"
#include "cartographer/sensor/point_cloud.h"
namespace cartographer { namespace sensor {
struct RangeData { Eigen::Vector3f origin; PointCloud returns, misses; };
} }
#endif  // ORACLE_REF_SHIMS_RANGE_DATA_H_
