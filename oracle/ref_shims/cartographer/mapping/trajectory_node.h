// Stand-in for mapping/trajectory_node.h: the four members of TrajectoryNode::Data
// (:45-63) the 3D loop-closure matcher reads.
#ifndef ORACLE_REF_SHIMS_TRAJECTORY_NODE_H_
#define ORACLE_REF_SHIMS_TRAJECTORY_NODE_H_
#include "Eigen/Core"
#include "Eigen/Geometry"
#include "cartographer/common/lua_parameter_dictionary.h"   // reaches grid_2d.h this way in the real tree
#include "cartographer/sensor/point_cloud.h"
namespace cartographer {
namespace mapping {
struct TrajectoryNode {
  struct Data {
    Eigen::Quaterniond gravity_alignment;
    sensor::PointCloud high_resolution_point_cloud;
    sensor::PointCloud low_resolution_point_cloud;
    Eigen::VectorXf rotational_scan_matcher_histogram;
  };
};
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_TRAJECTORY_NODE_H_
