// Stand-in for mapping/trajectory_node.h: of a node's constant data, the four members the 3D
// loop-closure matcher reads.
#ifndef ORACLE_REF_SHIMS_TRAJECTORY_NODE_H_
#define ORACLE_REF_SHIMS_TRAJECTORY_NODE_H_
#include "Eigen/Core"
#include "Eigen/Geometry"
#include "cartographer/common/lua_parameter_dictionary.h"   // reaches grid_2d.h this way upstream
#include "cartographer/sensor/point_cloud.h"
namespace cartographer { namespace mapping {
struct TrajectoryNode {
  struct Data {
    sensor::PointCloud high_resolution_point_cloud, low_resolution_point_cloud;
    Eigen::VectorXf rotational_scan_matcher_histogram;
    Eigen::Quaterniond gravity_alignment;
  };
};
} }
#endif  // ORACLE_REF_SHIMS_TRAJECTORY_NODE_H_
