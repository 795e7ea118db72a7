// Stand-in for mapping/trajectory_node.h: a node's constant data with every member, in the
// reference's order (trajectory_node.h:45-63) -- LocalTrajectoryBuilder2D::InsertIntoSubmap
// brace-initialises it.
#ifndef DROPIN_SHIMS_LOCAL_TRAJECTORY_NODE_H_
#define DROPIN_SHIMS_LOCAL_TRAJECTORY_NODE_H_
#include "Eigen/Core"
#include "Eigen/Geometry"
#include "cartographer/common/lua_parameter_dictionary.h"
#include "cartographer/common/time.h"
#include "cartographer/sensor/point_cloud.h"
#include "cartographer/transform/rigid_transform.h"
namespace cartographer { namespace mapping {
struct TrajectoryNode {
  struct Data {
    common::Time time;
    Eigen::Quaterniond gravity_alignment;
    sensor::PointCloud filtered_gravity_aligned_point_cloud;
    sensor::PointCloud high_resolution_point_cloud;
    sensor::PointCloud low_resolution_point_cloud;
    Eigen::VectorXf rotational_scan_matcher_histogram;
    transform::Rigid3d local_pose;
  };
};
} }
#endif  // DROPIN_SHIMS_LOCAL_TRAJECTORY_NODE_H_
