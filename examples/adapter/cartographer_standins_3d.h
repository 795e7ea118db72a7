// Stand-ins for the cartographer 3D types the scan-matcher interfaces mention (see
// cartographer_standins.h).  HybridGrid is reduced to what the adapters read through its public
// interface: resolution(), grid_size() and the (cell index, value) pairs its Iterator yields
// (mapping/3d/hybrid_grid.h:304-372).
#ifndef EXAMPLES_ADAPTER_CARTOGRAPHER_STANDINS_3D_H_
#define EXAMPLES_ADAPTER_CARTOGRAPHER_STANDINS_3D_H_

#include <array>
#include <cstdint>
#include <memory>
#include <vector>

#include "cartographer_standins.h"

namespace cartographer {
namespace transform {
struct Quaterniond { double w_, x_, y_, z_; double w() const { return w_; } double x() const { return x_; }
                     double y() const { return y_; } double z() const { return z_; } };
class Rigid3d {
 public:
  struct Vector { double x_, y_, z_; double x() const { return x_; } double y() const { return y_; }
                  double z() const { return z_; } };
  Rigid3d() : translation_{0., 0., 0.}, rotation_{1., 0., 0., 0.} {}
  Rigid3d(const Vector& t, const Quaterniond& q) : translation_(t), rotation_(q) {}
  const Vector& translation() const { return translation_; }
  const Quaterniond& rotation() const { return rotation_; }
 private:
  Vector translation_;
  Quaterniond rotation_;
};
}  // namespace transform

namespace mapping {
class HybridGrid {
 public:
  struct Voxel { std::array<int, 3> index; uint16_t value; };
  HybridGrid(float resolution, int grid_size, std::vector<Voxel> voxels)
      : resolution_(resolution), grid_size_(grid_size), voxels_(std::move(voxels)) {}
  float resolution() const { return resolution_; }
  int grid_size() const { return grid_size_; }
  const std::vector<Voxel>& voxels() const { return voxels_; }   // stands in for Iterator
 private:
  float resolution_;
  int grid_size_;
  std::vector<Voxel> voxels_;
};

struct TrajectoryNodeData {              // TrajectoryNode::Data (mapping/trajectory_node.h:45-63)
  transform::Quaterniond gravity_alignment{1., 0., 0., 0.};
  sensor::PointCloud high_resolution_point_cloud;
  sensor::PointCloud low_resolution_point_cloud;
  std::vector<float> rotational_scan_matcher_histogram;   // Eigen::VectorXf
};

namespace scan_matching {
namespace proto {
struct FastCorrelativeScanMatcherOptions3D {
  int branch_and_bound_depth_, full_resolution_depth_;
  double min_rotational_score_, min_low_resolution_score_, linear_xy_search_window_,
      linear_z_search_window_, angular_search_window_;
  int branch_and_bound_depth() const { return branch_and_bound_depth_; }
  int full_resolution_depth() const { return full_resolution_depth_; }
  double min_rotational_score() const { return min_rotational_score_; }
  double min_low_resolution_score() const { return min_low_resolution_score_; }
  double linear_xy_search_window() const { return linear_xy_search_window_; }
  double linear_z_search_window() const { return linear_z_search_window_; }
  double angular_search_window() const { return angular_search_window_; }
};
}  // namespace proto
}  // namespace scan_matching
}  // namespace mapping
}  // namespace cartographer

#endif  // EXAMPLES_ADAPTER_CARTOGRAPHER_STANDINS_3D_H_
