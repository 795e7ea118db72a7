// Minimal stand-ins for the cartographer types the scan-matcher interfaces mention, so that
// the adapter classes in this directory compile and run without the reference tree.  Only
// the members the adapters touch exist; names and meaning follow the reference
// (transform/rigid_transform.h, sensor/point_cloud.h, mapping/2d/{map_limits,grid_2d}.h,
// the *_options_2d protos).  In a real integration these come from cartographer itself and
// this header is dropped.
#ifndef EXAMPLES_ADAPTER_CARTOGRAPHER_STANDINS_H_
#define EXAMPLES_ADAPTER_CARTOGRAPHER_STANDINS_H_

#include <cstdint>
#include <vector>

namespace cartographer {
namespace transform {
class Rigid2d {
 public:
  struct Vector { double x_, y_; double x() const { return x_; } double y() const { return y_; } };
  struct Rotation2D { double angle_; double angle() const { return angle_; } };
  Rigid2d() : translation_{0., 0.}, rotation_{0.} {}
  Rigid2d(const Vector& translation, double rotation)
      : translation_(translation), rotation_{rotation} {}
  const Vector& translation() const { return translation_; }
  Rotation2D rotation() const { return rotation_; }
 private:
  Vector translation_;
  Rotation2D rotation_;
};
}  // namespace transform

namespace sensor {
struct RangefinderPoint {
  struct Vector3f { float x_, y_, z_; float x() const { return x_; } float y() const { return y_; }
                    float z() const { return z_; } } position;
};
using PointCloud = std::vector<RangefinderPoint>;
}  // namespace sensor

namespace mapping {
struct CellLimits { int num_x_cells, num_y_cells; };
class MapLimits {
 public:
  MapLimits(double resolution, double max_x, double max_y, CellLimits cell_limits)
      : resolution_(resolution), max_x_(max_x), max_y_(max_y), cell_limits_(cell_limits) {}
  double resolution() const { return resolution_; }
  struct Max { double x_, y_; double x() const { return x_; } double y() const { return y_; } };
  Max max() const { return Max{max_x_, max_y_}; }
  const CellLimits& cell_limits() const { return cell_limits_; }
 private:
  double resolution_, max_x_, max_y_;
  CellLimits cell_limits_;
};
class Grid2D {   // ProbabilityGrid: limits + correspondence-cost cells (grid_2d.h:93-116)
 public:
  Grid2D(const MapLimits& limits, std::vector<uint16_t> cells)
      : limits_(limits), cells_(std::move(cells)) {}
  const MapLimits& limits() const { return limits_; }
  float GetMinCorrespondenceCost() const { return 1.f - (1.f - 0.1f); }
  float GetMaxCorrespondenceCost() const { return 1.f - 0.1f; }
  const std::vector<uint16_t>& correspondence_cost_cells() const { return cells_; }
 private:
  MapLimits limits_;
  std::vector<uint16_t> cells_;
};

namespace scan_matching {
namespace proto {
struct FastCorrelativeScanMatcherOptions2D {
  double linear_search_window_, angular_search_window_;
  int branch_and_bound_depth_;
  double linear_search_window() const { return linear_search_window_; }
  double angular_search_window() const { return angular_search_window_; }
  int branch_and_bound_depth() const { return branch_and_bound_depth_; }
};
struct RealTimeCorrelativeScanMatcherOptions {
  double linear_search_window_, angular_search_window_, translation_delta_cost_weight_,
      rotation_delta_cost_weight_;
  double linear_search_window() const { return linear_search_window_; }
  double angular_search_window() const { return angular_search_window_; }
  double translation_delta_cost_weight() const { return translation_delta_cost_weight_; }
  double rotation_delta_cost_weight() const { return rotation_delta_cost_weight_; }
};
}  // namespace proto
}  // namespace scan_matching
}  // namespace mapping
}  // namespace cartographer

#endif  // EXAMPLES_ADAPTER_CARTOGRAPHER_STANDINS_H_
