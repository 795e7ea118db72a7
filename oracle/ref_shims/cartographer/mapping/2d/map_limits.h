// Stand-in for mapping/2d/map_limits.h (the real header needs protobuf and Eigen's geometry
// module).  GetCellIndex / Contains restate map_limits.h:69-76, :85-90 -- the same two
// expressions the oracle restates and the reference's DiscretizeScans test pins.
#ifndef ORACLE_REF_SHIMS_MAP_LIMITS_H_
#define ORACLE_REF_SHIMS_MAP_LIMITS_H_
#include "Eigen/Core"
#include "cartographer/common/math.h"
#include "cartographer/common/port.h"
#include "cartographer/mapping/2d/xy_index.h"
#include "glog/logging.h"
namespace cartographer {
namespace mapping {
class MapLimits {
 public:
  MapLimits(const double resolution, const Eigen::Vector2d& max, const CellLimits& cell_limits)
      : resolution_(resolution), max_(max), cell_limits_(cell_limits) {}
  double resolution() const { return resolution_; }
  const Eigen::Vector2d& max() const { return max_; }
  const CellLimits& cell_limits() const { return cell_limits_; }
  Eigen::Array2i GetCellIndex(const Eigen::Vector2f& point) const {
    return Eigen::Array2i(common::RoundToInt((max_.y() - point.y()) / resolution_ - 0.5),
                          common::RoundToInt((max_.x() - point.x()) / resolution_ - 0.5));
  }
  bool Contains(const Eigen::Array2i& cell_index) const {
    return 0 <= cell_index.x() && cell_index.x() < cell_limits_.num_x_cells &&
           0 <= cell_index.y() && cell_index.y() < cell_limits_.num_y_cells;
  }
 private:
  double resolution_;
  Eigen::Vector2d max_;
  CellLimits cell_limits_;
};
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_MAP_LIMITS_H_
