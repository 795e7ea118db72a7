// Stand-in for the generated protobuf message mapping::proto::MapLimits.
#ifndef ORACLE_REF_SHIMS_MAP_LIMITS_PB_H_
#define ORACLE_REF_SHIMS_MAP_LIMITS_PB_H_
#include "cartographer/mapping/proto/cell_limits_2d.pb.h"
#include "cartographer/transform/proto/transform.pb.h"
namespace cartographer {
namespace mapping {
namespace proto {
class MapLimits {
 public:
  double resolution() const { return resolution_; }
  void set_resolution(double v) { resolution_ = v; }
  const transform::proto::Vector2d& max() const { return max_; }
  transform::proto::Vector2d* mutable_max() { return &max_; }
  const CellLimits& cell_limits() const { return cell_limits_; }
  CellLimits* mutable_cell_limits() { return &cell_limits_; }
 private:
  double resolution_ = 0.;
  transform::proto::Vector2d max_;
  CellLimits cell_limits_;
};
}  // namespace proto
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_MAP_LIMITS_PB_H_
