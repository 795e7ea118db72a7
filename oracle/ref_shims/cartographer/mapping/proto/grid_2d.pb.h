// Stand-in for the generated protobuf message mapping::proto::Grid2D: the real Grid2D /
// ProbabilityGrid / TSDF2D classes (compiled as they are) are constructed from it and serialise
// into it, which is how oracle/ref_wrapper.cc hands raw uint16 cells in and reads them back.
#ifndef ORACLE_REF_SHIMS_GRID_2D_PB_H_
#define ORACLE_REF_SHIMS_GRID_2D_PB_H_
#include <vector>
#include "cartographer/mapping/proto/map_limits.pb.h"
namespace cartographer {
namespace mapping {
namespace proto {
class ProbabilityGrid {};
class TSDF2D {
 public:
  float truncation_distance() const { return truncation_distance_; }
  float max_weight() const { return max_weight_; }
  void set_truncation_distance(float v) { truncation_distance_ = v; }
  void set_max_weight(float v) { max_weight_ = v; }
  int weight_cells_size() const { return static_cast<int>(weight_cells_.size()); }
  const std::vector<int>& weight_cells() const { return weight_cells_; }
  std::vector<int>* mutable_weight_cells() { return &weight_cells_; }
 private:
  float truncation_distance_ = 0.f, max_weight_ = 0.f;
  std::vector<int> weight_cells_;
};
class Grid2D {
 public:
  class CellBox {
   public:
    int max_x() const { return max_x_; }
    int max_y() const { return max_y_; }
    int min_x() const { return min_x_; }
    int min_y() const { return min_y_; }
    void set_max_x(int v) { max_x_ = v; }
    void set_max_y(int v) { max_y_ = v; }
    void set_min_x(int v) { min_x_ = v; }
    void set_min_y(int v) { min_y_ = v; }
   private:
    int max_x_ = 0, max_y_ = 0, min_x_ = 0, min_y_ = 0;
  };
  const MapLimits& limits() const { return limits_; }
  MapLimits* mutable_limits() { return &limits_; }
  int cells_size() const { return static_cast<int>(cells_.size()); }
  const std::vector<int>& cells() const { return cells_; }
  std::vector<int>* mutable_cells() { return &cells_; }
  bool has_known_cells_box() const { return has_known_cells_box_; }
  const CellBox& known_cells_box() const { return known_cells_box_; }
  CellBox* mutable_known_cells_box() {
    has_known_cells_box_ = true;
    return &known_cells_box_;
  }
  float min_correspondence_cost() const { return min_correspondence_cost_; }
  float max_correspondence_cost() const { return max_correspondence_cost_; }
  void set_min_correspondence_cost(float v) { min_correspondence_cost_ = v; }
  void set_max_correspondence_cost(float v) { max_correspondence_cost_ = v; }
  bool has_probability_grid_2d() const { return has_probability_grid_2d_; }
  ProbabilityGrid* mutable_probability_grid_2d() {
    has_probability_grid_2d_ = true;
    return &probability_grid_2d_;
  }
  bool has_tsdf_2d() const { return has_tsdf_2d_; }
  const TSDF2D& tsdf_2d() const { return tsdf_2d_; }
  TSDF2D* mutable_tsdf_2d() {
    has_tsdf_2d_ = true;
    return &tsdf_2d_;
  }
 private:
  MapLimits limits_;
  std::vector<int> cells_;
  bool has_known_cells_box_ = false, has_probability_grid_2d_ = false, has_tsdf_2d_ = false;
  CellBox known_cells_box_;
  float min_correspondence_cost_ = 0.f, max_correspondence_cost_ = 0.f;
  ProbabilityGrid probability_grid_2d_;
  TSDF2D tsdf_2d_;
};
}  // namespace proto
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_GRID_2D_PB_H_
