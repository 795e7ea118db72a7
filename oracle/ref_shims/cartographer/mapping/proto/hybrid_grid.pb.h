// Stand-in for the generated protobuf message mapping::proto::HybridGrid: the real
// mapping/3d/hybrid_grid.h (compiled as it is) reads and writes it in HybridGrid(proto) /
// ToProto().  Parallel arrays of voxel indices and 16-bit values, like the .proto.
#ifndef ORACLE_REF_SHIMS_HYBRID_GRID_PB_H_
#define ORACLE_REF_SHIMS_HYBRID_GRID_PB_H_
#include <vector>
namespace cartographer {
namespace mapping {
namespace proto {
class HybridGrid {
 public:
  float resolution() const { return resolution_; }
  void set_resolution(float v) { resolution_ = v; }
  int values_size() const { return static_cast<int>(values_.size()); }
  int x_indices_size() const { return static_cast<int>(x_.size()); }
  int y_indices_size() const { return static_cast<int>(y_.size()); }
  int z_indices_size() const { return static_cast<int>(z_.size()); }
  int values(int i) const { return values_[i]; }
  int x_indices(int i) const { return x_[i]; }
  int y_indices(int i) const { return y_[i]; }
  int z_indices(int i) const { return z_[i]; }
  void add_values(int v) { values_.push_back(v); }
  void add_x_indices(int v) { x_.push_back(v); }
  void add_y_indices(int v) { y_.push_back(v); }
  void add_z_indices(int v) { z_.push_back(v); }
 private:
  float resolution_ = 0.f;
  std::vector<int> values_, x_, y_, z_;
};
}  // namespace proto
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_HYBRID_GRID_PB_H_
