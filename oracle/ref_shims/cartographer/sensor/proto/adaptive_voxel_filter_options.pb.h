// Stand-in for the generated message of sensor/proto/adaptive_voxel_filter_options.proto:
// the three fields with the generated accessors' names.
#ifndef ORACLE_REF_SHIMS_ADAPTIVE_VOXEL_FILTER_OPTIONS_PB_H_
#define ORACLE_REF_SHIMS_ADAPTIVE_VOXEL_FILTER_OPTIONS_PB_H_
namespace cartographer {
namespace sensor {
namespace proto {
class AdaptiveVoxelFilterOptions {
 public:
  float max_length() const { return max_length_; }
  float min_num_points() const { return min_num_points_; }
  float max_range() const { return max_range_; }
  void set_max_length(float v) { max_length_ = v; }
  void set_min_num_points(float v) { min_num_points_ = v; }
  void set_max_range(float v) { max_range_ = v; }
 private:
  // (all three are `float` in the .proto, min_num_points included)
  float max_length_ = 0.f, min_num_points_ = 0.f, max_range_ = 0.f;
};
}  // namespace proto
}  // namespace sensor
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_ADAPTIVE_VOXEL_FILTER_OPTIONS_PB_H_
