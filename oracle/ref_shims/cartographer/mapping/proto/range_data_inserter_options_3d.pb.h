// Stand-in for the generated protobuf options message.
#ifndef ORACLE_REF_SHIMS_RANGE_DATA_INSERTER_OPTIONS_3D_PB_H_
#define ORACLE_REF_SHIMS_RANGE_DATA_INSERTER_OPTIONS_3D_PB_H_
namespace cartographer {
namespace mapping {
namespace proto {
class RangeDataInserterOptions3D {
 public:
  double hit_probability() const { return hit_probability_; }
  double miss_probability() const { return miss_probability_; }
  int num_free_space_voxels() const { return num_free_space_voxels_; }
  double intensity_threshold() const { return intensity_threshold_; }
  void set_hit_probability(double v) { hit_probability_ = v; }
  void set_miss_probability(double v) { miss_probability_ = v; }
  void set_num_free_space_voxels(int v) { num_free_space_voxels_ = v; }
  void set_intensity_threshold(double v) { intensity_threshold_ = v; }
 private:
  double hit_probability_ = 0., miss_probability_ = 0., intensity_threshold_ = 0.;
  int num_free_space_voxels_ = 0;
};
}  // namespace proto
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_RANGE_DATA_INSERTER_OPTIONS_3D_PB_H_
