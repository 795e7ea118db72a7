// Stand-in for the generated protobuf options message.
#ifndef ORACLE_REF_SHIMS_TSDF_INSERTER_OPTIONS_2D_PB_H_
#define ORACLE_REF_SHIMS_TSDF_INSERTER_OPTIONS_2D_PB_H_
#include "cartographer/mapping/proto/normal_estimation_options_2d.pb.h"
namespace cartographer {
namespace mapping {
namespace proto {
class TSDFRangeDataInserterOptions2D {
 public:
  double truncation_distance() const { return truncation_distance_; }
  double maximum_weight() const { return maximum_weight_; }
  bool update_free_space() const { return update_free_space_; }
  const NormalEstimationOptions2D& normal_estimation_options() const {
    return normal_estimation_options_;
  }
  NormalEstimationOptions2D* mutable_normal_estimation_options() {
    return &normal_estimation_options_;
  }
  bool project_sdf_distance_to_scan_normal() const { return project_sdf_distance_to_scan_normal_; }
  int update_weight_range_exponent() const { return update_weight_range_exponent_; }
  double update_weight_angle_scan_normal_to_ray_kernel_bandwidth() const {
    return update_weight_angle_scan_normal_to_ray_kernel_bandwidth_;
  }
  double update_weight_distance_cell_to_hit_kernel_bandwidth() const {
    return update_weight_distance_cell_to_hit_kernel_bandwidth_;
  }
  void set_truncation_distance(double v) { truncation_distance_ = v; }
  void set_maximum_weight(double v) { maximum_weight_ = v; }
  void set_update_free_space(bool v) { update_free_space_ = v; }
  void set_project_sdf_distance_to_scan_normal(bool v) { project_sdf_distance_to_scan_normal_ = v; }
  void set_update_weight_range_exponent(int v) { update_weight_range_exponent_ = v; }
  void set_update_weight_angle_scan_normal_to_ray_kernel_bandwidth(double v) {
    update_weight_angle_scan_normal_to_ray_kernel_bandwidth_ = v;
  }
  void set_update_weight_distance_cell_to_hit_kernel_bandwidth(double v) {
    update_weight_distance_cell_to_hit_kernel_bandwidth_ = v;
  }
 private:
  double truncation_distance_ = 0., maximum_weight_ = 0.;
  bool update_free_space_ = false, project_sdf_distance_to_scan_normal_ = false;
  NormalEstimationOptions2D normal_estimation_options_;
  int update_weight_range_exponent_ = 0;
  double update_weight_angle_scan_normal_to_ray_kernel_bandwidth_ = 0.,
         update_weight_distance_cell_to_hit_kernel_bandwidth_ = 0.;
};
}  // namespace proto
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_TSDF_INSERTER_OPTIONS_2D_PB_H_
