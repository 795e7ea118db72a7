// Stand-in for the generated protobuf messages (ceres_scan_matcher_options_3d.proto).
#ifndef ORACLE_REF_SHIMS_CERES_SCAN_MATCHER_OPTIONS_3D_PB_H_
#define ORACLE_REF_SHIMS_CERES_SCAN_MATCHER_OPTIONS_3D_PB_H_
#include <vector>
#include "cartographer/common/proto/ceres_solver_options.pb.h"
namespace cartographer {
namespace mapping {
namespace scan_matching {
namespace proto {
class IntensityCostFunctionOptions {
 public:
  double weight() const { return weight_; }
  double huber_scale() const { return huber_scale_; }
  float intensity_threshold() const { return intensity_threshold_; }
  void set_weight(double v) { weight_ = v; }
  void set_huber_scale(double v) { huber_scale_ = v; }
  void set_intensity_threshold(float v) { intensity_threshold_ = v; }
 private:
  double weight_ = 0., huber_scale_ = 0.;
  float intensity_threshold_ = 0.f;
};
class CeresScanMatcherOptions3D {
 public:
  int occupied_space_weight_size() const { return static_cast<int>(occupied_.size()); }
  double occupied_space_weight(int i) const { return occupied_[i]; }
  void add_occupied_space_weight(double v) { occupied_.push_back(v); }
  int intensity_cost_function_options_size() const { return static_cast<int>(intensity_.size()); }
  const IntensityCostFunctionOptions& intensity_cost_function_options(int i) const {
    return intensity_[i];
  }
  IntensityCostFunctionOptions* add_intensity_cost_function_options() {
    intensity_.emplace_back();
    return &intensity_.back();
  }
  double translation_weight() const { return translation_weight_; }
  double rotation_weight() const { return rotation_weight_; }
  bool only_optimize_yaw() const { return only_optimize_yaw_; }
  void set_translation_weight(double v) { translation_weight_ = v; }
  void set_rotation_weight(double v) { rotation_weight_ = v; }
  void set_only_optimize_yaw(bool v) { only_optimize_yaw_ = v; }
  const common::proto::CeresSolverOptions& ceres_solver_options() const { return solver_; }
  common::proto::CeresSolverOptions* mutable_ceres_solver_options() { return &solver_; }
 private:
  std::vector<double> occupied_;
  std::vector<IntensityCostFunctionOptions> intensity_;
  double translation_weight_ = 0., rotation_weight_ = 0.;
  bool only_optimize_yaw_ = false;
  common::proto::CeresSolverOptions solver_;
};
}  // namespace proto
}  // namespace scan_matching
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_CERES_SCAN_MATCHER_OPTIONS_3D_PB_H_
