// Stand-in for the generated message of ceres_scan_matcher_options_3d.proto: the fields
// CeresScanMatcher3D reads, with the generated accessors' names.
#ifndef DROPIN_SHIMS_CERES_SCAN_MATCHER_OPTIONS_3D_PB_H_
#define DROPIN_SHIMS_CERES_SCAN_MATCHER_OPTIONS_3D_PB_H_
#include <vector>
#include "cartographer/mapping/proto/scan_matching/ceres_scan_matcher_options_2d.pb.h"
namespace cartographer { namespace mapping { namespace scan_matching { namespace proto {
struct IntensityCostFunctionOptions {       // ceres_scan_matcher_options_3d.proto:21-25
  double weight_ = 0., huber_scale_ = 0., intensity_threshold_ = 0.;
  double weight() const { return weight_; }
  double huber_scale() const { return huber_scale_; }
  double intensity_threshold() const { return intensity_threshold_; }
  void set_weight(double v) { weight_ = v; }
  void set_huber_scale(double v) { huber_scale_ = v; }
  void set_intensity_threshold(double v) { intensity_threshold_ = v; }
};
struct CeresScanMatcherOptions3D {
  std::vector<double> occupied_space_weight_;
  std::vector<IntensityCostFunctionOptions> intensity_cost_function_options_;
  int intensity_cost_function_options_size() const {
    return static_cast<int>(intensity_cost_function_options_.size());
  }
  const IntensityCostFunctionOptions& intensity_cost_function_options(int i) const {
    return intensity_cost_function_options_.at(i);
  }
  IntensityCostFunctionOptions* add_intensity_cost_function_options() {
    intensity_cost_function_options_.emplace_back();
    return &intensity_cost_function_options_.back();
  }
  double translation_weight_ = 0., rotation_weight_ = 0.;
  bool only_optimize_yaw_ = false;
  common::proto::CeresSolverOptions solver_;
  int occupied_space_weight_size() const { return static_cast<int>(occupied_space_weight_.size()); }
  double occupied_space_weight(int i) const { return occupied_space_weight_.at(i); }
  void add_occupied_space_weight(double v) { occupied_space_weight_.push_back(v); }
  double translation_weight() const { return translation_weight_; }
  double rotation_weight() const { return rotation_weight_; }
  bool only_optimize_yaw() const { return only_optimize_yaw_; }
  const common::proto::CeresSolverOptions& ceres_solver_options() const { return solver_; }
  common::proto::CeresSolverOptions* mutable_ceres_solver_options() { return &solver_; }
  void set_translation_weight(double v) { translation_weight_ = v; }
  void set_rotation_weight(double v) { rotation_weight_ = v; }
  void set_only_optimize_yaw(bool v) { only_optimize_yaw_ = v; }
};
} } } }
#endif  // DROPIN_SHIMS_CERES_SCAN_MATCHER_OPTIONS_3D_PB_H_
