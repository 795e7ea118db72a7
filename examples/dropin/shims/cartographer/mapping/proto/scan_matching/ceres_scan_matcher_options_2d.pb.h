// Stand-in for the generated message of ceres_scan_matcher_options_2d.proto (+ the nested
// common.proto.CeresSolverOptions): fields with the generated accessors' names.
#ifndef DROPIN_SHIMS_CERES_SCAN_MATCHER_OPTIONS_2D_PB_H_
#define DROPIN_SHIMS_CERES_SCAN_MATCHER_OPTIONS_2D_PB_H_
namespace cartographer {
namespace common { namespace proto {
struct CeresSolverOptions {
  bool use_nonmonotonic_steps_ = false;
  int max_num_iterations_ = 0, num_threads_ = 1;
  bool use_nonmonotonic_steps() const { return use_nonmonotonic_steps_; }
  int max_num_iterations() const { return max_num_iterations_; }
  int num_threads() const { return num_threads_; }
  void set_use_nonmonotonic_steps(bool v) { use_nonmonotonic_steps_ = v; }
  void set_max_num_iterations(int v) { max_num_iterations_ = v; }
  void set_num_threads(int v) { num_threads_ = v; }
};
} }
namespace mapping { namespace scan_matching { namespace proto {
struct CeresScanMatcherOptions2D {
  double occupied_space_weight_ = 0., translation_weight_ = 0., rotation_weight_ = 0.;
  common::proto::CeresSolverOptions solver_;
  double occupied_space_weight() const { return occupied_space_weight_; }
  double translation_weight() const { return translation_weight_; }
  double rotation_weight() const { return rotation_weight_; }
  const common::proto::CeresSolverOptions& ceres_solver_options() const { return solver_; }
  common::proto::CeresSolverOptions* mutable_ceres_solver_options() { return &solver_; }
  void set_occupied_space_weight(double v) { occupied_space_weight_ = v; }
  void set_translation_weight(double v) { translation_weight_ = v; }
  void set_rotation_weight(double v) { rotation_weight_ = v; }
};
} } }
}  // namespace cartographer
#endif  // DROPIN_SHIMS_CERES_SCAN_MATCHER_OPTIONS_2D_PB_H_
