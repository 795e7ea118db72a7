// Stand-in for the generated protobuf options message.
#ifndef ORACLE_REF_SHIMS_RT_CSM_OPTIONS_PB_H_
#define ORACLE_REF_SHIMS_RT_CSM_OPTIONS_PB_H_
namespace cartographer {
namespace mapping {
namespace scan_matching {
namespace proto {
class RealTimeCorrelativeScanMatcherOptions {
 public:
  double linear_search_window() const { return linear_search_window_; }
  double angular_search_window() const { return angular_search_window_; }
  double translation_delta_cost_weight() const { return translation_delta_cost_weight_; }
  double rotation_delta_cost_weight() const { return rotation_delta_cost_weight_; }
  void set_linear_search_window(double v) { linear_search_window_ = v; }
  void set_angular_search_window(double v) { angular_search_window_ = v; }
  void set_translation_delta_cost_weight(double v) { translation_delta_cost_weight_ = v; }
  void set_rotation_delta_cost_weight(double v) { rotation_delta_cost_weight_ = v; }
 private:
  double linear_search_window_ = 0., angular_search_window_ = 0.,
         translation_delta_cost_weight_ = 0., rotation_delta_cost_weight_ = 0.;
};
}  // namespace proto
}  // namespace scan_matching
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_RT_CSM_OPTIONS_PB_H_
