// Stand-in for the generated protobuf options message.
#ifndef ORACLE_REF_SHIMS_FAST_CSM_OPTIONS_2D_PB_H_
#define ORACLE_REF_SHIMS_FAST_CSM_OPTIONS_2D_PB_H_
namespace cartographer {
namespace mapping {
namespace scan_matching {
namespace proto {
class FastCorrelativeScanMatcherOptions2D {
 public:
  double linear_search_window() const { return linear_search_window_; }
  double angular_search_window() const { return angular_search_window_; }
  int branch_and_bound_depth() const { return branch_and_bound_depth_; }
  void set_linear_search_window(double v) { linear_search_window_ = v; }
  void set_angular_search_window(double v) { angular_search_window_ = v; }
  void set_branch_and_bound_depth(int v) { branch_and_bound_depth_ = v; }
 private:
  double linear_search_window_ = 0., angular_search_window_ = 0.;
  int branch_and_bound_depth_ = 0;
};
}  // namespace proto
}  // namespace scan_matching
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_FAST_CSM_OPTIONS_2D_PB_H_
