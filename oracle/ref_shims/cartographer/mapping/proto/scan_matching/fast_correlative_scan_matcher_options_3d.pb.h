// Stand-in for the generated protobuf options message.
#ifndef ORACLE_REF_SHIMS_FAST_CSM_OPTIONS_3D_PB_H_
#define ORACLE_REF_SHIMS_FAST_CSM_OPTIONS_3D_PB_H_
namespace cartographer {
namespace mapping {
namespace scan_matching {
namespace proto {
class FastCorrelativeScanMatcherOptions3D {
 public:
  int branch_and_bound_depth() const { return branch_and_bound_depth_; }
  int full_resolution_depth() const { return full_resolution_depth_; }
  double min_rotational_score() const { return min_rotational_score_; }
  double min_low_resolution_score() const { return min_low_resolution_score_; }
  double linear_xy_search_window() const { return linear_xy_search_window_; }
  double linear_z_search_window() const { return linear_z_search_window_; }
  double angular_search_window() const { return angular_search_window_; }
  void set_branch_and_bound_depth(int v) { branch_and_bound_depth_ = v; }
  void set_full_resolution_depth(int v) { full_resolution_depth_ = v; }
  void set_min_rotational_score(double v) { min_rotational_score_ = v; }
  void set_min_low_resolution_score(double v) { min_low_resolution_score_ = v; }
  void set_linear_xy_search_window(double v) { linear_xy_search_window_ = v; }
  void set_linear_z_search_window(double v) { linear_z_search_window_ = v; }
  void set_angular_search_window(double v) { angular_search_window_ = v; }
 private:
  int branch_and_bound_depth_ = 0, full_resolution_depth_ = 0;
  double min_rotational_score_ = 0., min_low_resolution_score_ = 0.,
         linear_xy_search_window_ = 0., linear_z_search_window_ = 0.,
         angular_search_window_ = 0.;
};
}  // namespace proto
}  // namespace scan_matching
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_FAST_CSM_OPTIONS_3D_PB_H_
