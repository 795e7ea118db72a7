// Stand-in for the generated protobuf message (ceres_scan_matcher_options_2d.proto).
#ifndef ORACLE_REF_SHIMS_CERES_SCAN_MATCHER_OPTIONS_2D_PB_H_
#define ORACLE_REF_SHIMS_CERES_SCAN_MATCHER_OPTIONS_2D_PB_H_
#include "cartographer/common/proto/ceres_solver_options.pb.h"
namespace cartographer {
namespace mapping {
namespace scan_matching {
namespace proto {
class CeresScanMatcherOptions2D {
 public:
  double occupied_space_weight() const { return occupied_space_weight_; }
  double translation_weight() const { return translation_weight_; }
  double rotation_weight() const { return rotation_weight_; }
  void set_occupied_space_weight(double v) { occupied_space_weight_ = v; }
  void set_translation_weight(double v) { translation_weight_ = v; }
  void set_rotation_weight(double v) { rotation_weight_ = v; }
  const common::proto::CeresSolverOptions& ceres_solver_options() const { return solver_; }
  common::proto::CeresSolverOptions* mutable_ceres_solver_options() { return &solver_; }
 private:
  double occupied_space_weight_ = 0., translation_weight_ = 0., rotation_weight_ = 0.;
  common::proto::CeresSolverOptions solver_;
};
}  // namespace proto
}  // namespace scan_matching
}  // namespace mapping
}  // namespace cartographer
#endif  // ORACLE_REF_SHIMS_CERES_SCAN_MATCHER_OPTIONS_2D_PB_H_
