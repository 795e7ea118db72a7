// Stand-in for the generated message of pose_graph/constraint_builder_options.proto: the
// fields ConstraintBuilder2D / 3D read, with the generated accessors' names.
#ifndef DROPIN_SHIMS_CONSTRAINT_BUILDER_OPTIONS_PB_H_
#define DROPIN_SHIMS_CONSTRAINT_BUILDER_OPTIONS_PB_H_
#include "cartographer/mapping/proto/scan_matching/ceres_scan_matcher_options_2d.pb.h"
#include "cartographer/mapping/proto/scan_matching/ceres_scan_matcher_options_3d.pb.h"
#include "cartographer/mapping/proto/scan_matching/fast_correlative_scan_matcher_options_2d.pb.h"
#include "cartographer/mapping/proto/scan_matching/fast_correlative_scan_matcher_options_3d.pb.h"
namespace cartographer { namespace mapping { namespace constraints { namespace proto {
struct ConstraintBuilderOptions {
  double sampling_ratio_ = 1., max_constraint_distance_ = 0., min_score_ = 0.,
         global_localization_min_score_ = 0., loop_closure_translation_weight_ = 0.,
         loop_closure_rotation_weight_ = 0.;
  bool log_matches_ = false;
  scan_matching::proto::FastCorrelativeScanMatcherOptions2D fast_;
  scan_matching::proto::CeresScanMatcherOptions2D ceres_;
  scan_matching::proto::FastCorrelativeScanMatcherOptions3D fast_3d_;
  scan_matching::proto::CeresScanMatcherOptions3D ceres_3d_;
  double sampling_ratio() const { return sampling_ratio_; }
  double max_constraint_distance() const { return max_constraint_distance_; }
  double min_score() const { return min_score_; }
  double global_localization_min_score() const { return global_localization_min_score_; }
  double loop_closure_translation_weight() const { return loop_closure_translation_weight_; }
  double loop_closure_rotation_weight() const { return loop_closure_rotation_weight_; }
  bool log_matches() const { return log_matches_; }
  const scan_matching::proto::FastCorrelativeScanMatcherOptions2D&
  fast_correlative_scan_matcher_options() const { return fast_; }
  const scan_matching::proto::CeresScanMatcherOptions2D& ceres_scan_matcher_options() const {
    return ceres_;
  }
  const scan_matching::proto::FastCorrelativeScanMatcherOptions3D&
  fast_correlative_scan_matcher_options_3d() const { return fast_3d_; }
  const scan_matching::proto::CeresScanMatcherOptions3D& ceres_scan_matcher_options_3d() const {
    return ceres_3d_;
  }
};
} } } }
#endif  // DROPIN_SHIMS_CONSTRAINT_BUILDER_OPTIONS_PB_H_
