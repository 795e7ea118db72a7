// The reference's ceres_scan_matcher_2d.h with its public interface unchanged
// (SM2/ceres_scan_matcher_2d.h:39-66); the body forwards to cmx_ceres2d_match.
#ifndef DROPIN_CERES_SCAN_MATCHER_2D_H_
#define DROPIN_CERES_SCAN_MATCHER_2D_H_
#include "Eigen/Core"
#include "cartographer/mapping/2d/grid_2d.h"
#include "cartographer/mapping/proto/scan_matching/ceres_scan_matcher_options_2d.pb.h"
#include "cartographer/sensor/point_cloud.h"
#include "cartographer/transform/rigid_transform.h"
#include "ceres/ceres.h"
namespace cartographer { namespace mapping { namespace scan_matching {
class CeresScanMatcher2D {
 public:
  explicit CeresScanMatcher2D(const proto::CeresScanMatcherOptions2D& options)
      : options_(options) {}
  CeresScanMatcher2D(const CeresScanMatcher2D&) = delete;
  CeresScanMatcher2D& operator=(const CeresScanMatcher2D&) = delete;
  void Match(const Eigen::Vector2d& target_translation,
             const transform::Rigid2d& initial_pose_estimate,
             const sensor::PointCloud& point_cloud, const Grid2D& grid,
             transform::Rigid2d* pose_estimate, ceres::Solver::Summary* summary) const;
 private:
  const proto::CeresScanMatcherOptions2D options_;
};
} } }
#endif  // DROPIN_CERES_SCAN_MATCHER_2D_H_
