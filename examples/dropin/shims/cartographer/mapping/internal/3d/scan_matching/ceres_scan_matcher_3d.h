// The reference's ceres_scan_matcher_3d.h with its public interface unchanged
// (SM3/ceres_scan_matcher_3d.h:36-66); the body forwards to cmx_ceres3d_match.
#ifndef DROPIN_CERES_SCAN_MATCHER_3D_H_
#define DROPIN_CERES_SCAN_MATCHER_3D_H_
#include <vector>
#include "Eigen/Core"
#include "cartographer/mapping/3d/hybrid_grid.h"
#include "cartographer/mapping/proto/scan_matching/ceres_scan_matcher_options_3d.pb.h"
#include "cartographer/sensor/point_cloud.h"
#include "cartographer/transform/rigid_transform.h"
#include "ceres/ceres.h"
namespace cartographer { namespace mapping { namespace scan_matching {
struct PointCloudAndHybridGridsPointers {
  const sensor::PointCloud* point_cloud;
  const HybridGrid* hybrid_grid;
  const IntensityHybridGrid* intensity_hybrid_grid;  // optional
};
class CeresScanMatcher3D {
 public:
  explicit CeresScanMatcher3D(const proto::CeresScanMatcherOptions3D& options)
      : options_(options) {}
  CeresScanMatcher3D(const CeresScanMatcher3D&) = delete;
  CeresScanMatcher3D& operator=(const CeresScanMatcher3D&) = delete;
  void Match(const Eigen::Vector3d& target_translation,
             const transform::Rigid3d& initial_pose_estimate,
             const std::vector<PointCloudAndHybridGridsPointers>& point_clouds_and_hybrid_grids,
             transform::Rigid3d* pose_estimate, ceres::Solver::Summary* summary) const;
 private:
  const proto::CeresScanMatcherOptions3D options_;
};
} } }
#endif  // DROPIN_CERES_SCAN_MATCHER_3D_H_
