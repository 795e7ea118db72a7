// The reference's fast_correlative_scan_matcher_2d.h with its public interface unchanged
// (SM2/fast_correlative_scan_matcher_2d.h:109-136) and the private section replaced by a
// handle of libcartographer_mi355x: what a maintainer edits to drop the MI355X matcher in.
// ConstraintBuilder2D (compiled UNMODIFIED from the reference tree) includes this header.
#ifndef DROPIN_FAST_CORRELATIVE_SCAN_MATCHER_2D_H_
#define DROPIN_FAST_CORRELATIVE_SCAN_MATCHER_2D_H_
#include "cartographer/mapping/2d/grid_2d.h"
#include "cartographer/mapping/proto/scan_matching/fast_correlative_scan_matcher_options_2d.pb.h"
#include "cartographer/sensor/point_cloud.h"
#include "cartographer/transform/rigid_transform.h"
#include "cartographer_mi355x.h"
namespace cartographer { namespace mapping { namespace scan_matching {
class FastCorrelativeScanMatcher2D {
 public:
  FastCorrelativeScanMatcher2D(const Grid2D& grid,
                               const proto::FastCorrelativeScanMatcherOptions2D& options);
  ~FastCorrelativeScanMatcher2D();
  FastCorrelativeScanMatcher2D(const FastCorrelativeScanMatcher2D&) = delete;
  FastCorrelativeScanMatcher2D& operator=(const FastCorrelativeScanMatcher2D&) = delete;
  bool Match(const transform::Rigid2d& initial_pose_estimate,
             const sensor::PointCloud& point_cloud, float min_score, float* score,
             transform::Rigid2d* pose_estimate) const;
  bool MatchFullSubmap(const sensor::PointCloud& point_cloud, float min_score, float* score,
                       transform::Rigid2d* pose_estimate) const;
 private:
  cmx_fast2d* handle_ = nullptr;
};
} } }
#endif  // DROPIN_FAST_CORRELATIVE_SCAN_MATCHER_2D_H_
