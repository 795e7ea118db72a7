// The reference's 2D scan-matcher classes with their public interfaces unchanged
// (SM2/fast_correlative_scan_matcher_2d.h:109-136, SM2/real_time_correlative_scan_matcher_2d.h:
// 48-83) and their bodies forwarding to libcartographer_mi355x.so.  Callers
// (LocalTrajectoryBuilder2D, ConstraintBuilder2D) compile against this header unchanged.
#ifndef EXAMPLES_ADAPTER_SCAN_MATCHERS_2D_MI355X_H_
#define EXAMPLES_ADAPTER_SCAN_MATCHERS_2D_MI355X_H_

#include "cartographer_mi355x.h"
#include "cartographer_standins.h"   // in-tree: the cartographer headers

namespace cartographer {
namespace mapping {
namespace scan_matching {

class FastCorrelativeScanMatcher2D {
 public:
  FastCorrelativeScanMatcher2D(const Grid2D& grid,
                               const proto::FastCorrelativeScanMatcherOptions2D& options);
  ~FastCorrelativeScanMatcher2D();
  FastCorrelativeScanMatcher2D(const FastCorrelativeScanMatcher2D&) = delete;
  FastCorrelativeScanMatcher2D& operator=(const FastCorrelativeScanMatcher2D&) = delete;

  // Returns true if a score above 'min_score' (excluding equality) is possible; then
  // 'score' and 'pose_estimate' are updated.
  bool Match(const transform::Rigid2d& initial_pose_estimate,
             const sensor::PointCloud& point_cloud, float min_score, float* score,
             transform::Rigid2d* pose_estimate) const;
  bool MatchFullSubmap(const sensor::PointCloud& point_cloud, float min_score, float* score,
                       transform::Rigid2d* pose_estimate) const;

 private:
  cmx_fast2d* handle_ = nullptr;
};

class RealTimeCorrelativeScanMatcher2D {
 public:
  explicit RealTimeCorrelativeScanMatcher2D(
      const proto::RealTimeCorrelativeScanMatcherOptions& options)
      : options_(options) {}
  double Match(const transform::Rigid2d& initial_pose_estimate,
               const sensor::PointCloud& point_cloud, const Grid2D& grid,
               transform::Rigid2d* pose_estimate) const;

 private:
  const proto::RealTimeCorrelativeScanMatcherOptions options_;
};

}  // namespace scan_matching
}  // namespace mapping
}  // namespace cartographer

#endif  // EXAMPLES_ADAPTER_SCAN_MATCHERS_2D_MI355X_H_
