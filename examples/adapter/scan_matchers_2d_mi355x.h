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
  // Uploads the grid and builds its precomputation stack in HBM (cmx_fast2d_create).
  FastCorrelativeScanMatcher2D(const Grid2D& finished_submap_grid,
                               const proto::FastCorrelativeScanMatcherOptions2D& opts);
  ~FastCorrelativeScanMatcher2D();
  FastCorrelativeScanMatcher2D(const FastCorrelativeScanMatcher2D&) = delete;
  FastCorrelativeScanMatcher2D& operator=(const FastCorrelativeScanMatcher2D&) = delete;

  // true iff some pose scores strictly above `threshold`; only then are *score and *pose
  // written.  Same argument order and types as the reference's methods.
  bool Match(const transform::Rigid2d& start, const sensor::PointCloud& cloud, float threshold,
             float* score, transform::Rigid2d* pose) const;
  bool MatchFullSubmap(const sensor::PointCloud& cloud, float threshold, float* score,
                       transform::Rigid2d* pose) const;

 private:
  cmx_fast2d* handle_ = nullptr;
};

class RealTimeCorrelativeScanMatcher2D {
 public:
  explicit RealTimeCorrelativeScanMatcher2D(
      const proto::RealTimeCorrelativeScanMatcherOptions& opts) : options_(opts) {}
  // Best weighted score over the exhaustive window around `start`; *pose always written.
  double Match(const transform::Rigid2d& start, const sensor::PointCloud& cloud,
               const Grid2D& active_grid, transform::Rigid2d* pose) const;

 private:
  const proto::RealTimeCorrelativeScanMatcherOptions options_;
};

}  // namespace scan_matching
}  // namespace mapping
}  // namespace cartographer

#endif  // EXAMPLES_ADAPTER_SCAN_MATCHERS_2D_MI355X_H_
