// The reference's 3D scan-matcher classes (SM3/fast_correlative_scan_matcher_3d.h:63-129,
// SM3/real_time_correlative_scan_matcher_3d.h:36-71) with their public interfaces unchanged and
// their bodies forwarding to libcartographer_mi355x.so.
#ifndef EXAMPLES_ADAPTER_SCAN_MATCHERS_3D_MI355X_H_
#define EXAMPLES_ADAPTER_SCAN_MATCHERS_3D_MI355X_H_

#include "cartographer_mi355x.h"
#include "cartographer_standins_3d.h"   // in-tree: the cartographer headers

namespace cartographer {
namespace mapping {
namespace scan_matching {

class FastCorrelativeScanMatcher3D {
 public:
  struct Result {
    float score;
    transform::Rigid3d pose_estimate;
    float rotational_score;
    float low_resolution_score;
  };

  FastCorrelativeScanMatcher3D(const HybridGrid& hybrid_grid,
                               const HybridGrid* low_resolution_hybrid_grid,
                               const std::vector<float>* rotational_scan_matcher_histogram,
                               const proto::FastCorrelativeScanMatcherOptions3D& options);
  ~FastCorrelativeScanMatcher3D();
  FastCorrelativeScanMatcher3D(const FastCorrelativeScanMatcher3D&) = delete;
  FastCorrelativeScanMatcher3D& operator=(const FastCorrelativeScanMatcher3D&) = delete;

  // nullptr when no candidate above 'min_score' passes the low-resolution check.
  std::unique_ptr<Result> Match(const transform::Rigid3d& global_node_pose,
                                const transform::Rigid3d& global_submap_pose,
                                const TrajectoryNodeData& constant_data, float min_score) const;
  std::unique_ptr<Result> MatchFullSubmap(const transform::Quaterniond& global_node_rotation,
                                          const transform::Quaterniond& global_submap_rotation,
                                          const TrajectoryNodeData& constant_data,
                                          float min_score) const;

 private:
  cmx_fast3d* handle_ = nullptr;
};

class RealTimeCorrelativeScanMatcher3D {
 public:
  explicit RealTimeCorrelativeScanMatcher3D(
      const proto::RealTimeCorrelativeScanMatcherOptions& options)
      : options_(options) {}
  float Match(const transform::Rigid3d& initial_pose_estimate,
              const sensor::PointCloud& point_cloud, const HybridGrid& hybrid_grid,
              transform::Rigid3d* pose_estimate) const;

 private:
  const proto::RealTimeCorrelativeScanMatcherOptions options_;
};

}  // namespace scan_matching
}  // namespace mapping
}  // namespace cartographer

#endif  // EXAMPLES_ADAPTER_SCAN_MATCHERS_3D_MI355X_H_
