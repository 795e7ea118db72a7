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

  // Flattens both grids and builds the precomputation stack in HBM (cmx_fast3d_create).
  FastCorrelativeScanMatcher3D(const HybridGrid& high_resolution,
                               const HybridGrid* low_resolution,
                               const std::vector<float>* submap_histogram,
                               const proto::FastCorrelativeScanMatcherOptions3D& opts);
  ~FastCorrelativeScanMatcher3D();
  FastCorrelativeScanMatcher3D(const FastCorrelativeScanMatcher3D&) = delete;
  FastCorrelativeScanMatcher3D& operator=(const FastCorrelativeScanMatcher3D&) = delete;

  // nullptr unless a candidate scoring above `threshold` also passes the low-resolution check.
  std::unique_ptr<Result> Match(const transform::Rigid3d& node, const transform::Rigid3d& submap,
                                const TrajectoryNodeData& node_data, float threshold) const;
  std::unique_ptr<Result> MatchFullSubmap(const transform::Quaterniond& node_rotation,
                                          const transform::Quaterniond& submap_rotation,
                                          const TrajectoryNodeData& node_data,
                                          float threshold) const;

 private:
  cmx_fast3d* handle_ = nullptr;
};

class RealTimeCorrelativeScanMatcher3D {
 public:
  explicit RealTimeCorrelativeScanMatcher3D(
      const proto::RealTimeCorrelativeScanMatcherOptions& opts) : options_(opts) {}
  float Match(const transform::Rigid3d& start, const sensor::PointCloud& cloud,
              const HybridGrid& active_grid, transform::Rigid3d* pose) const;

 private:
  const proto::RealTimeCorrelativeScanMatcherOptions options_;
};

}  // namespace scan_matching
}  // namespace mapping
}  // namespace cartographer

#endif  // EXAMPLES_ADAPTER_SCAN_MATCHERS_3D_MI355X_H_
