// The reference's fast_correlative_scan_matcher_3d.h with its public interface unchanged
// (SM3/fast_correlative_scan_matcher_3d.h:66-101) and the private section replaced by a handle
// of libcartographer_mi355x.  ConstraintBuilder3D (compiled UNMODIFIED from the reference tree)
// includes this header.
#ifndef DROPIN_FAST_CORRELATIVE_SCAN_MATCHER_3D_H_
#define DROPIN_FAST_CORRELATIVE_SCAN_MATCHER_3D_H_
#include <memory>
#include "Eigen/Core"
#include "Eigen/Geometry"
#include "cartographer/mapping/3d/hybrid_grid.h"
#include "cartographer/mapping/proto/scan_matching/fast_correlative_scan_matcher_options_3d.pb.h"
#include "cartographer/mapping/trajectory_node.h"
#include "cartographer/sensor/point_cloud.h"
#include "cartographer/transform/rigid_transform.h"
#include "cartographer_mi355x.h"
namespace cartographer { namespace mapping { namespace scan_matching {
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
                               const Eigen::VectorXf* rotational_scan_matcher_histogram,
                               const proto::FastCorrelativeScanMatcherOptions3D& options);
  ~FastCorrelativeScanMatcher3D();
  FastCorrelativeScanMatcher3D(const FastCorrelativeScanMatcher3D&) = delete;
  FastCorrelativeScanMatcher3D& operator=(const FastCorrelativeScanMatcher3D&) = delete;
  std::unique_ptr<Result> Match(const transform::Rigid3d& global_node_pose,
                                const transform::Rigid3d& global_submap_pose,
                                const TrajectoryNode::Data& constant_data, float min_score) const;
  std::unique_ptr<Result> MatchFullSubmap(const Eigen::Quaterniond& global_node_rotation,
                                          const Eigen::Quaterniond& global_submap_rotation,
                                          const TrajectoryNode::Data& constant_data,
                                          float min_score) const;
 private:
  cmx_fast3d* handle_ = nullptr;
};
} } }
#endif  // DROPIN_FAST_CORRELATIVE_SCAN_MATCHER_3D_H_
