// Stand-in for mapping/3d/submap_3d.h (+ mapping/submaps.h's Submap): what ConstraintBuilder3D
// reads of a finished submap (constraint_builder_3d.cc:176-186) -- both hybrid grids (the
// reference's own HybridGrid, compiled from mapping/3d/hybrid_grid.h) and the histogram of its
// rotational scan matcher.  Same constructor shape as the reference's (submap_3d.h:46-50) plus
// mutable access for the fixture loader.
#ifndef DROPIN_SHIMS_SUBMAP_3D_H_
#define DROPIN_SHIMS_SUBMAP_3D_H_
#include <memory>
#include "Eigen/Core"
#include "cartographer/mapping/3d/hybrid_grid.h"
#include "cartographer/transform/rigid_transform.h"
namespace cartographer { namespace mapping {
class Submap3D {
 public:
  Submap3D(float high_resolution, float low_resolution, const transform::Rigid3d& local_submap_pose,
           const Eigen::VectorXf& rotational_scan_matcher_histogram)
      : local_pose_(local_submap_pose),
        high_resolution_hybrid_grid_(std::make_unique<HybridGrid>(high_resolution)),
        low_resolution_hybrid_grid_(std::make_unique<HybridGrid>(low_resolution)),
        rotational_scan_matcher_histogram_(rotational_scan_matcher_histogram) {}
  transform::Rigid3d local_pose() const { return local_pose_; }
  const HybridGrid& high_resolution_hybrid_grid() const { return *high_resolution_hybrid_grid_; }
  const HybridGrid& low_resolution_hybrid_grid() const { return *low_resolution_hybrid_grid_; }
  const Eigen::VectorXf& rotational_scan_matcher_histogram() const {
    return rotational_scan_matcher_histogram_;
  }
  HybridGrid* mutable_high_resolution_hybrid_grid() { return high_resolution_hybrid_grid_.get(); }
  HybridGrid* mutable_low_resolution_hybrid_grid() { return low_resolution_hybrid_grid_.get(); }
 private:
  const transform::Rigid3d local_pose_;
  std::unique_ptr<HybridGrid> high_resolution_hybrid_grid_, low_resolution_hybrid_grid_;
  Eigen::VectorXf rotational_scan_matcher_histogram_;
};
} }
#endif  // DROPIN_SHIMS_SUBMAP_3D_H_
