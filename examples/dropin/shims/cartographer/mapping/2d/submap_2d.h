// Stand-in for mapping/2d/submap_2d.h (+ mapping/submaps.h's Submap): what ConstraintBuilder2D
// reads of a submap -- its local pose and its grid.  Same constructor as the reference's
// (submap_2d.cc:65-71: the local pose is the translation by `origin`).
#ifndef DROPIN_SHIMS_SUBMAP_2D_H_
#define DROPIN_SHIMS_SUBMAP_2D_H_
#include <memory>
#include "Eigen/Core"
#include "cartographer/mapping/2d/grid_2d.h"
#include "cartographer/mapping/value_conversion_tables.h"
#include "cartographer/transform/rigid_transform.h"
namespace cartographer { namespace mapping {
class Submap2D {
 public:
  Submap2D(const Eigen::Vector2f& origin, std::unique_ptr<Grid2D> grid,
           ValueConversionTables* conversion_tables)
      : local_pose_(transform::Rigid3d::Translation(
            Eigen::Vector3d(origin.x(), origin.y(), 0.))),
        grid_(std::move(grid)), conversion_tables_(conversion_tables) {}
  transform::Rigid3d local_pose() const { return local_pose_; }
  const Grid2D* grid() const { return grid_.get(); }
 private:
  const transform::Rigid3d local_pose_;
  std::unique_ptr<Grid2D> grid_;
  ValueConversionTables* conversion_tables_;
};
} }
#endif  // DROPIN_SHIMS_SUBMAP_2D_H_
