// Stand-in for mapping/2d/submap_2d.h (+ mapping/submaps.h's Submap) in the
// local-trajectory-builder build: Submap2D and ActiveSubmaps2D with the members
// LocalTrajectoryBuilder2D and the scan matchers use.  The bookkeeping restates
// mapping/2d/submap_2d.cc:70-76,140-155,159-183,221-236 -- at most two submaps; a new one starts
// when the newest holds num_range_data scans; every scan goes into both; the older one is
// finished (cropped) at 2 * num_range_data -- over the reference's OWN ProbabilityGrid and
// ProbabilityGridRangeDataInserter2D, which this build compiles where they lie.  Left out: the
// proto round trip, the TSDF grid type, the submap texture.
#ifndef DROPIN_SHIMS_LOCAL_SUBMAP_2D_H_
#define DROPIN_SHIMS_LOCAL_SUBMAP_2D_H_
#include <memory>
#include <vector>
#include "Eigen/Core"
#include "absl/types/optional.h"   // reaches local_trajectory_builder_2d.h through Abseil upstream
#include "cartographer/mapping/2d/grid_2d.h"
#include "cartographer/mapping/2d/map_limits.h"
#include "cartographer/mapping/2d/probability_grid.h"
#include "cartographer/mapping/2d/probability_grid_range_data_inserter_2d.h"
#include "cartographer/mapping/proto/submaps_options_2d.pb.h"
#include "cartographer/mapping/trajectory_node.h"
#include "cartographer/mapping/value_conversion_tables.h"
#include "cartographer/sensor/range_data.h"
#include "cartographer/transform/rigid_transform.h"
namespace cartographer { namespace mapping {
inline void DropinSyncGridToHost(const Grid2D&) {}   // (resident/: downloads the device grid)

class Submap2D {
 public:
  Submap2D(const Eigen::Vector2f& origin, std::unique_ptr<Grid2D> grid,
           ValueConversionTables* conversion_tables)
      : local_pose_(transform::Rigid3d::Translation(
            Eigen::Vector3d(origin.x(), origin.y(), 0.))),
        grid_(std::move(grid)), conversion_tables_(conversion_tables) {}
  transform::Rigid3d local_pose() const { return local_pose_; }
  const Grid2D* grid() const { return grid_.get(); }
  int num_range_data() const { return num_range_data_; }
  bool insertion_finished() const { return insertion_finished_; }
  void InsertRangeData(const sensor::RangeData& range_data,
                       const RangeDataInserterInterface* range_data_inserter) {
    CHECK(grid_);
    CHECK(!insertion_finished_);
    range_data_inserter->Insert(range_data, grid_.get());
    ++num_range_data_;
  }
  void Finish() {
    CHECK(grid_);
    CHECK(!insertion_finished_);
    grid_ = grid_->ComputeCroppedGrid();
    insertion_finished_ = true;
  }
 private:
  const transform::Rigid3d local_pose_;
  std::unique_ptr<Grid2D> grid_;
  ValueConversionTables* conversion_tables_;
  int num_range_data_ = 0;
  bool insertion_finished_ = false;
};

class ActiveSubmaps2D {
 public:
  explicit ActiveSubmaps2D(const proto::SubmapsOptions2D& options)
      : options_(options),
        range_data_inserter_(options.probability_grid_range_data_inserter_options_2d()) {}
  ActiveSubmaps2D(const ActiveSubmaps2D&) = delete;
  ActiveSubmaps2D& operator=(const ActiveSubmaps2D&) = delete;

  std::vector<std::shared_ptr<const Submap2D>> submaps() const {
    return std::vector<std::shared_ptr<const Submap2D>>(submaps_.begin(), submaps_.end());
  }
  std::vector<std::shared_ptr<const Submap2D>> InsertRangeData(
      const sensor::RangeData& range_data) {
    if (submaps_.empty() || submaps_.back()->num_range_data() == options_.num_range_data()) {
      AddSubmap(range_data.origin.head<2>());
    }
    for (auto& submap : submaps_) submap->InsertRangeData(range_data, &range_data_inserter_);
    if (submaps_.front()->num_range_data() == 2 * options_.num_range_data()) {
      submaps_.front()->Finish();
    }
    return submaps();
  }
 private:
  void AddSubmap(const Eigen::Vector2f& origin) {
    if (submaps_.size() >= 2) {
      CHECK(submaps_.front()->insertion_finished());
      submaps_.erase(submaps_.begin());
    }
    constexpr int kInitialSubmapSize = 100;   // cells per side before the grid grows
    const float resolution = options_.grid_options_2d().resolution();
    const double half = 0.5 * kInitialSubmapSize * resolution;
    submaps_.push_back(std::make_shared<Submap2D>(
        origin,
        std::make_unique<ProbabilityGrid>(
            MapLimits(resolution, Eigen::Vector2d(origin.x() + half, origin.y() + half),
                      CellLimits(kInitialSubmapSize, kInitialSubmapSize)),
            &conversion_tables_),
        &conversion_tables_));
  }
  const proto::SubmapsOptions2D options_;
  std::vector<std::shared_ptr<Submap2D>> submaps_;
  ProbabilityGridRangeDataInserter2D range_data_inserter_;
  ValueConversionTables conversion_tables_;
};
} }
#endif  // DROPIN_SHIMS_LOCAL_SUBMAP_2D_H_
