// Submap2D / ActiveSubmaps2D with the probability grid RESIDENT IN HBM: the third build of the
// local-trajectory-builder test.  Same bookkeeping as shims_local/.../submap_2d.h (which restates
// mapping/2d/submap_2d.cc:70-76,140-155,159-183,221-236), but the grid is a cmx_grid2d:
// InsertRangeData is cmx_grid2d_insert (the reference's ProbabilityGridRangeDataInserter2D +
// FinishUpdate on the device, bit for bit), Finish is cmx_grid2d_crop, and grid() hands the
// matchers a Grid2D that is also a dropin::DeviceGrid2DView, so that the adapters take the
// *_match_grid entry points: per scan only the point clouds cross PCIe.
#ifndef DROPIN_RESIDENT_SUBMAP_2D_H_
#define DROPIN_RESIDENT_SUBMAP_2D_H_
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>
#include "Eigen/Core"
#include "absl/types/optional.h"
#include "cartographer/mapping/2d/grid_2d.h"
#include "cartographer/mapping/2d/map_limits.h"
#include "cartographer/mapping/probability_values.h"
#include "cartographer/mapping/proto/submaps_options_2d.pb.h"
#include "cartographer/mapping/trajectory_node.h"
#include "cartographer/mapping/value_conversion_tables.h"
#include "cartographer/sensor/range_data.h"
#include "cartographer/transform/rigid_transform.h"
#include "device_grids.h"
namespace cartographer { namespace mapping {

inline void DropinCheckOk(cmx_status status, const char* what) {
  if (status == CMX_OK) return;
  std::fprintf(stderr, "Check failed: %s: %s (%s)\n", what, cmx_status_string(status),
               cmx_last_error());
  std::abort();
}

// What the matchers are handed: a Grid2D with the device grid's CURRENT limits and no cells of
// its own until SyncToHost() is asked for them (the test main's digest).
class DeviceGrid2D : public Grid2D, public dropin::DeviceGrid2DView {
 public:
  DeviceGrid2D(const cmx_grid2d* grid, const cmx_grid2d_limits& l, ValueConversionTables* tables)
      : Grid2D(MapLimits(l.resolution, Eigen::Vector2d(l.max_x, l.max_y),
                         CellLimits(l.num_x_cells, l.num_y_cells)),
               kMinCorrespondenceCost, kMaxCorrespondenceCost, tables),
        grid_(grid) {}
  const cmx_grid2d* device_grid() const override { return grid_; }
  GridType GetGridType() const override { return GridType::PROBABILITY_GRID; }
  std::unique_ptr<Grid2D> ComputeCroppedGrid() const override {
    std::fprintf(stderr, "DeviceGrid2D is cropped through its submap (cmx_grid2d_crop)\n");
    std::abort();
  }
  bool DrawToSubmapTexture(proto::SubmapQuery::Response::SubmapTexture*,
                           transform::Rigid3d) const override {
    return false;
  }
  void SyncToHost() {
    DropinCheckOk(cmx_grid2d_download(grid_, mutable_correspondence_cost_cells()->data()),
                  "cmx_grid2d_download");
  }
 private:
  const cmx_grid2d* grid_;
};

inline void DropinSyncGridToHost(const Grid2D& grid) {
  if (const auto* device = dynamic_cast<const DeviceGrid2D*>(&grid))
    const_cast<DeviceGrid2D*>(device)->SyncToHost();
}

class Submap2D {
 public:
  Submap2D(const Eigen::Vector2f& origin, const MapLimits& limits, const int device,
           ValueConversionTables* conversion_tables)
      : local_pose_(transform::Rigid3d::Translation(
            Eigen::Vector3d(origin.x(), origin.y(), 0.))),
        conversion_tables_(conversion_tables) {
    const cmx_grid2d_limits l{limits.resolution(), limits.max().x(), limits.max().y(),
                              limits.cell_limits().num_x_cells, limits.cell_limits().num_y_cells,
                              kMinCorrespondenceCost, kMaxCorrespondenceCost};
    DropinCheckOk(cmx_grid2d_create(&l, nullptr, device, &device_grid_), "cmx_grid2d_create");
    RefreshView();
  }
  ~Submap2D() { cmx_grid2d_destroy(device_grid_); }
  Submap2D(const Submap2D&) = delete;
  Submap2D& operator=(const Submap2D&) = delete;

  transform::Rigid3d local_pose() const { return local_pose_; }
  const Grid2D* grid() const { return view_.get(); }
  int num_range_data() const { return num_range_data_; }
  bool insertion_finished() const { return insertion_finished_; }

  void InsertRangeData(const sensor::RangeData& range_data,
                       const proto::ProbabilityGridRangeDataInserterOptions2D& options) {
    CHECK(!insertion_finished_);
    const float origin[2] = {range_data.origin.x(), range_data.origin.y()};
    const std::vector<float> returns = Flatten(range_data.returns);
    const std::vector<float> misses = Flatten(range_data.misses);
    DropinCheckOk(cmx_grid2d_insert(device_grid_, origin, returns.data(),
                                    static_cast<int32_t>(range_data.returns.size()), misses.data(),
                                    static_cast<int32_t>(range_data.misses.size()),
                                    static_cast<float>(options.hit_probability()),
                                    static_cast<float>(options.miss_probability()),
                                    options.insert_free_space() ? 1 : 0),
                  "cmx_grid2d_insert");
    RefreshView();
    ++num_range_data_;
  }
  void Finish() {
    CHECK(!insertion_finished_);
    DropinCheckOk(cmx_grid2d_crop(device_grid_), "cmx_grid2d_crop");
    RefreshView();
    insertion_finished_ = true;
  }

 private:
  static std::vector<float> Flatten(const sensor::PointCloud& cloud) {
    std::vector<float> xyz;
    xyz.reserve(3 * cloud.size());
    for (const sensor::RangefinderPoint& p : cloud) {
      xyz.push_back(p.position.x());
      xyz.push_back(p.position.y());
      xyz.push_back(p.position.z());
    }
    return xyz;
  }
  // A new view only when the limits moved (the grid grew or was cropped): scans of a known area
  // leave the matchers' object alone.
  void RefreshView() {
    cmx_grid2d_limits l{};
    DropinCheckOk(cmx_grid2d_get_limits(device_grid_, &l), "cmx_grid2d_get_limits");
    if (view_ != nullptr) {
      const MapLimits& have = view_->limits();
      if (have.cell_limits().num_x_cells == l.num_x_cells &&
          have.cell_limits().num_y_cells == l.num_y_cells && have.max().x() == l.max_x &&
          have.max().y() == l.max_y)
        return;
    }
    view_ = std::make_unique<DeviceGrid2D>(device_grid_, l, conversion_tables_);
  }
  const transform::Rigid3d local_pose_;
  ValueConversionTables* conversion_tables_;
  cmx_grid2d* device_grid_ = nullptr;
  std::unique_ptr<DeviceGrid2D> view_;
  int num_range_data_ = 0;
  bool insertion_finished_ = false;
};

class ActiveSubmaps2D {
 public:
  explicit ActiveSubmaps2D(const proto::SubmapsOptions2D& options) : options_(options) {
    const char* e = std::getenv("CMX_DEVICE");
    device_ = e ? std::atoi(e) : 0;
  }
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
    for (auto& submap : submaps_)
      submap->InsertRangeData(range_data,
                              options_.probability_grid_range_data_inserter_options_2d());
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
    constexpr int kInitialSubmapSize = 100;
    const float resolution = options_.grid_options_2d().resolution();
    const double half = 0.5 * kInitialSubmapSize * resolution;
    submaps_.push_back(std::make_shared<Submap2D>(
        origin,
        MapLimits(resolution, Eigen::Vector2d(origin.x() + half, origin.y() + half),
                  CellLimits(kInitialSubmapSize, kInitialSubmapSize)),
        device_, &conversion_tables_));
  }
  const proto::SubmapsOptions2D options_;
  int device_ = 0;
  std::vector<std::shared_ptr<Submap2D>> submaps_;
  ValueConversionTables conversion_tables_;
};
} }
#endif  // DROPIN_RESIDENT_SUBMAP_2D_H_
