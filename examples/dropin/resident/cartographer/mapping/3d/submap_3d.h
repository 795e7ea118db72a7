// Submap3D / ActiveSubmaps3D with both hybrid grids RESIDENT IN HBM: the third build of the 3D
// local-trajectory-builder test.  Same bookkeeping as shims_local/.../submap_3d.h (which restates
// mapping/3d/submap_3d.cc:162-177,276-327), but each grid is a cmx_grid3d: InsertData is two
// cmx_grid3d_insert calls (the reference's RangeDataInserter3D::Insert on the device, bit for
// bit).  The HybridGrid objects the accessors hand out stay empty; their addresses are registered
// with the device grids (device_grids.h), so that the adapters take cmx_rt3d_match_grid and
// cmx_ceres3d_match_grids: per scan only the point clouds cross PCIe.
#ifndef DROPIN_RESIDENT_SUBMAP_3D_H_
#define DROPIN_RESIDENT_SUBMAP_3D_H_
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>
#include "Eigen/Core"
#include "Eigen/Geometry"
#include "absl/types/optional.h"
#include "cartographer/mapping/3d/hybrid_grid.h"
#include "cartographer/mapping/internal/3d/scan_matching/rotational_scan_matcher.h"
#include "cartographer/mapping/proto/submaps_options_3d.pb.h"
#include "cartographer/mapping/trajectory_node.h"
#include "cartographer/sensor/range_data.h"
#include "cartographer/transform/rigid_transform.h"
#include "cartographer/transform/transform.h"
#include "device_grids.h"
namespace cartographer { namespace mapping {

inline void DropinCheckOk(cmx_status status, const char* what) {
  if (status == CMX_OK) return;
  std::fprintf(stderr, "Check failed: %s: %s (%s)\n", what, cmx_status_string(status),
               cmx_last_error());
  std::abort();
}

class Submap3D {
 public:
  Submap3D(float high_resolution, float low_resolution, const transform::Rigid3d& local_submap_pose,
           const Eigen::VectorXf& rotational_scan_matcher_histogram, const int device)
      : local_pose_(local_submap_pose),
        high_resolution_hybrid_grid_(std::make_unique<HybridGrid>(high_resolution)),
        low_resolution_hybrid_grid_(std::make_unique<HybridGrid>(low_resolution)),
        high_resolution_intensity_hybrid_grid_(std::make_unique<IntensityHybridGrid>(high_resolution)),
        rotational_scan_matcher_histogram_(rotational_scan_matcher_histogram) {
    DropinCheckOk(cmx_grid3d_create(high_resolution, device, &high_), "cmx_grid3d_create");
    DropinCheckOk(cmx_grid3d_create(low_resolution, device, &low_), "cmx_grid3d_create");
    dropin::RegisterDeviceGrid(high_resolution_hybrid_grid_.get(), high_);
    dropin::RegisterDeviceGrid(low_resolution_hybrid_grid_.get(), low_);
  }
  ~Submap3D() {
    dropin::UnregisterDeviceGrid(high_resolution_hybrid_grid_.get());
    dropin::UnregisterDeviceGrid(low_resolution_hybrid_grid_.get());
    cmx_grid3d_destroy(high_);
    cmx_grid3d_destroy(low_);
  }
  Submap3D(const Submap3D&) = delete;
  Submap3D& operator=(const Submap3D&) = delete;

  transform::Rigid3d local_pose() const { return local_pose_; }
  int num_range_data() const { return num_range_data_; }
  bool insertion_finished() const { return insertion_finished_; }
  const HybridGrid& high_resolution_hybrid_grid() const { return *high_resolution_hybrid_grid_; }
  const HybridGrid& low_resolution_hybrid_grid() const { return *low_resolution_hybrid_grid_; }
  const IntensityHybridGrid& high_resolution_intensity_hybrid_grid() const {
    return *high_resolution_intensity_hybrid_grid_;
  }
  const Eigen::VectorXf& rotational_scan_matcher_histogram() const {
    return rotational_scan_matcher_histogram_;
  }
  void InsertData(const sensor::RangeData& range_data_in_local,
                  const proto::RangeDataInserterOptions3D& inserter, float high_resolution_max_range,
                  const Eigen::Quaterniond& local_from_gravity_aligned,
                  const Eigen::VectorXf& scan_histogram_in_gravity) {
    CHECK(!insertion_finished_);
    const sensor::RangeData in_submap =
        sensor::TransformRangeData(range_data_in_local, local_pose_.inverse().cast<float>());
    const float origin[3] = {in_submap.origin.x(), in_submap.origin.y(), in_submap.origin.z()};
    std::vector<float> all, near;
    for (const sensor::RangefinderPoint& hit : in_submap.returns) {
      const float xyz[3] = {hit.position.x(), hit.position.y(), hit.position.z()};
      all.insert(all.end(), xyz, xyz + 3);
      if ((hit.position - in_submap.origin).norm() <= high_resolution_max_range)
        near.insert(near.end(), xyz, xyz + 3);
    }
    DropinCheckOk(cmx_grid3d_insert(high_, origin, near.data(), static_cast<int32_t>(near.size() / 3),
                                    static_cast<float>(inserter.hit_probability()),
                                    static_cast<float>(inserter.miss_probability()),
                                    inserter.num_free_space_voxels()),
                  "cmx_grid3d_insert");
    DropinCheckOk(cmx_grid3d_insert(low_, origin, all.data(), static_cast<int32_t>(all.size() / 3),
                                    static_cast<float>(inserter.hit_probability()),
                                    static_cast<float>(inserter.miss_probability()),
                                    inserter.num_free_space_voxels()),
                  "cmx_grid3d_insert");
    ++num_range_data_;
    const float yaw_in_submap_from_gravity =
        transform::GetYaw(local_pose_.inverse().rotation() * local_from_gravity_aligned);
    rotational_scan_matcher_histogram_ += scan_matching::RotationalScanMatcher::RotateHistogram(
        scan_histogram_in_gravity, yaw_in_submap_from_gravity);
  }
  void Finish() {
    CHECK(!insertion_finished_);
    insertion_finished_ = true;
  }
  // For the test main's digest: the voxels of both device grids into the host objects.
  void SyncToHost() const {
    Download(high_, high_resolution_hybrid_grid_.get());
    Download(low_, low_resolution_hybrid_grid_.get());
  }
 private:
  static void Download(const cmx_grid3d* from, HybridGrid* to) {
    int64_t count = 0;
    DropinCheckOk(cmx_grid3d_info(from, nullptr, nullptr, &count), "cmx_grid3d_info");
    std::vector<cmx_voxel> voxels(static_cast<size_t>(count) + 1);
    DropinCheckOk(cmx_grid3d_download(from, voxels.data(), count + 1, &count), "cmx_grid3d_download");
    for (int64_t i = 0; i != count; ++i)
      *to->mutable_value(Eigen::Array3i(voxels[i].x, voxels[i].y, voxels[i].z)) = voxels[i].value;
  }
  const transform::Rigid3d local_pose_;
  std::unique_ptr<HybridGrid> high_resolution_hybrid_grid_, low_resolution_hybrid_grid_;
  std::unique_ptr<IntensityHybridGrid> high_resolution_intensity_hybrid_grid_;
  Eigen::VectorXf rotational_scan_matcher_histogram_;
  cmx_grid3d* high_ = nullptr;
  cmx_grid3d* low_ = nullptr;
  int num_range_data_ = 0;
  bool insertion_finished_ = false;
};

inline void DropinSyncSubmapToHost(const Submap3D& submap) { submap.SyncToHost(); }

class ActiveSubmaps3D {
 public:
  explicit ActiveSubmaps3D(const proto::SubmapsOptions3D& options) : options_(options) {
    const char* e = std::getenv("CMX_DEVICE");
    device_ = e ? std::atoi(e) : 0;
  }
  ActiveSubmaps3D(const ActiveSubmaps3D&) = delete;
  ActiveSubmaps3D& operator=(const ActiveSubmaps3D&) = delete;
  std::vector<std::shared_ptr<const Submap3D>> submaps() const {
    return std::vector<std::shared_ptr<const Submap3D>>(submaps_.begin(), submaps_.end());
  }
  std::vector<std::shared_ptr<const Submap3D>> InsertData(
      const sensor::RangeData& range_data, const Eigen::Quaterniond& local_from_gravity_aligned,
      const Eigen::VectorXf& rotational_scan_matcher_histogram_in_gravity) {
    if (submaps_.empty() || submaps_.back()->num_range_data() == options_.num_range_data()) {
      if (submaps_.size() >= 2) {
        CHECK(submaps_.front()->insertion_finished());
        submaps_.erase(submaps_.begin());
      }
      submaps_.push_back(std::make_shared<Submap3D>(
          options_.high_resolution(), options_.low_resolution(),
          transform::Rigid3d(range_data.origin.cast<double>(), local_from_gravity_aligned),
          Eigen::VectorXf::Zero(rotational_scan_matcher_histogram_in_gravity.size()), device_));
    }
    for (auto& submap : submaps_)
      submap->InsertData(range_data, options_.range_data_inserter_options(),
                         options_.high_resolution_max_range(), local_from_gravity_aligned,
                         rotational_scan_matcher_histogram_in_gravity);
    if (submaps_.front()->num_range_data() == 2 * options_.num_range_data())
      submaps_.front()->Finish();
    return submaps();
  }
 private:
  const proto::SubmapsOptions3D options_;
  int device_ = 0;
  std::vector<std::shared_ptr<Submap3D>> submaps_;
};
} }
#endif  // DROPIN_RESIDENT_SUBMAP_3D_H_
