// Stand-in for mapping/3d/submap_3d.h (+ mapping/submaps.h's Submap) in the
// local-trajectory-builder build: Submap3D and ActiveSubmaps3D with the members
// LocalTrajectoryBuilder3D and the scan matchers use.  The bookkeeping restates
// mapping/3d/submap_3d.cc:162-177 (the high-resolution grid takes the returns within
// high_resolution_max_range only), :276-327 (both grids, the histogram rotated into the submap,
// two active submaps, a new one every num_range_data scans, the older finished at twice that)
// over the reference's OWN HybridGrid, RangeDataInserter3D and
// RotationalScanMatcher::RotateHistogram, which this build compiles where they lie.  Left out:
// the proto round trip and the submap textures.
#ifndef DROPIN_SHIMS_LOCAL_SUBMAP_3D_H_
#define DROPIN_SHIMS_LOCAL_SUBMAP_3D_H_
#include <memory>
#include <vector>
#include "Eigen/Core"
#include "Eigen/Geometry"
#include "absl/types/optional.h"   // reaches local_trajectory_builder_3d.h through Abseil upstream
#include "cartographer/mapping/3d/hybrid_grid.h"
#include "cartographer/mapping/3d/range_data_inserter_3d.h"
#include "cartographer/mapping/internal/3d/scan_matching/rotational_scan_matcher.h"
#include "cartographer/mapping/proto/submaps_options_3d.pb.h"
#include "cartographer/mapping/trajectory_node.h"
#include "cartographer/sensor/range_data.h"
#include "cartographer/transform/rigid_transform.h"
#include "cartographer/transform/transform.h"
namespace cartographer { namespace mapping {
class Submap3D {
 public:
  Submap3D(float high_resolution, float low_resolution, const transform::Rigid3d& local_submap_pose,
           const Eigen::VectorXf& rotational_scan_matcher_histogram)
      : local_pose_(local_submap_pose),
        high_resolution_hybrid_grid_(std::make_unique<HybridGrid>(high_resolution)),
        low_resolution_hybrid_grid_(std::make_unique<HybridGrid>(low_resolution)),
        high_resolution_intensity_hybrid_grid_(std::make_unique<IntensityHybridGrid>(high_resolution)),
        rotational_scan_matcher_histogram_(rotational_scan_matcher_histogram) {}
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
                  const RangeDataInserter3D& range_data_inserter, float high_resolution_max_range,
                  const Eigen::Quaterniond& local_from_gravity_aligned,
                  const Eigen::VectorXf& scan_histogram_in_gravity) {
    CHECK(!insertion_finished_);
    const sensor::RangeData in_submap =
        sensor::TransformRangeData(range_data_in_local, local_pose_.inverse().cast<float>());
    sensor::RangeData near{in_submap.origin, {}, {}};
    for (const sensor::RangefinderPoint& hit : in_submap.returns)
      if ((hit.position - in_submap.origin).norm() <= high_resolution_max_range)
        near.returns.push_back(hit);
    range_data_inserter.Insert(near, high_resolution_hybrid_grid_.get(),
                               high_resolution_intensity_hybrid_grid_.get());
    range_data_inserter.Insert(in_submap, low_resolution_hybrid_grid_.get(), nullptr);
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
 private:
  const transform::Rigid3d local_pose_;
  std::unique_ptr<HybridGrid> high_resolution_hybrid_grid_, low_resolution_hybrid_grid_;
  std::unique_ptr<IntensityHybridGrid> high_resolution_intensity_hybrid_grid_;
  Eigen::VectorXf rotational_scan_matcher_histogram_;
  int num_range_data_ = 0;
  bool insertion_finished_ = false;
};

inline void DropinSyncSubmapToHost(const Submap3D&) {}   // (resident/: downloads the device grids)

class ActiveSubmaps3D {
 public:
  explicit ActiveSubmaps3D(const proto::SubmapsOptions3D& options)
      : options_(options), range_data_inserter_(options.range_data_inserter_options()) {}
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
          Eigen::VectorXf::Zero(rotational_scan_matcher_histogram_in_gravity.size())));
    }
    for (auto& submap : submaps_)
      submap->InsertData(range_data, range_data_inserter_, options_.high_resolution_max_range(),
                         local_from_gravity_aligned, rotational_scan_matcher_histogram_in_gravity);
    if (submaps_.front()->num_range_data() == 2 * options_.num_range_data())
      submaps_.front()->Finish();
    return submaps();
  }
 private:
  const proto::SubmapsOptions3D options_;
  std::vector<std::shared_ptr<Submap3D>> submaps_;
  RangeDataInserter3D range_data_inserter_;
};
} }
#endif  // DROPIN_SHIMS_LOCAL_SUBMAP_3D_H_
