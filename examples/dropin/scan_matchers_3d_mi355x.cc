// Bodies of FastCorrelativeScanMatcher3D and CeresScanMatcher3D over libcartographer_mi355x:
// the ONLY source a maintainer swaps for the reference's fast_correlative_scan_matcher_3d.cc
// (+ precomputation_grid_3d.cc, rotational_scan_matcher.cc, low_resolution_matcher.cc) and
// ceres_scan_matcher_3d.cc (+ the occupied-space / delta cost functors).  Everything that calls
// them -- here the reference's own constraint_builder_3d.cc -- compiles unmodified.  The grids
// cross the boundary as the voxel list HybridGrid::Iterator yields (hybrid_grid.h:304-372).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cartographer/mapping/internal/3d/scan_matching/ceres_scan_matcher_3d.h"
#include "cartographer/mapping/internal/3d/scan_matching/fast_correlative_scan_matcher_3d.h"
#include "device_grids.h"

namespace cartographer {
namespace mapping {
// The reference never iterates an IntensityHybridGrid, so AverageIntensityData has no operator==
// for IsDefaultValue (hybrid_grid.h:55-58); found by ADL when the iterator is instantiated below.
inline bool operator==(const AverageIntensityData& a, const AverageIntensityData& b) {
  return a.sum == b.sum && a.count == b.count;
}
namespace scan_matching {
namespace {

void CheckOk(cmx_status status, const char* what) {
  if (status == CMX_OK) return;
  std::fprintf(stderr, "Check failed: %s: %s (%s)\n", what, cmx_status_string(status),
               cmx_last_error());
  std::abort();
}

// (host voxels: the HybridGrid a resident submap hands out is EMPTY -- its voxels are in HBM,
// device_grids.h -- and must never be uploaded in its place: a path that has no resident form
// stops here instead of silently matching against nothing)
std::vector<cmx_voxel> Flatten(const HybridGrid& grid) {
  if (dropin::DeviceGridOf(&grid) != nullptr) {
    std::fprintf(stderr, "Check failed: a hybrid grid resident in HBM reached a host-upload path "
                         "of the 3D scan matcher adapters (scan_matchers_3d_mi355x.cc): with an "
                         "intensity grid in the call, keep the submap's grids on the host\n");
    std::abort();
  }
  std::vector<cmx_voxel> out;
  for (auto it = HybridGrid::Iterator(grid); !it.Done(); it.Next()) {
    const Eigen::Array3i index = it.GetCellIndex();
    out.push_back(cmx_voxel{index.x(), index.y(), index.z(), it.GetValue(), 0});
  }
  return out;
}

std::vector<float> Flatten(const sensor::PointCloud& cloud) {
  std::vector<float> xyz;
  xyz.reserve(3 * cloud.size());
  for (const sensor::RangefinderPoint& p : cloud) {
    xyz.push_back(p.position.x());
    xyz.push_back(p.position.y());
    xyz.push_back(p.position.z());
  }
  return xyz;
}

cmx_pose3d PoseOf(const transform::Rigid3d& t) {
  return cmx_pose3d{{t.translation().x(), t.translation().y(), t.translation().z()},
                    {t.rotation().w(), t.rotation().x(), t.rotation().y(), t.rotation().z()}};
}

transform::Rigid3d PoseFrom(const cmx_pose3d& p) {
  return transform::Rigid3d(Eigen::Vector3d(p.t[0], p.t[1], p.t[2]),
                            Eigen::Quaterniond(p.q[0], p.q[1], p.q[2], p.q[3]));
}

struct FlatNodeData {
  std::vector<float> high, low, histogram;
  cmx_node_data3d data;
  explicit FlatNodeData(const TrajectoryNode::Data& d)
      : high(Flatten(d.high_resolution_point_cloud)), low(Flatten(d.low_resolution_point_cloud)) {
    for (int i = 0; i != d.rotational_scan_matcher_histogram.size(); ++i)
      histogram.push_back(d.rotational_scan_matcher_histogram[i]);
    data.gravity_alignment[0] = d.gravity_alignment.w();
    data.gravity_alignment[1] = d.gravity_alignment.x();
    data.gravity_alignment[2] = d.gravity_alignment.y();
    data.gravity_alignment[3] = d.gravity_alignment.z();
    data.high_resolution_point_cloud = high.data();
    data.num_high_resolution_points = static_cast<int32_t>(d.high_resolution_point_cloud.size());
    data.low_resolution_point_cloud = low.data();
    data.num_low_resolution_points = static_cast<int32_t>(d.low_resolution_point_cloud.size());
    data.rotational_scan_matcher_histogram = histogram.data();
    data.histogram_size = static_cast<int32_t>(histogram.size());
  }
};

std::unique_ptr<FastCorrelativeScanMatcher3D::Result> ResultFrom(int32_t found,
                                                                 const cmx_result3d& r) {
  if (!found) return nullptr;
  return std::unique_ptr<FastCorrelativeScanMatcher3D::Result>(
      new FastCorrelativeScanMatcher3D::Result{r.score, PoseFrom(r.pose_estimate),
                                               r.rotational_score, r.low_resolution_score});
}

int Device() {
  const char* e = std::getenv("CMX_DEVICE");
  return e ? std::atoi(e) : 0;
}

}  // namespace

FastCorrelativeScanMatcher3D::FastCorrelativeScanMatcher3D(
    const HybridGrid& hybrid_grid, const HybridGrid* const low_resolution_hybrid_grid,
    const Eigen::VectorXf* rotational_scan_matcher_histogram,
    const proto::FastCorrelativeScanMatcherOptions3D& options) {
  const cmx_fast3d_options o{options.branch_and_bound_depth(), options.full_resolution_depth(),
                             options.min_rotational_score(), options.min_low_resolution_score(),
                             options.linear_xy_search_window(), options.linear_z_search_window(),
                             options.angular_search_window()};
  const std::vector<cmx_voxel> voxels = Flatten(hybrid_grid);
  const std::vector<cmx_voxel> low = Flatten(*low_resolution_hybrid_grid);
  std::vector<float> histogram;
  for (int i = 0; i != rotational_scan_matcher_histogram->size(); ++i)
    histogram.push_back((*rotational_scan_matcher_histogram)[i]);
  CheckOk(cmx_fast3d_create(&o, hybrid_grid.resolution(), hybrid_grid.grid_size(), voxels.data(),
                            static_cast<int64_t>(voxels.size()),
                            low_resolution_hybrid_grid->resolution(), low.data(),
                            static_cast<int64_t>(low.size()), histogram.data(),
                            static_cast<int32_t>(histogram.size()), Device(), &handle_),
          "cmx_fast3d_create");
}

FastCorrelativeScanMatcher3D::~FastCorrelativeScanMatcher3D() { cmx_fast3d_destroy(handle_); }

std::unique_ptr<FastCorrelativeScanMatcher3D::Result> FastCorrelativeScanMatcher3D::Match(
    const transform::Rigid3d& global_node_pose, const transform::Rigid3d& global_submap_pose,
    const TrajectoryNode::Data& constant_data, const float min_score) const {
  const FlatNodeData flat(constant_data);
  const cmx_pose3d node = PoseOf(global_node_pose), submap = PoseOf(global_submap_pose);
  int32_t found = 0;
  cmx_result3d result{};
  CheckOk(cmx_fast3d_match(handle_, &node, &submap, &flat.data, min_score, &found, &result, nullptr),
          "cmx_fast3d_match");
  return ResultFrom(found, result);
}

std::unique_ptr<FastCorrelativeScanMatcher3D::Result>
FastCorrelativeScanMatcher3D::MatchFullSubmap(const Eigen::Quaterniond& global_node_rotation,
                                              const Eigen::Quaterniond& global_submap_rotation,
                                              const TrajectoryNode::Data& constant_data,
                                              const float min_score) const {
  const FlatNodeData flat(constant_data);
  const double node_q[4] = {global_node_rotation.w(), global_node_rotation.x(),
                            global_node_rotation.y(), global_node_rotation.z()};
  const double submap_q[4] = {global_submap_rotation.w(), global_submap_rotation.x(),
                              global_submap_rotation.y(), global_submap_rotation.z()};
  int32_t found = 0;
  cmx_result3d result{};
  CheckOk(cmx_fast3d_match_full_submap(handle_, node_q, submap_q, &flat.data, min_score, &found,
                                       &result, nullptr),
          "cmx_fast3d_match_full_submap");
  return ResultFrom(found, result);
}

void CeresScanMatcher3D::Match(
    const Eigen::Vector3d& target_translation, const transform::Rigid3d& initial_pose_estimate,
    const std::vector<PointCloudAndHybridGridsPointers>& point_clouds_and_hybrid_grids,
    transform::Rigid3d* const pose_estimate, ceres::Solver::Summary* const summary) const {
  cmx_ceres3d_options o{};
  o.num_pairs = static_cast<int32_t>(point_clouds_and_hybrid_grids.size());
  if (o.num_pairs != options_.occupied_space_weight_size() || o.num_pairs > 3) {   // CHECK_EQ :108
    std::fprintf(stderr, "Check failed: %d (cloud, grid) pairs, %d occupied_space_weights\n",
                 o.num_pairs, options_.occupied_space_weight_size());
    std::abort();
  }
  for (int i = 0; i != o.num_pairs; ++i) o.occupied_space_weight[i] = options_.occupied_space_weight(i);
  o.translation_weight = options_.translation_weight();
  o.rotation_weight = options_.rotation_weight();
  o.only_optimize_yaw = options_.only_optimize_yaw() ? 1 : 0;
  o.use_nonmonotonic_steps = options_.ceres_solver_options().use_nonmonotonic_steps() ? 1 : 0;
  o.max_num_iterations = options_.ceres_solver_options().max_num_iterations();
  // Every grid of the call in HBM already (device_grids.h) and no intensity term: only the clouds
  // cross PCIe.
  {
    std::vector<const cmx_grid3d*> resident;
    for (const PointCloudAndHybridGridsPointers& p : point_clouds_and_hybrid_grids) {
      const cmx_grid3d* grid =
          p.intensity_hybrid_grid == nullptr ? dropin::DeviceGridOf(p.hybrid_grid) : nullptr;
      if (grid == nullptr) break;
      resident.push_back(grid);
    }
    if (static_cast<int>(resident.size()) == o.num_pairs) {
      std::vector<std::vector<float>> flat;
      std::vector<const float*> xyz;
      std::vector<int32_t> counts;
      for (const PointCloudAndHybridGridsPointers& p : point_clouds_and_hybrid_grids) {
        flat.push_back(Flatten(*p.point_cloud));
        counts.push_back(static_cast<int32_t>(p.point_cloud->size()));
      }
      for (const std::vector<float>& f : flat) xyz.push_back(f.data());
      const double target[3] = {target_translation.x(), target_translation.y(),
                                target_translation.z()};
      const cmx_pose3d init = PoseOf(initial_pose_estimate);
      cmx_pose3d pose{};
      cmx_ceres_summary s{};
      CheckOk(cmx_ceres3d_match_grids(&o, target, &init, resident.data(), xyz.data(), counts.data(),
                                      &pose, &s),
              "cmx_ceres3d_match_grids");
      *pose_estimate = PoseFrom(pose);
      if (summary) {
        summary->initial_cost = s.initial_cost;
        summary->final_cost = s.final_cost;
        summary->num_successful_steps = s.num_successful_steps;
        summary->num_unsuccessful_steps = s.num_unsuccessful_steps;
        summary->termination_type = s.termination;
      }
      return;
    }
  }
  std::vector<std::vector<cmx_voxel>> voxels;
  std::vector<std::vector<float>> clouds;
  std::vector<cmx_ceres3d_pair> pairs;
  std::vector<std::vector<cmx_intensity_voxel>> intensity_voxels;
  voxels.reserve(o.num_pairs); clouds.reserve(o.num_pairs); intensity_voxels.reserve(o.num_pairs);
  for (int i = 0; i != o.num_pairs; ++i) {
    const PointCloudAndHybridGridsPointers& p = point_clouds_and_hybrid_grids[i];
    voxels.push_back(Flatten(*p.hybrid_grid));
    clouds.push_back(Flatten(*p.point_cloud));
    cmx_ceres3d_pair pair{};
    pair.point_cloud_xyz = clouds.back().data();
    pair.num_points = static_cast<int32_t>(p.point_cloud->size());
    pair.resolution = p.hybrid_grid->resolution();
    pair.voxels = voxels.back().data();
    pair.num_voxels = static_cast<int64_t>(voxels.back().size());
    if (p.intensity_hybrid_grid != nullptr) {                  // ceres_scan_matcher_3d.cc:118-137
      const auto& io = options_.intensity_cost_function_options(i);
      if (!(io.huber_scale() > 0.) || !(io.weight() > 0.) ||
          p.point_cloud->intensities().size() != p.point_cloud->size()) {     // CHECK_GT :124-125
        std::fprintf(stderr, "Check failed: intensity_cost_function_options_%d / intensities\n", i);
        std::abort();
      }
      intensity_voxels.emplace_back();                         // the AverageIntensityData cells
      for (auto it = IntensityHybridGrid::Iterator(*p.intensity_hybrid_grid); !it.Done(); it.Next())
        intensity_voxels.back().push_back(cmx_intensity_voxel{it.GetCellIndex().x(), it.GetCellIndex().y(),
                                                              it.GetCellIndex().z(), it.GetValue().count,
                                                              it.GetValue().sum});
      pair.intensities = p.point_cloud->intensities().data();
      pair.intensity_voxels = intensity_voxels.back().data();
      pair.num_intensity_voxels = static_cast<int64_t>(intensity_voxels.back().size());
      pair.intensity_weight = io.weight();
      pair.intensity_huber_scale = io.huber_scale();
      pair.intensity_threshold = static_cast<float>(io.intensity_threshold());
    }
    pairs.push_back(pair);
  }
  const double target[3] = {target_translation.x(), target_translation.y(),
                            target_translation.z()};
  const cmx_pose3d init = PoseOf(initial_pose_estimate);
  cmx_pose3d pose{};
  cmx_ceres_summary s{};
  CheckOk(cmx_ceres3d_match(&o, target, &init, pairs.data(), Device(), &pose, &s),
          "cmx_ceres3d_match");
  *pose_estimate = PoseFrom(pose);
  if (summary) {
    summary->initial_cost = s.initial_cost;
    summary->final_cost = s.final_cost;
    summary->num_successful_steps = s.num_successful_steps;
    summary->num_unsuccessful_steps = s.num_unsuccessful_steps;
    summary->termination_type = s.termination;
  }
}

}  // namespace scan_matching
}  // namespace mapping
}  // namespace cartographer
