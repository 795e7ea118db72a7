#include "scan_matchers_3d_mi355x.h"

#include <cstdio>
#include <cstdlib>

namespace cartographer {
namespace mapping {
namespace scan_matching {
namespace {

void CheckOk(cmx_status status, const char* what) {
  if (status == CMX_OK) return;
  std::fprintf(stderr, "Check failed: %s: %s (%s)\n", what, cmx_status_string(status),
               cmx_last_error());
  std::abort();
}

std::vector<cmx_voxel> Flatten(const HybridGrid& grid) {     // the HybridGrid::Iterator walk
  std::vector<cmx_voxel> out;
  out.reserve(grid.voxels().size());
  for (const HybridGrid::Voxel& v : grid.voxels())
    out.push_back(cmx_voxel{v.index[0], v.index[1], v.index[2], v.value, 0});
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
  return transform::Rigid3d({p.t[0], p.t[1], p.t[2]}, {p.q[0], p.q[1], p.q[2], p.q[3]});
}

struct FlatNodeData {
  std::vector<float> high, low;
  cmx_node_data3d data;
  explicit FlatNodeData(const TrajectoryNodeData& d)
      : high(Flatten(d.high_resolution_point_cloud)), low(Flatten(d.low_resolution_point_cloud)) {
    data.gravity_alignment[0] = d.gravity_alignment.w();
    data.gravity_alignment[1] = d.gravity_alignment.x();
    data.gravity_alignment[2] = d.gravity_alignment.y();
    data.gravity_alignment[3] = d.gravity_alignment.z();
    data.high_resolution_point_cloud = high.data();
    data.num_high_resolution_points = static_cast<int32_t>(d.high_resolution_point_cloud.size());
    data.low_resolution_point_cloud = low.data();
    data.num_low_resolution_points = static_cast<int32_t>(d.low_resolution_point_cloud.size());
    data.rotational_scan_matcher_histogram = d.rotational_scan_matcher_histogram.data();
    data.histogram_size = static_cast<int32_t>(d.rotational_scan_matcher_histogram.size());
  }
};

std::unique_ptr<FastCorrelativeScanMatcher3D::Result> ResultFrom(int32_t found,
                                                                 const cmx_result3d& r) {
  if (!found) return nullptr;
  return std::unique_ptr<FastCorrelativeScanMatcher3D::Result>(
      new FastCorrelativeScanMatcher3D::Result{r.score, PoseFrom(r.pose_estimate),
                                               r.rotational_score, r.low_resolution_score});
}

}  // namespace

FastCorrelativeScanMatcher3D::FastCorrelativeScanMatcher3D(
    const HybridGrid& hybrid_grid, const HybridGrid* const low_resolution_hybrid_grid,
    const std::vector<float>* rotational_scan_matcher_histogram,
    const proto::FastCorrelativeScanMatcherOptions3D& options) {
  const cmx_fast3d_options o{options.branch_and_bound_depth(), options.full_resolution_depth(),
                             options.min_rotational_score(), options.min_low_resolution_score(),
                             options.linear_xy_search_window(), options.linear_z_search_window(),
                             options.angular_search_window()};
  const std::vector<cmx_voxel> voxels = Flatten(hybrid_grid);
  const std::vector<cmx_voxel> low = Flatten(*low_resolution_hybrid_grid);
  CheckOk(cmx_fast3d_create(&o, hybrid_grid.resolution(), hybrid_grid.grid_size(), voxels.data(),
                            static_cast<int64_t>(voxels.size()),
                            low_resolution_hybrid_grid->resolution(), low.data(),
                            static_cast<int64_t>(low.size()),
                            rotational_scan_matcher_histogram->data(),
                            static_cast<int32_t>(rotational_scan_matcher_histogram->size()),
                            /*device=*/0, &handle_),
          "cmx_fast3d_create");
}

FastCorrelativeScanMatcher3D::~FastCorrelativeScanMatcher3D() { cmx_fast3d_destroy(handle_); }

std::unique_ptr<FastCorrelativeScanMatcher3D::Result> FastCorrelativeScanMatcher3D::Match(
    const transform::Rigid3d& global_node_pose, const transform::Rigid3d& global_submap_pose,
    const TrajectoryNodeData& constant_data, const float min_score) const {
  const FlatNodeData flat(constant_data);
  const cmx_pose3d node = PoseOf(global_node_pose), submap = PoseOf(global_submap_pose);
  int32_t found = 0;
  cmx_result3d result{};
  CheckOk(cmx_fast3d_match(handle_, &node, &submap, &flat.data, min_score, &found, &result, nullptr),
          "cmx_fast3d_match");
  return ResultFrom(found, result);
}

std::unique_ptr<FastCorrelativeScanMatcher3D::Result>
FastCorrelativeScanMatcher3D::MatchFullSubmap(const transform::Quaterniond& global_node_rotation,
                                              const transform::Quaterniond& global_submap_rotation,
                                              const TrajectoryNodeData& constant_data,
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

float RealTimeCorrelativeScanMatcher3D::Match(const transform::Rigid3d& initial_pose_estimate,
                                              const sensor::PointCloud& point_cloud,
                                              const HybridGrid& hybrid_grid,
                                              transform::Rigid3d* pose_estimate) const {
  const cmx_rt_options o{options_.linear_search_window(), options_.angular_search_window(),
                         options_.translation_delta_cost_weight(),
                         options_.rotation_delta_cost_weight()};
  const std::vector<cmx_voxel> voxels = Flatten(hybrid_grid);
  const std::vector<float> xyz = Flatten(point_cloud);
  const cmx_pose3d init = PoseOf(initial_pose_estimate);
  float score = 0.f;
  cmx_pose3d pose{};
  CheckOk(cmx_rt3d_match(&o, hybrid_grid.resolution(), voxels.data(),
                         static_cast<int64_t>(voxels.size()), &init, xyz.data(),
                         static_cast<int32_t>(point_cloud.size()), /*device=*/0, &score,
                         pose_estimate ? &pose : nullptr, nullptr),
          "cmx_rt3d_match");
  *pose_estimate = PoseFrom(pose);
  return score;
}

}  // namespace scan_matching
}  // namespace mapping
}  // namespace cartographer
