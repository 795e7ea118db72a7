#include "scan_matchers_2d_mi355x.h"

#include <cstdio>
#include <cstdlib>

namespace cartographer {
namespace mapping {
namespace scan_matching {
namespace {

// glog's CHECK in the reference: abort with the library's message.
void CheckOk(cmx_status status, const char* what) {
  if (status == CMX_OK) return;
  std::fprintf(stderr, "Check failed: %s: %s (%s)\n", what, cmx_status_string(status),
               cmx_last_error());
  std::abort();
}

cmx_grid2d_limits LimitsOf(const Grid2D& grid) {
  const MapLimits& l = grid.limits();
  return cmx_grid2d_limits{l.resolution(), l.max().x(), l.max().y(),
                           l.cell_limits().num_x_cells, l.cell_limits().num_y_cells,
                           grid.GetMinCorrespondenceCost(), grid.GetMaxCorrespondenceCost()};
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

cmx_pose2d PoseOf(const transform::Rigid2d& t) {
  return cmx_pose2d{t.translation().x(), t.translation().y(), t.rotation().angle()};
}

}  // namespace

FastCorrelativeScanMatcher2D::FastCorrelativeScanMatcher2D(
    const Grid2D& grid, const proto::FastCorrelativeScanMatcherOptions2D& options) {
  const cmx_fast2d_options o{options.linear_search_window(), options.angular_search_window(),
                             options.branch_and_bound_depth()};
  const cmx_grid2d_limits limits = LimitsOf(grid);
  CheckOk(cmx_fast2d_create(&o, &limits, grid.correspondence_cost_cells().data(), /*device=*/0,
                            &handle_),
          "cmx_fast2d_create");
}

FastCorrelativeScanMatcher2D::~FastCorrelativeScanMatcher2D() { cmx_fast2d_destroy(handle_); }

bool FastCorrelativeScanMatcher2D::Match(const transform::Rigid2d& initial_pose_estimate,
                                         const sensor::PointCloud& point_cloud,
                                         const float min_score, float* score,
                                         transform::Rigid2d* pose_estimate) const {
  const cmx_pose2d init = PoseOf(initial_pose_estimate);
  const std::vector<float> xyz = Flatten(point_cloud);
  int32_t found = 0;
  float found_score = 0.f;
  cmx_pose2d pose{};
  CheckOk(cmx_fast2d_match(handle_, &init, xyz.data(), static_cast<int32_t>(point_cloud.size()),
                           min_score, &found, score ? &found_score : nullptr,
                           pose_estimate ? &pose : nullptr, nullptr),
          "cmx_fast2d_match");                  // null outputs: the reference CHECKs, so do we
  if (!found) return false;                     // score / pose untouched, like the reference
  *score = found_score;
  *pose_estimate = transform::Rigid2d({pose.x, pose.y}, pose.theta);
  return true;
}

bool FastCorrelativeScanMatcher2D::MatchFullSubmap(const sensor::PointCloud& point_cloud,
                                                   const float min_score, float* score,
                                                   transform::Rigid2d* pose_estimate) const {
  const std::vector<float> xyz = Flatten(point_cloud);
  int32_t found = 0;
  float found_score = 0.f;
  cmx_pose2d pose{};
  CheckOk(cmx_fast2d_match_full_submap(handle_, xyz.data(),
                                       static_cast<int32_t>(point_cloud.size()), min_score,
                                       &found, score ? &found_score : nullptr,
                                       pose_estimate ? &pose : nullptr, nullptr),
          "cmx_fast2d_match_full_submap");
  if (!found) return false;
  *score = found_score;
  *pose_estimate = transform::Rigid2d({pose.x, pose.y}, pose.theta);
  return true;
}

double RealTimeCorrelativeScanMatcher2D::Match(const transform::Rigid2d& initial_pose_estimate,
                                               const sensor::PointCloud& point_cloud,
                                               const Grid2D& grid,
                                               transform::Rigid2d* pose_estimate) const {
  const cmx_rt_options o{options_.linear_search_window(), options_.angular_search_window(),
                         options_.translation_delta_cost_weight(),
                         options_.rotation_delta_cost_weight()};
  const cmx_grid2d_limits limits = LimitsOf(grid);
  const cmx_pose2d init = PoseOf(initial_pose_estimate);
  const std::vector<float> xyz = Flatten(point_cloud);
  double score = 0.;
  cmx_pose2d pose{};
  CheckOk(cmx_rt2d_match(&o, &limits, grid.correspondence_cost_cells().data(), &init, xyz.data(),
                         static_cast<int32_t>(point_cloud.size()), /*device=*/0, &score,
                         pose_estimate ? &pose : nullptr, nullptr),
          "cmx_rt2d_match");
  *pose_estimate = transform::Rigid2d({pose.x, pose.y}, pose.theta);
  return score;
}

}  // namespace scan_matching
}  // namespace mapping
}  // namespace cartographer
