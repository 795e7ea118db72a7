// Bodies of FastCorrelativeScanMatcher2D and CeresScanMatcher2D over libcartographer_mi355x:
// the ONLY source a maintainer swaps for the reference's fast_correlative_scan_matcher_2d.cc /
// ceres_scan_matcher_2d.cc (+ occupied_space_cost_function_2d.cc).  Everything that calls them
// -- here the reference's own constraint_builder_2d.cc -- compiles unmodified.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cartographer/mapping/internal/2d/scan_matching/ceres_scan_matcher_2d.h"
#include "cartographer/mapping/internal/2d/scan_matching/fast_correlative_scan_matcher_2d.h"
#include "device_grids.h"

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

// Grid2D keeps its raw uint16 cells behind a protected accessor (mapping/2d/grid_2d.h:96); a
// derived class may name it, which is all a pointer-to-member needs.
struct CellAccess : Grid2D {
  using Grid2D::correspondence_cost_cells;
};
// (host cells: a grid that lives in HBM -- device_grids.h -- hands out an all-zero host image,
// which must never be uploaded in its place: every caller below looks for the resident grid
// first, and a path that has no resident form stops here instead of matching against nothing)
const uint16_t* CellsOf(const Grid2D& grid) {
  if (dynamic_cast<const dropin::DeviceGrid2DView*>(&grid) != nullptr) {
    std::fprintf(stderr, "Check failed: a grid resident in HBM reached a host-upload path of the "
                         "2D scan matcher adapters (scan_matchers_2d_mi355x.cc)\n");
    std::abort();
  }
  return (grid.*(&CellAccess::correspondence_cost_cells))().data();
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

// Which GPU a process-wide matcher lands on: CMX_DEVICE (default 0).  A multi-GPU host
// creates its matchers through cmx_comm_device_of instead.
int Device() {
  const char* e = std::getenv("CMX_DEVICE");
  return e ? std::atoi(e) : 0;
}

}  // namespace

FastCorrelativeScanMatcher2D::FastCorrelativeScanMatcher2D(
    const Grid2D& grid, const proto::FastCorrelativeScanMatcherOptions2D& options) {
  const cmx_fast2d_options o{options.linear_search_window(), options.angular_search_window(),
                             options.branch_and_bound_depth()};
  if (const auto* resident = dynamic_cast<const dropin::DeviceGrid2DView*>(&grid)) {
    // the submap's grid is in HBM already: the stack is built from it there
    CheckOk(cmx_fast2d_create_from_grid(&o, resident->device_grid(), &handle_),
            "cmx_fast2d_create_from_grid");
    return;
  }
  const cmx_grid2d_limits limits = LimitsOf(grid);
  CheckOk(cmx_fast2d_create(&o, &limits, CellsOf(grid), Device(),
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
          "cmx_fast2d_match");
  if (!found) return false;
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

void CeresScanMatcher2D::Match(const Eigen::Vector2d& target_translation,
                               const transform::Rigid2d& initial_pose_estimate,
                               const sensor::PointCloud& point_cloud, const Grid2D& grid,
                               transform::Rigid2d* const pose_estimate,
                               ceres::Solver::Summary* const summary) const {
  const cmx_ceres2d_options o{options_.occupied_space_weight(), options_.translation_weight(),
                              options_.rotation_weight(),
                              options_.ceres_solver_options().use_nonmonotonic_steps() ? 1 : 0,
                              options_.ceres_solver_options().max_num_iterations()};
  const cmx_grid2d_limits limits = LimitsOf(grid);
  const double target[2] = {target_translation.x(), target_translation.y()};
  const cmx_pose2d init = PoseOf(initial_pose_estimate);
  const std::vector<float> xyz = Flatten(point_cloud);
  cmx_pose2d pose{};
  cmx_ceres_summary s{};
  if (const auto* resident = dynamic_cast<const dropin::DeviceGrid2DView*>(&grid)) {
    CheckOk(cmx_ceres2d_match_grid(&o, resident->device_grid(), target, &init, xyz.data(),
                                   static_cast<int32_t>(point_cloud.size()), &pose, &s),
            "cmx_ceres2d_match_grid");
  } else {
    CheckOk(cmx_ceres2d_match(&o, &limits, CellsOf(grid), target, &init,
                              xyz.data(), static_cast<int32_t>(point_cloud.size()), Device(), &pose,
                              &s),
            "cmx_ceres2d_match");
  }
  *pose_estimate = transform::Rigid2d({pose.x, pose.y}, pose.theta);
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
