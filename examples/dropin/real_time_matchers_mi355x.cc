// Bodies of RealTimeCorrelativeScanMatcher2D and RealTimeCorrelativeScanMatcher3D over
// libcartographer_mi355x, compiled against the reference's REAL headers
// (mapping/internal/2d/scan_matching/real_time_correlative_scan_matcher_2d.h:48-83,
// mapping/internal/3d/scan_matching/real_time_correlative_scan_matcher_3d.h:36-66): both classes
// keep nothing but their options, so no member changes and no stand-in header.  This file is the
// ONLY source a maintainer swaps for real_time_correlative_scan_matcher_2d.cc / _3d.cc; their
// callers (2d/local_trajectory_builder_2d.cc:78-80, 3d/local_trajectory_builder_3d.cc:96-108)
// compile unmodified.  The private helpers of the reference's implementation
// (GenerateExhaustiveSearchCandidates / GenerateExhaustiveSearchTransforms / ScoreCandidate) run
// inside the library and are not defined here; nothing outside those .cc files names them.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cartographer/mapping/2d/probability_grid.h"
#include "cartographer/mapping/internal/2d/scan_matching/real_time_correlative_scan_matcher_2d.h"
#include "cartographer/mapping/internal/2d/tsdf_2d.h"
#include "cartographer/mapping/internal/3d/scan_matching/real_time_correlative_scan_matcher_3d.h"
#include "cartographer_mi355x.h"
#include "device_grids.h"

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

int Device() {
  const char* e = std::getenv("CMX_DEVICE");
  return e ? std::atoi(e) : 0;
}

cmx_rt_options OptionsOf(const proto::RealTimeCorrelativeScanMatcherOptions& o) {
  return cmx_rt_options{o.linear_search_window(), o.angular_search_window(),
                        o.translation_delta_cost_weight(), o.rotation_delta_cost_weight()};
}

// Grid2D keeps its raw uint16 cells behind a protected accessor (mapping/2d/grid_2d.h:96).
struct CellAccess : Grid2D {
  using Grid2D::correspondence_cost_cells;
};
// (host cells: a grid resident in HBM -- device_grids.h -- hands out an all-zero host image;
// Match() looks for the resident grid first, a path without a resident form stops here)
const std::vector<uint16>& CellsOf(const Grid2D& grid) {
  if (dynamic_cast<const dropin::DeviceGrid2DView*>(&grid) != nullptr) {
    std::fprintf(stderr, "Check failed: a grid resident in HBM reached a host-upload path of the "
                         "real-time matcher adapters (real_time_matchers_mi355x.cc)\n");
    std::abort();
  }
  return (grid.*(&CellAccess::correspondence_cost_cells))();
}

cmx_grid2d_limits LimitsOf(const Grid2D& grid) {
  const MapLimits& l = grid.limits();
  return cmx_grid2d_limits{l.resolution(), l.max().x(), l.max().y(),
                           l.cell_limits().num_x_cells, l.cell_limits().num_y_cells,
                           grid.GetMinCorrespondenceCost(), grid.GetMaxCorrespondenceCost()};
}

// The TSDF2D weight plane is private; ToProto() (tsdf_2d.cc:113-124) is its public image.
struct TsdfPlanes {
  std::vector<uint16_t> weights;
  float truncation_distance = 0.f, max_weight = 0.f;
  explicit TsdfPlanes(const TSDF2D& tsdf) {
    const mapping::proto::Grid2D proto = tsdf.ToProto();
    for (const auto w : proto.tsdf_2d().weight_cells()) weights.push_back(static_cast<uint16_t>(w));
    truncation_distance = proto.tsdf_2d().truncation_distance();
    max_weight = proto.tsdf_2d().max_weight();
  }
};

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

}  // namespace

RealTimeCorrelativeScanMatcher2D::RealTimeCorrelativeScanMatcher2D(
    const proto::RealTimeCorrelativeScanMatcherOptions& options)
    : options_(options) {}

double RealTimeCorrelativeScanMatcher2D::Match(const transform::Rigid2d& initial_pose_estimate,
                                               const sensor::PointCloud& point_cloud,
                                               const Grid2D& grid,
                                               transform::Rigid2d* pose_estimate) const {
  CHECK(pose_estimate != nullptr);
  const cmx_rt_options o = OptionsOf(options_);
  const cmx_grid2d_limits limits = LimitsOf(grid);
  const cmx_pose2d init{initial_pose_estimate.translation().x(),
                        initial_pose_estimate.translation().y(),
                        initial_pose_estimate.rotation().angle()};
  const std::vector<float> xyz = Flatten(point_cloud);
  double score = 0.;
  cmx_pose2d pose{};
  if (const auto* resident = dynamic_cast<const dropin::DeviceGrid2DView*>(&grid)) {
    // the submap's grid is in HBM already (device_grids.h): only the scan crosses PCIe
    CheckOk(cmx_rt2d_match_grid(&o, resident->device_grid(), &init, xyz.data(),
                                static_cast<int32_t>(point_cloud.size()), &score, &pose, nullptr),
            "cmx_rt2d_match_grid");
    *pose_estimate = transform::Rigid2d({pose.x, pose.y}, pose.theta);
    return score;
  }
  switch (grid.GetGridType()) {
    case GridType::PROBABILITY_GRID:
      CheckOk(cmx_rt2d_match(&o, &limits, CellsOf(grid).data(), &init, xyz.data(),
                             static_cast<int32_t>(point_cloud.size()), Device(), &score, &pose,
                             nullptr),
              "cmx_rt2d_match");
      break;
    case GridType::TSDF: {
      const TsdfPlanes planes(static_cast<const TSDF2D&>(grid));
      CheckOk(cmx_rt2d_match_tsdf(&o, &limits, CellsOf(grid).data(), planes.weights.data(),
                                  planes.truncation_distance, planes.max_weight, &init,
                                  xyz.data(), static_cast<int32_t>(point_cloud.size()), Device(),
                                  &score, &pose, nullptr),
              "cmx_rt2d_match_tsdf");
      break;
    }
  }
  *pose_estimate = transform::Rigid2d({pose.x, pose.y}, pose.theta);
  return score;
}

void RealTimeCorrelativeScanMatcher2D::ScoreCandidates(
    const Grid2D& grid, const std::vector<DiscreteScan2D>& discrete_scans,
    const SearchParameters& /*search_parameters*/, std::vector<Candidate2D>* const candidates) const {
  const cmx_rt_options o = OptionsOf(options_);
  const cmx_grid2d_limits limits = LimitsOf(grid);
  std::vector<int32_t> xy, begin{0};
  for (const DiscreteScan2D& scan : discrete_scans) {
    for (const Eigen::Array2i& cell : scan) {
      xy.push_back(cell.x());
      xy.push_back(cell.y());
    }
    begin.push_back(static_cast<int32_t>(xy.size() / 2));
  }
  std::vector<cmx_candidate2d> flat;
  flat.reserve(candidates->size());
  for (const Candidate2D& c : *candidates)
    flat.push_back(cmx_candidate2d{c.scan_index, c.x_index_offset, c.y_index_offset, 0.f, c.x, c.y,
                                   c.orientation});
  std::vector<uint16_t> weights;
  float truncation_distance = 0.f, max_weight = 0.f;
  if (grid.GetGridType() == GridType::TSDF) {
    const TsdfPlanes planes(static_cast<const TSDF2D&>(grid));
    weights = planes.weights;
    truncation_distance = planes.truncation_distance;
    max_weight = planes.max_weight;
  }
  CheckOk(cmx_rt2d_score_candidates(&o, &limits, CellsOf(grid).data(),
                                    weights.empty() ? nullptr : weights.data(),
                                    truncation_distance, max_weight, xy.data(), begin.data(),
                                    static_cast<int32_t>(discrete_scans.size()), flat.data(),
                                    static_cast<int32_t>(flat.size()), Device()),
          "cmx_rt2d_score_candidates");
  for (size_t i = 0; i != flat.size(); ++i) (*candidates)[i].score = flat[i].score;
}

RealTimeCorrelativeScanMatcher3D::RealTimeCorrelativeScanMatcher3D(
    const scan_matching::proto::RealTimeCorrelativeScanMatcherOptions& options)
    : options_(options) {}

float RealTimeCorrelativeScanMatcher3D::Match(const transform::Rigid3d& initial_pose_estimate,
                                              const sensor::PointCloud& point_cloud,
                                              const HybridGrid& hybrid_grid,
                                              transform::Rigid3d* pose_estimate) const {
  CHECK(pose_estimate != nullptr);
  const cmx_rt_options o = OptionsOf(options_);
  if (const cmx_grid3d* resident = dropin::DeviceGridOf(&hybrid_grid)) {
    const cmx_pose3d init{
        {initial_pose_estimate.translation().x(), initial_pose_estimate.translation().y(),
         initial_pose_estimate.translation().z()},
        {initial_pose_estimate.rotation().w(), initial_pose_estimate.rotation().x(),
         initial_pose_estimate.rotation().y(), initial_pose_estimate.rotation().z()}};
    const std::vector<float> xyz = Flatten(point_cloud);
    float score = 0.f;
    cmx_pose3d pose{};
    CheckOk(cmx_rt3d_match_grid(&o, resident, &init, xyz.data(),
                                static_cast<int32_t>(point_cloud.size()), &score, &pose, nullptr),
            "cmx_rt3d_match_grid");
    *pose_estimate = transform::Rigid3d(Eigen::Vector3d(pose.t[0], pose.t[1], pose.t[2]),
                                        Eigen::Quaterniond(pose.q[0], pose.q[1], pose.q[2], pose.q[3]));
    return score;
  }
  std::vector<cmx_voxel> voxels;             // what HybridGrid::Iterator yields (hybrid_grid.h:304-372)
  for (auto it = HybridGrid::Iterator(hybrid_grid); !it.Done(); it.Next()) {
    const Eigen::Array3i index = it.GetCellIndex();
    voxels.push_back(cmx_voxel{index.x(), index.y(), index.z(), it.GetValue(), 0});
  }
  const cmx_pose3d init{
      {initial_pose_estimate.translation().x(), initial_pose_estimate.translation().y(),
       initial_pose_estimate.translation().z()},
      {initial_pose_estimate.rotation().w(), initial_pose_estimate.rotation().x(),
       initial_pose_estimate.rotation().y(), initial_pose_estimate.rotation().z()}};
  const std::vector<float> xyz = Flatten(point_cloud);
  float score = 0.f;
  cmx_pose3d pose{};
  CheckOk(cmx_rt3d_match(&o, hybrid_grid.resolution(), voxels.data(),
                         static_cast<int64_t>(voxels.size()), &init, xyz.data(),
                         static_cast<int32_t>(point_cloud.size()), Device(), &score, &pose, nullptr),
          "cmx_rt3d_match");
  *pose_estimate = transform::Rigid3d(Eigen::Vector3d(pose.t[0], pose.t[1], pose.t[2]),
                                      Eigen::Quaterniond(pose.q[0], pose.q[1], pose.q[2], pose.q[3]));
  return score;
}

}  // namespace scan_matching
}  // namespace mapping
}  // namespace cartographer
