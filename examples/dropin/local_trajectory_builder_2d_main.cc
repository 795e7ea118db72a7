// A simulated 2D lidar drive through the reference's LocalTrajectoryBuilder2D
// (mapping/internal/2d/local_trajectory_builder_2d.cc, compiled UNMODIFIED where it lies): range
// data collation, pose extrapolation, gravity alignment, voxel filtering, the real-time
// correlative matcher, the Ceres matcher, the motion filter and submap insertion, scan after scan.
//
// The same file is linked twice (Makefile):
//   _build/local_trajectory_builder_2d_reference  with the reference's own
//       real_time_correlative_scan_matcher_2d.cc, ceres_scan_matcher_2d.cc (over the stand-in
//       solver of oracle/ref_shims/ceres) and sensor/internal/voxel_filter.cc      (CPU, anywhere)
//   _build/local_trajectory_builder_2d_mi355x     with real_time_matchers_mi355x.cc,
//       scan_matchers_2d_mi355x.cc and voxel_filter_mi355x.cc over the library     (MI355X)
// Each prints one line per scan (estimated pose, true pose, points matched, submaps inserted into)
// and a digest of the active submaps' grids at the end; tests/test_dropin.py compares the two.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "cartographer/mapping/internal/2d/local_trajectory_builder_2d.h"
#include "cartographer/metrics/family_factory.h"
#include "cartographer/transform/transform.h"

using namespace cartographer;
using namespace cartographer::mapping;

namespace {

struct Segment {
  double ax, ay, bx, by;
};

// A hall of 30 m x 20 m with pillars and a partition: most of it is out of the lidar's 12 m reach
// from any one place, so scans have returns and misses.
std::vector<Segment> World() {
  std::vector<Segment> w;
  auto box = [&w](double x0, double y0, double x1, double y1) {
    w.push_back({x0, y0, x1, y0});
    w.push_back({x1, y0, x1, y1});
    w.push_back({x1, y1, x0, y1});
    w.push_back({x0, y1, x0, y0});
  };
  box(-6., -8., 24., 12.);
  box(2., 2.5, 3., 3.5);
  box(6.5, -3.5, 7.5, -2.);
  box(-3., -4., -2.2, -3.2);
  box(10., 4., 10.6, 7.);
  box(3.5, -6., 5.5, -5.6);
  w.push_back({12., -8., 12., -1.});
  w.push_back({-6., 5., -1., 7.5});
  return w;
}

// Distance along the ray from (ox, oy) in direction (dx, dy) to the nearest wall, or -1.
double Cast(const std::vector<Segment>& world, double ox, double oy, double dx, double dy) {
  double best = -1.;
  for (const Segment& s : world) {
    const double ex = s.bx - s.ax, ey = s.by - s.ay;
    const double det = ex * dy - ey * dx;
    if (std::abs(det) < 1e-12) continue;
    const double wx = s.ax - ox, wy = s.ay - oy;
    const double t = (ex * wy - ey * wx) / det;      // along the ray
    const double u = (dx * wy - dy * wx) / det;      // along the segment
    if (t > 1e-6 && u >= 0. && u <= 1. && (best < 0. || t < best)) best = t;
  }
  return best;
}

struct Pose {
  double x, y, yaw;
};

// The drive: a slalom, pulling away from rest to 0.5 m/s (the builder starts without a velocity
// estimate, as it does on a robot), sampled at the 10 Hz of the lidar.
Pose TruePose(double t) {
  const double s = t - (1. - std::exp(-t));          // path parameter: speed 1 - exp(-t)
  const double x = 0.5 * s;
  const double y = 1.2 * std::sin(0.35 * s);
  return Pose{x, y, std::atan2(1.2 * 0.35 * std::cos(0.35 * s), 0.5)};
}

struct Noise {   // a fixed stream in [-1, 1): the scans are the same in every build
  uint32_t state = 12345u;
  double Next() {
    state = state * 1664525u + 1013904223u;
    return (state >> 8) * (2. / 16777216.) - 1.;
  }
};

proto::LocalTrajectoryBuilderOptions2D Options() {
  // configuration_files/trajectory_builder_2d.lua, with the online correlative matcher switched on
  // and submaps of 8 scans so that the drive finishes submaps and starts new ones.
  proto::LocalTrajectoryBuilderOptions2D o;
  o.set_min_range(0.f);
  o.set_max_range(30.f);
  o.set_min_z(-0.8f);
  o.set_max_z(2.f);
  o.set_missing_data_ray_length(5.f);
  o.set_num_accumulated_range_data(1);
  o.set_voxel_filter_size(0.025f);
  o.set_use_online_correlative_scan_matching(true);
  o.set_use_imu_data(false);
  o.mutable_adaptive_voxel_filter_options()->set_max_length(0.5f);
  o.mutable_adaptive_voxel_filter_options()->set_min_num_points(200.f);
  o.mutable_adaptive_voxel_filter_options()->set_max_range(50.f);
  auto* rt = o.mutable_real_time_correlative_scan_matcher_options();
  rt->set_linear_search_window(0.1);
  rt->set_angular_search_window(20. * M_PI / 180.);
  rt->set_translation_delta_cost_weight(1e-1);
  rt->set_rotation_delta_cost_weight(1e-1);
  auto* ceres = o.mutable_ceres_scan_matcher_options();
  ceres->set_occupied_space_weight(1.);
  ceres->set_translation_weight(10.);
  ceres->set_rotation_weight(40.);
  ceres->mutable_ceres_solver_options()->set_use_nonmonotonic_steps(false);
  ceres->mutable_ceres_solver_options()->set_max_num_iterations(20);
  ceres->mutable_ceres_solver_options()->set_num_threads(1);
  o.mutable_motion_filter_options()->set_max_time_seconds(5.);
  o.mutable_motion_filter_options()->set_max_distance_meters(0.2);
  o.mutable_motion_filter_options()->set_max_angle_radians(M_PI / 180.);
  o.mutable_pose_extrapolator_options()->set_use_imu_based(false);
  o.mutable_pose_extrapolator_options()->mutable_constant_velocity()->set_pose_queue_duration(
      0.001);
  o.mutable_pose_extrapolator_options()
      ->mutable_constant_velocity()
      ->set_imu_gravity_time_constant(10.);
  auto* submaps = o.mutable_submaps_options();
  submaps->set_num_range_data(8);
  submaps->mutable_grid_options_2d()->set_grid_type(
      proto::GridOptions2D_GridType_PROBABILITY_GRID);
  submaps->mutable_grid_options_2d()->set_resolution(0.05);
  auto* inserter = submaps->mutable_probability_grid_range_data_inserter_options_2d();
  inserter->set_hit_probability(0.55);
  inserter->set_miss_probability(0.49);
  inserter->set_insert_free_space(true);
  return o;
}

}  // namespace

int main(int argc, char** argv) {
  const int num_scans = argc > 1 ? std::atoi(argv[1]) : 80;
  const int num_rays = argc > 2 ? std::atoi(argv[2]) : 720;
  const double reach = 12., scan_period = 0.1, sweep = 0.05;
  const std::vector<Segment> world = World();
  Noise noise;

  metrics::FamilyFactory* null_factory = nullptr;
  (void)null_factory;   // the metrics stay at their null objects (local_trajectory_builder_2d.cc:29-35)
  LocalTrajectoryBuilder2D builder(Options(), {"scan"});
  const common::Time start = common::FromUniversal(636727077355276250);

  std::shared_ptr<const Submap2D> last_front, last_back;
  int num_results = 0, num_insertions = 0;
  Pose first{0., 0., 0.};
  double worst = 0., seconds = 0.;
  std::vector<double> per_call;
  for (int k = 0; k != num_scans; ++k) {
    const double t_end = k * scan_period;
    sensor::TimedPointCloudData scan;
    scan.time = start + common::FromSeconds(t_end);
    scan.origin = Eigen::Vector3f::Zero();
    for (int r = 0; r != num_rays; ++r) {
      // The lidar turns once per `sweep` seconds, the last ray at the scan's time stamp; the
      // measurement is taken from where the robot is at that moment, expressed in its frame then.
      const double dt = -sweep * (num_rays - 1 - r) / num_rays;
      const Pose at = TruePose(t_end + dt);
      const double bearing = -M_PI + 2. * M_PI * r / num_rays;
      const double dx = std::cos(at.yaw + bearing), dy = std::sin(at.yaw + bearing);
      double range = Cast(world, at.x, at.y, dx, dy);
      if (range < 0. || range > reach) {
        range = 35.;                                   // nothing seen: becomes a miss
      } else {
        range += 0.01 * noise.Next();
      }
      scan.ranges.push_back(sensor::TimedRangefinderPoint{
          Eigen::Vector3f(static_cast<float>(range * std::cos(bearing)),
                          static_cast<float>(range * std::sin(bearing)), 0.f),
          static_cast<float>(dt)});
    }
    const auto t0 = std::chrono::steady_clock::now();
    const std::unique_ptr<LocalTrajectoryBuilder2D::MatchingResult> result =
        builder.AddRangeData("scan", scan);
    const double call_seconds =
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    seconds += call_seconds;
    per_call.push_back(call_seconds);
    if (result == nullptr) {
      std::printf("scan %3d  no result\n", k);
      continue;
    }
    ++num_results;
    const double t_result = common::ToSeconds(result->time - start);
    const Pose truth = TruePose(t_result);
    // The first scan only initialises the extrapolator; the second is "matched" against no submap
    // and inserted at the identity: the local frame is the robot's frame at THAT scan.
    if (num_results == 1) first = truth;
    const double c = std::cos(-first.yaw), s = std::sin(-first.yaw);
    const double tx = c * (truth.x - first.x) - s * (truth.y - first.y);
    const double ty = s * (truth.x - first.x) + c * (truth.y - first.y);
    const transform::Rigid2d estimate = transform::Project2D(result->local_pose);
    const double error = std::hypot(estimate.translation().x() - tx, estimate.translation().y() - ty);
    if (error > worst) worst = error;
    int inserted_into = 0;
    if (result->insertion_result != nullptr) {
      ++num_insertions;
      inserted_into = static_cast<int>(result->insertion_result->insertion_submaps.size());
      last_front = result->insertion_result->insertion_submaps.front();
      last_back = result->insertion_result->insertion_submaps.back();
    }
    std::printf("scan %3d  t %.2f  pose %.9f %.9f %.9f  truth %.6f %.6f %.6f  points %zu  inserted %d\n",
                k, t_result, estimate.translation().x(), estimate.translation().y(),
                estimate.rotation().angle(), tx, ty, truth.yaw - first.yaw,
                result->insertion_result != nullptr
                    ? result->insertion_result->constant_data->filtered_gravity_aligned_point_cloud.size()
                    : size_t(0),
                inserted_into);
  }
  for (const auto& submap : {last_front, last_back}) {
    if (submap == nullptr) continue;
    const Grid2D& grid = *submap->grid();
    DropinSyncGridToHost(grid);
    const CellLimits& cells = grid.limits().cell_limits();
    int known = 0;
    double sum = 0.;
    for (int y = 0; y != cells.num_y_cells; ++y)
      for (int x = 0; x != cells.num_x_cells; ++x)
        if (grid.IsKnown({x, y})) {
          ++known;
          sum += grid.GetCorrespondenceCost({x, y});
        }
    std::printf("submap  scans %d  finished %d  cells %d x %d  known %d  cost_sum %.6f\n",
                submap->num_range_data(), submap->insertion_finished() ? 1 : 0, cells.num_x_cells,
                cells.num_y_cells, known, sum);
  }
  std::printf("results %d  insertions %d  worst_position_error %.6f\n", num_results, num_insertions,
              worst);
  // (the mean includes what the first calls pay once -- runtime start-up, code objects, the first
  // allocations: 150 ms on a device build, nothing on the CPU; the median is a call's own time)
  std::sort(per_call.begin(), per_call.end());
  std::fprintf(stderr, "%.3f ms per AddRangeData (%d scans of %d rays); median %.3f ms\n",
               1e3 * seconds / num_scans, num_scans, num_rays,
               per_call.empty() ? 0. : 1e3 * per_call[per_call.size() / 2]);
  return worst < 0.15 ? 0 : 1;
}
