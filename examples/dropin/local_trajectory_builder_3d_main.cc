// A simulated 3D lidar + IMU drive through the reference's LocalTrajectoryBuilder3D
// (mapping/internal/3d/local_trajectory_builder_3d.cc, compiled UNMODIFIED where it lies): range
// data collation, the per-point pose extrapolation, three voxel filters, the real-time correlative
// matcher, the Ceres matcher on the high- and low-resolution grids, the motion filter, the
// rotational histogram and the insertion into both hybrid grids, scan after scan.
//
// The same file is linked twice (Makefile):
//   _build/local_trajectory_builder_3d_reference  with the reference's own
//       real_time_correlative_scan_matcher_3d.cc, ceres_scan_matcher_3d.cc (over the stand-in
//       solver of oracle/ref_shims/ceres) and sensor/internal/voxel_filter.cc      (CPU, anywhere)
//   _build/local_trajectory_builder_3d_mi355x     with real_time_matchers_mi355x.cc,
//       scan_matchers_3d_mi355x.cc and voxel_filter_mi355x.cc over the library     (MI355X)
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

#include "cartographer/mapping/internal/3d/local_trajectory_builder_3d.h"
#include "cartographer/transform/transform.h"

using namespace cartographer;
using namespace cartographer::mapping;

namespace {

struct Box {
  double lo[3], hi[3];
};

// A hall of 24 m x 16 m x 5 m with pillars, a gallery and crates.
const Box kHall{{-6., -7., 0.}, {18., 9., 5.}};
std::vector<Box> Obstacles() {
  return {Box{{2., 2.5, 0.}, {3., 3.5, 5.}},      Box{{6.5, -3.5, 0.}, {7.5, -2., 5.}},
          Box{{-3., -4., 0.}, {-2.2, -3.2, 2.}},  Box{{10., 4., 0.}, {10.6, 7., 3.}},
          Box{{3.5, -6., 0.}, {5.5, -5.6, 1.5}},  Box{{-6., 6., 2.5}, {18., 9., 2.8}},
          Box{{12., -7., 0.}, {12.3, -1., 5.}},   Box{{14., 1., 0.}, {15.5, 2.5, 1.2}},
          // ledges and sills along the walls: what ties the height down for a lidar that mostly
          // sees walls
          Box{{-6., -7., 3.2}, {18., -6.6, 3.5}}, Box{{-6., -7., 1.4}, {18., -6.8, 1.6}},
          Box{{-6., -7., 2.2}, {-5.6, 9., 2.5}},  Box{{17.6, -7., 1.8}, {18., 9., 2.1}},
          Box{{17.7, -7., 3.6}, {18., 9., 3.8}},  Box{{-6., 8.7, 1.0}, {18., 9., 1.3}}};
}

// Where the ray origin + t * dir (t > 0) first meets a surface: the hall from the inside, the
// obstacles from the outside.
double Cast(const std::vector<Box>& obstacles, const double o[3], const double d[3]) {
  double best = 1e30;
  for (int a = 0; a != 3; ++a) {                       // leaving the hall
    if (std::abs(d[a]) < 1e-12) continue;
    const double t = ((d[a] > 0. ? kHall.hi[a] : kHall.lo[a]) - o[a]) / d[a];
    if (t > 0. && t < best) best = t;
  }
  for (const Box& b : obstacles) {                     // entering an obstacle (slab test)
    double t0 = 0., t1 = best;
    bool hit = true;
    for (int a = 0; a != 3 && hit; ++a) {
      if (std::abs(d[a]) < 1e-12) {
        hit = o[a] >= b.lo[a] && o[a] <= b.hi[a];
        continue;
      }
      double ta = (b.lo[a] - o[a]) / d[a], tb = (b.hi[a] - o[a]) / d[a];
      if (ta > tb) std::swap(ta, tb);
      t0 = std::max(t0, ta);
      t1 = std::min(t1, tb);
      hit = t0 <= t1;
    }
    if (hit && t0 > 1e-6 && t0 < best) best = t0;
  }
  return best;
}

struct Pose {
  double x, y, yaw;
};

// The drive: a slalom on the floor, pulling away from rest towards 0.5 m/s (time constant 3 s);
// the lidar sits 2.5 m up.
Pose TruePose(double t) {
  if (t < 0.) t = 0.;
  const double s = t - 3. * (1. - std::exp(-t / 3.));
  return Pose{0.5 * s, 1.2 * std::sin(0.35 * s), std::atan2(1.2 * 0.35 * std::cos(0.35 * s), 0.5)};
}

struct Noise {   // a fixed stream in [-1, 1): the scans are the same in every build
  uint32_t state = 12345u;
  double Next() {
    state = state * 1664525u + 1013904223u;
    return (state >> 8) * (2. / 16777216.) - 1.;
  }
};

proto::LocalTrajectoryBuilderOptions3D Options() {
  // configuration_files/trajectory_builder_3d.lua, with the online correlative matcher switched on
  // and submaps of 10 scans so that the drive finishes submaps and starts new ones.
  proto::LocalTrajectoryBuilderOptions3D o;
  o.set_min_range(1.f);
  o.set_max_range(60.f);
  o.set_num_accumulated_range_data(1);
  o.set_voxel_filter_size(0.15f);
  o.set_use_online_correlative_scan_matching(true);
  o.set_use_intensities(false);
  o.set_rotational_histogram_size(120);
  auto* high = o.mutable_high_resolution_adaptive_voxel_filter_options();
  high->set_max_length(1.f);
  high->set_min_num_points(400.f);
  high->set_max_range(15.f);
  auto* low = o.mutable_low_resolution_adaptive_voxel_filter_options();
  low->set_max_length(4.f);
  low->set_min_num_points(300.f);
  low->set_max_range(60.f);
  auto* rt = o.mutable_real_time_correlative_scan_matcher_options();
  rt->set_linear_search_window(0.15);
  rt->set_angular_search_window(M_PI / 180.);
  rt->set_translation_delta_cost_weight(1e-1);
  rt->set_rotation_delta_cost_weight(1e-1);
  auto* ceres = o.mutable_ceres_scan_matcher_options();
  ceres->add_occupied_space_weight(1.);
  ceres->add_occupied_space_weight(6.);
  ceres->set_translation_weight(5.);
  ceres->set_rotation_weight(4e2);
  ceres->set_only_optimize_yaw(false);
  ceres->mutable_ceres_solver_options()->set_use_nonmonotonic_steps(false);
  ceres->mutable_ceres_solver_options()->set_max_num_iterations(12);
  ceres->mutable_ceres_solver_options()->set_num_threads(1);
  o.mutable_motion_filter_options()->set_max_time_seconds(0.5);
  o.mutable_motion_filter_options()->set_max_distance_meters(0.1);
  o.mutable_motion_filter_options()->set_max_angle_radians(0.004);
  o.mutable_pose_extrapolator_options()->set_use_imu_based(false);
  o.mutable_pose_extrapolator_options()->mutable_constant_velocity()->set_pose_queue_duration(
      0.001);
  o.mutable_pose_extrapolator_options()
      ->mutable_constant_velocity()
      ->set_imu_gravity_time_constant(10.);
  auto* submaps = o.mutable_submaps_options();
  submaps->set_high_resolution(0.10);
  submaps->set_high_resolution_max_range(20.);
  submaps->set_low_resolution(0.45);
  submaps->set_num_range_data(10);
  auto* inserter = submaps->mutable_range_data_inserter_options();
  inserter->set_hit_probability(0.55);
  inserter->set_miss_probability(0.49);
  inserter->set_num_free_space_voxels(2);
  inserter->set_intensity_threshold(40.);
  return o;
}

void Digest(const char* name, const HybridGrid& grid) {
  int64_t voxels = 0, sum = 0;
  for (auto it = HybridGrid::Iterator(grid); !it.Done(); it.Next()) {
    ++voxels;
    sum += it.GetValue();
  }
  std::printf("  %s %.2f m: voxels %lld value_sum %lld\n", name, grid.resolution(),
              static_cast<long long>(voxels), static_cast<long long>(sum));
}

}  // namespace

int main(int argc, char** argv) {
  const int num_scans = argc > 1 ? std::atoi(argv[1]) : 60;
  const int num_beams = argc > 2 ? std::atoi(argv[2]) : 16;
  const int num_columns = argc > 3 ? std::atoi(argv[3]) : 450;
  const double scan_period = 0.1, sweep = 0.05, height = 2.5;
  const std::vector<Box> obstacles = Obstacles();
  Noise noise;

  LocalTrajectoryBuilder3D builder(Options(), {"lidar"});
  const common::Time start = common::FromUniversal(636727077355276250);

  std::shared_ptr<const Submap3D> last_front, last_back;
  int num_results = 0, num_insertions = 0;
  double worst = 0., seconds = 0.;
  std::vector<double> per_call;
  for (int k = 0; k != num_scans; ++k) {
    const double t_end = k * scan_period;
    // One IMU packet just before the sweep: the robot is level, gravity reads straight up, the
    // gyro reads the yaw rate of the drive.
    {
      const double t_imu = t_end - sweep - 0.01;
      const double yaw_rate = (TruePose(t_imu + 0.005).yaw - TruePose(t_imu - 0.005).yaw) / 0.01;
      builder.AddImuData(sensor::ImuData{start + common::FromSeconds(t_imu),
                                         Eigen::Vector3d(0., 0., 9.80665),
                                         Eigen::Vector3d(0., 0., yaw_rate)});
    }
    sensor::TimedPointCloudData scan;
    scan.time = start + common::FromSeconds(t_end);
    scan.origin = Eigen::Vector3f::Zero();
    const int num_rays = num_beams * num_columns;
    for (int c = 0; c != num_columns; ++c) {
      for (int b = 0; b != num_beams; ++b) {
        const int r = c * num_beams + b;
        const double dt = -sweep * (num_rays - 1 - r) / num_rays;
        const Pose at = TruePose(t_end + dt);
        const double bearing = -M_PI + 2. * M_PI * c / num_columns;
        const double elevation = (-10. + 20. * b / (num_beams - 1)) * M_PI / 180.;
        const double local[3] = {std::cos(elevation) * std::cos(bearing),
                                 std::cos(elevation) * std::sin(bearing), std::sin(elevation)};
        const double cy = std::cos(at.yaw), sy = std::sin(at.yaw);
        const double origin[3] = {at.x, at.y, height};
        const double dir[3] = {cy * local[0] - sy * local[1], sy * local[0] + cy * local[1], local[2]};
        const double range = Cast(obstacles, origin, dir) + 0.005 * noise.Next();
        scan.ranges.push_back(sensor::TimedRangefinderPoint{
            Eigen::Vector3f(static_cast<float>(range * local[0]), static_cast<float>(range * local[1]),
                            static_cast<float>(range * local[2])),
            static_cast<float>(dt)});
      }
    }
    const auto t0 = std::chrono::steady_clock::now();
    const std::unique_ptr<LocalTrajectoryBuilder3D::MatchingResult> result =
        builder.AddRangeData("lidar", scan);
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
    const Pose truth = TruePose(t_result), first = TruePose(0.);   // the local frame: the robot at rest
    const double c = std::cos(-first.yaw), s = std::sin(-first.yaw);
    const double tx = c * (truth.x - first.x) - s * (truth.y - first.y);
    const double ty = s * (truth.x - first.x) + c * (truth.y - first.y);
    const transform::Rigid3d& pose = result->local_pose;
    const double error = std::sqrt((pose.translation().x() - tx) * (pose.translation().x() - tx) +
                                   (pose.translation().y() - ty) * (pose.translation().y() - ty) +
                                   pose.translation().z() * pose.translation().z());
    if (error > worst) worst = error;
    int inserted_into = 0;
    size_t high_points = 0, low_points = 0;
    if (result->insertion_result != nullptr) {
      ++num_insertions;
      inserted_into = static_cast<int>(result->insertion_result->insertion_submaps.size());
      last_front = result->insertion_result->insertion_submaps.front();
      last_back = result->insertion_result->insertion_submaps.back();
      high_points = result->insertion_result->constant_data->high_resolution_point_cloud.size();
      low_points = result->insertion_result->constant_data->low_resolution_point_cloud.size();
    }
    std::printf("scan %3d  t %.2f  pose %.9f %.9f %.9f  q %.9f %.9f %.9f %.9f  truth %.6f %.6f %.6f  "
                "points %zu %zu  inserted %d\n",
                k, t_result, pose.translation().x(), pose.translation().y(), pose.translation().z(),
                pose.rotation().w(), pose.rotation().x(), pose.rotation().y(), pose.rotation().z(),
                tx, ty, truth.yaw - first.yaw, high_points, low_points, inserted_into);
  }
  for (const auto& submap : {last_front, last_back}) {
    if (submap == nullptr) continue;
    DropinSyncSubmapToHost(*submap);
    float histogram_sum = 0.f;
    for (int i = 0; i != submap->rotational_scan_matcher_histogram().size(); ++i)
      histogram_sum += submap->rotational_scan_matcher_histogram()(i);
    std::printf("submap  scans %d  finished %d  histogram_sum %.4f\n", submap->num_range_data(),
                submap->insertion_finished() ? 1 : 0, histogram_sum);
    Digest("high", submap->high_resolution_hybrid_grid());
    Digest("low", submap->low_resolution_hybrid_grid());
  }
  std::printf("results %d  insertions %d  worst_position_error %.6f\n", num_results, num_insertions,
              worst);
  // (the mean includes what the first calls pay once -- runtime start-up, code objects, the first
  // allocations: 150 ms on a device build, nothing on the CPU; the median is a call's own time)
  std::sort(per_call.begin(), per_call.end());
  std::fprintf(stderr, "%.3f ms per AddRangeData (%d scans of %d rays); median %.3f ms\n",
               1e3 * seconds / num_scans, num_scans, num_beams * num_columns,
               per_call.empty() ? 0. : 1e3 * per_call[per_call.size() / 2]);
  return worst < 0.3 ? 0 : 1;
}
