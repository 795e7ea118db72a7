// Bodies of sensor::VoxelFilter(PointCloud) and sensor::AdaptiveVoxelFilter over
// libcartographer_mi355x, compiled against the reference's REAL sensor/internal/voxel_filter.h:
// the source a maintainer swaps for sensor/internal/voxel_filter.cc in a build whose local
// trajectory builder filters on the device (2d/local_trajectory_builder_2d.cc:59-61,227-229
// and 3d/local_trajectory_builder_3d.cc:158-159,250-252,281-296 compile unmodified).  The
// range-measurement overload (the 3D builder's first filter) runs the same device filter on the
// positions (cmx_voxel_filter_indices); the two remaining overloads of the header have no caller on
// either path and are not defined here.  Every overload asks the device WHICH points it kept
// (cmx_voxel_filter_indices, cmx_adaptive_voxel_filter_indices) and selects the payload -- the
// points themselves and, for a PointCloud, its intensities (voxel_filter.cc:138-161) -- here.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cartographer/sensor/internal/voxel_filter.h"
#include "cartographer_mi355x.h"

namespace cartographer {
namespace sensor {
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

std::vector<float> Flatten(const PointCloud& cloud) {
  std::vector<float> xyz;
  xyz.reserve(3 * cloud.size());
  for (const RangefinderPoint& p : cloud) {
    xyz.push_back(p.position.x());
    xyz.push_back(p.position.y());
    xyz.push_back(p.position.z());
  }
  return xyz;
}

// The points the device filter kept, WITH their intensities when the cloud carries any
// (voxel_filter.cc:138-161: PointCloud(filtered_points, filtered_intensities)): the device says
// which points, the payload is selected here.
PointCloud Select(const PointCloud& cloud, const std::vector<int32_t>& kept, const int32_t count) {
  std::vector<RangefinderPoint> points;
  points.reserve(count);
  for (int32_t k = 0; k != count; ++k) points.push_back(cloud[kept[k]]);
  std::vector<float> intensities;
  if (!cloud.intensities().empty()) {
    if (cloud.intensities().size() != cloud.size()) {
      std::fprintf(stderr, "Check failed: a point cloud with %zu points and %zu intensities\n",
                   cloud.size(), cloud.intensities().size());
      std::abort();
    }
    intensities.reserve(count);
    for (int32_t k = 0; k != count; ++k) intensities.push_back(cloud.intensities()[kept[k]]);
  }
  return PointCloud(std::move(points), std::move(intensities));
}

}  // namespace

PointCloud VoxelFilter(const PointCloud& point_cloud, const float resolution) {
  if (point_cloud.empty()) return PointCloud();
  const std::vector<float> xyz = Flatten(point_cloud);
  std::vector<int32_t> kept(point_cloud.size());
  int32_t count = 0;
  CheckOk(cmx_voxel_filter_indices(xyz.data(), static_cast<int32_t>(point_cloud.size()), resolution,
                                   Device(), kept.data(), &count),
          "cmx_voxel_filter_indices");
  return Select(point_cloud, kept, count);
}

// sensor::VoxelFilter over range measurements (voxel_filter.cc:176-191): the same filter on
// their positions; cmx_voxel_filter_indices says which measurements it kept.
std::vector<TimedPointCloudOriginData::RangeMeasurement> VoxelFilter(
    const std::vector<TimedPointCloudOriginData::RangeMeasurement>& range_measurements,
    const float resolution) {
  std::vector<TimedPointCloudOriginData::RangeMeasurement> results;
  if (range_measurements.empty()) return results;
  std::vector<float> xyz;
  xyz.reserve(3 * range_measurements.size());
  for (const auto& m : range_measurements) {
    xyz.push_back(m.point_time.position.x());
    xyz.push_back(m.point_time.position.y());
    xyz.push_back(m.point_time.position.z());
  }
  std::vector<int32_t> kept(range_measurements.size());
  int32_t count = 0;
  CheckOk(cmx_voxel_filter_indices(xyz.data(), static_cast<int32_t>(range_measurements.size()),
                                   resolution, Device(), kept.data(), &count),
          "cmx_voxel_filter_indices");
  results.reserve(count);
  for (int32_t k = 0; k != count; ++k) results.push_back(range_measurements[kept[k]]);
  return results;
}

PointCloud AdaptiveVoxelFilter(const PointCloud& point_cloud,
                               const proto::AdaptiveVoxelFilterOptions& options) {
  if (point_cloud.empty()) return PointCloud();
  const std::vector<float> xyz = Flatten(point_cloud);
  std::vector<int32_t> kept(point_cloud.size());
  int32_t count = 0;
  CheckOk(cmx_adaptive_voxel_filter_indices(xyz.data(), static_cast<int32_t>(point_cloud.size()),
                                            options.max_length(), options.min_num_points(),
                                            options.max_range(), Device(), kept.data(), &count),
          "cmx_adaptive_voxel_filter_indices");
  return Select(point_cloud, kept, count);
}

}  // namespace sensor
}  // namespace cartographer
