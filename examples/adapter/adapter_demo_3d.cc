// Drives the 3D adapter classes the way ConstraintBuilder3D / LocalTrajectoryBuilder3D do, on a
// case the test writes:  adapter_demo_3d <input.bin>
//   int32 num_voxels, grid_size, n_hi, n_lo, hist_size; float32 resolution;
//   float64 node_pose[7] (t, q wxyz);  {int32 x,y,z; uint16 value; uint16 pad}[num_voxels];
//   float32 hi[3*n_hi]; float32 lo[3*n_lo]; float32 hist[hist_size]
#include <cstdio>
#include <vector>

#include "scan_matchers_3d_mi355x.h"

using namespace cartographer;

static sensor::PointCloud Cloud(const std::vector<float>& xyz) {
  sensor::PointCloud c;
  for (size_t i = 0; i + 2 < xyz.size(); i += 3) c.push_back({{xyz[i], xyz[i + 1], xyz[i + 2]}});
  return c;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t h[5];
  float resolution;
  double pose[7];
  if (std::fread(h, 4, 5, f) != 5 || std::fread(&resolution, 4, 1, f) != 1 ||
      std::fread(pose, 8, 7, f) != 7)
    return 2;
  std::vector<cmx_voxel> raw(h[0]);
  std::vector<float> hi(3 * static_cast<size_t>(h[2])), lo(3 * static_cast<size_t>(h[3])),
      hist(h[4]);
  if (std::fread(raw.data(), sizeof(cmx_voxel), raw.size(), f) != raw.size() ||
      std::fread(hi.data(), 4, hi.size(), f) != hi.size() ||
      std::fread(lo.data(), 4, lo.size(), f) != lo.size() ||
      std::fread(hist.data(), 4, hist.size(), f) != hist.size())
    return 2;
  std::fclose(f);
  if (cmx_device_count() < 1) {
    std::printf("no device\n");
    return 0;
  }
  std::vector<mapping::HybridGrid::Voxel> voxels;
  for (const cmx_voxel& v : raw) voxels.push_back({{v.x, v.y, v.z}, v.value});
  const mapping::HybridGrid grid(resolution, h[1], voxels);
  mapping::TrajectoryNodeData data;
  data.high_resolution_point_cloud = Cloud(hi);
  data.low_resolution_point_cloud = Cloud(lo);
  data.rotational_scan_matcher_histogram = hist;
  const transform::Rigid3d node({pose[0], pose[1], pose[2]}, {pose[3], pose[4], pose[5], pose[6]});

  const mapping::scan_matching::FastCorrelativeScanMatcher3D fast(
      grid, &grid, &hist, {5, 2, 0.5, 0.25, 1.0, 0.5, 0.1});
  const auto result = fast.Match(node, transform::Rigid3d(), data, 0.3f);
  if (result) {
    std::printf("fast 1 %.9g %.9g %.9g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", result->score,
                result->rotational_score, result->low_resolution_score,
                result->pose_estimate.translation().x(), result->pose_estimate.translation().y(),
                result->pose_estimate.translation().z(), result->pose_estimate.rotation().w(),
                result->pose_estimate.rotation().x(), result->pose_estimate.rotation().y(),
                result->pose_estimate.rotation().z());
  } else {
    std::printf("fast 0\n");
  }
  const auto none = fast.Match(node, transform::Rigid3d(), data, 0.999f);
  std::printf("none %d\n", none ? 1 : 0);
  const mapping::scan_matching::RealTimeCorrelativeScanMatcher3D rt({0.1, 0.02, 0.1, 0.1});
  transform::Rigid3d estimate;
  const float score = rt.Match(node, data.high_resolution_point_cloud, grid, &estimate);
  std::printf("rt %.9g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", score,
              estimate.translation().x(), estimate.translation().y(), estimate.translation().z(),
              estimate.rotation().w(), estimate.rotation().x(), estimate.rotation().y(),
              estimate.rotation().z());
  return 0;
}
