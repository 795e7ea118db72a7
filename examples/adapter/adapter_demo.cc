// Drives the adapter classes exactly the way the reference's callers do
// (ConstraintBuilder2D::ComputeConstraint, LocalTrajectoryBuilder2D::ScanMatch) on a grid and a
// scan read from a file the test writes:  adapter_demo <input.bin>
//   int32 nx, ny, n; float64 resolution, max_x, max_y, init_x, init_y, init_theta;
//   uint16 cells[ny*nx]; float32 xyz[3*n]
// Prints one line per matcher; tests/test_adapter.py compares them with the ctypes path.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "scan_matchers_2d_mi355x.h"

using namespace cartographer;

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int32_t dims[3];
  double v[6];
  if (std::fread(dims, sizeof(int32_t), 3, f) != 3 || std::fread(v, sizeof(double), 6, f) != 6)
    return 2;
  std::vector<uint16_t> cells(static_cast<size_t>(dims[0]) * dims[1]);
  std::vector<float> xyz(3 * static_cast<size_t>(dims[2]));
  if (std::fread(cells.data(), 2, cells.size(), f) != cells.size() ||
      std::fread(xyz.data(), 4, xyz.size(), f) != xyz.size())
    return 2;
  std::fclose(f);
  if (cmx_device_count() < 1) {
    std::printf("no device\n");                 // there is no CPU fallback
    return 0;
  }
  const mapping::Grid2D grid(mapping::MapLimits(v[0], v[1], v[2], {dims[0], dims[1]}),
                             std::move(cells));
  sensor::PointCloud cloud;
  for (int i = 0; i < dims[2]; ++i)
    cloud.push_back({{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]}});
  const transform::Rigid2d initial({v[3], v[4]}, v[5]);

  const mapping::scan_matching::FastCorrelativeScanMatcher2D fast(grid, {7., 0.5, 6});
  float score = -1.f;
  transform::Rigid2d pose;
  bool ok = fast.MatchFullSubmap(cloud, 0.5f, &score, &pose);
  std::printf("full %d %.9g %.17g %.17g %.17g\n", ok, score, pose.translation().x(),
              pose.translation().y(), pose.rotation().angle());
  score = -1.f;
  ok = fast.Match(initial, cloud, 0.5f, &score, &pose);
  std::printf("window %d %.9g %.17g %.17g %.17g\n", ok, score, pose.translation().x(),
              pose.translation().y(), pose.rotation().angle());
  ok = fast.Match(initial, cloud, 0.999f, &score, &pose);     // nothing scores that high
  std::printf("none %d\n", ok);
  const mapping::scan_matching::RealTimeCorrelativeScanMatcher2D rt({0.3, 0.12, 0.1, 0.1});
  const double rt_score = rt.Match(initial, cloud, grid, &pose);
  std::printf("rt %.17g %.17g %.17g %.17g\n", rt_score, pose.translation().x(),
              pose.translation().y(), pose.rotation().angle());
  return 0;
}
