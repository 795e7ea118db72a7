// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_filters.h).
#include "oracle_filters.h"

#include <algorithm>
#include <cmath>
#include <map>
#include <random>
#include <unordered_map>

namespace oracle {
namespace {

// voxel_filter.cc:79-86: per-axis RoundToInt(p / resolution) in f32, converted to uint64
// (sign-extending) and packed with wrapping arithmetic.
uint64_t VoxelKey(const Point3f& p, float resolution) {
  const uint64_t x = RoundToInt(p.x / resolution);
  const uint64_t y = RoundToInt(p.y / resolution);
  const uint64_t z = RoundToInt(p.z / resolution);
  return (x << 42) + (y << 21) + z;
}

}  // namespace

std::vector<uint8_t> VoxelFilterFlags(const PointCloud& cloud, float resolution) {
  // voxel_filter.cc:88-115, reservoir sampling with a default-constructed minstd_rand0.
  std::minstd_rand0 generator;
  std::unordered_map<uint64_t, std::pair<int, int>> voxel_count_and_point_index;
  for (size_t i = 0; i < cloud.size(); i++) {
    auto& voxel = voxel_count_and_point_index[VoxelKey(cloud[i], resolution)];
    voxel.first++;
    if (voxel.first == 1) {
      voxel.second = static_cast<int>(i);
    } else {
      std::uniform_int_distribution<> distribution(1, voxel.first);
      if (distribution(generator) == voxel.first) voxel.second = static_cast<int>(i);
    }
  }
  std::vector<uint8_t> used(cloud.size(), 0);
  for (const auto& kv : voxel_count_and_point_index) used[kv.second.second] = 1;
  return used;
}

PointCloud VoxelFilter(const PointCloud& cloud, float resolution) {
  const std::vector<uint8_t> used = VoxelFilterFlags(cloud, resolution);
  PointCloud out;
  for (size_t i = 0; i < cloud.size(); ++i)
    if (used[i]) out.push_back(cloud[i]);
  return out;
}

PointCloud AdaptiveVoxelFilter(const PointCloud& cloud, float max_length, float min_num_points,
                               float max_range) {
  // FilterByMaxRange: position.norm() <= max_range, norm = sqrt((x^2 + y^2) + z^2) in f32.
  PointCloud in_range;
  for (const Point3f& p : cloud)
    if (std::sqrt((p.x * p.x + p.y * p.y) + p.z * p.z) <= max_range) in_range.push_back(p);
  // AdaptivelyVoxelFiltered (:38-75); min_num_points is a float field of the proto, the
  // comparisons are size_t vs float.
  if (in_range.size() <= min_num_points) return in_range;
  PointCloud result = VoxelFilter(in_range, max_length);
  if (result.size() >= min_num_points) return result;
  for (float high_length = max_length; high_length > 1e-2f * max_length; high_length /= 2.f) {
    float low_length = high_length / 2.f;
    result = VoxelFilter(in_range, low_length);
    if (result.size() >= min_num_points) {
      while ((high_length - low_length) / low_length > 1e-1f) {
        const float mid_length = (low_length + high_length) / 2.f;
        const PointCloud candidate = VoxelFilter(in_range, mid_length);
        if (candidate.size() >= min_num_points) {
          low_length = mid_length;
          result = candidate;
        } else {
          high_length = mid_length;
        }
      }
      return result;
    }
  }
  return result;
}

namespace {
constexpr float kMinDistance = 0.2f, kMaxDistance = 0.9f, kSliceHeight = 0.2f;

void AddValueToHistogram(float angle, float value, std::vector<float>* histogram) {
  while (angle > static_cast<float>(M_PI)) angle -= static_cast<float>(M_PI);
  while (angle < 0.f) angle += static_cast<float>(M_PI);
  const float zero_to_one = angle / static_cast<float>(M_PI);
  const int size = static_cast<int>(histogram->size());
  const int bucket = std::min(std::max(RoundToInt(size * zero_to_one - 0.5f), 0), size - 1);
  (*histogram)[bucket] += value;
}

V3f Centroid(const PointCloud& slice) {
  V3f sum{0.f, 0.f, 0.f};
  for (const Point3f& p : slice) { sum.x += p.x; sum.y += p.y; sum.z += p.z; }
  const float n = static_cast<float>(slice.size());
  return V3f{sum.x / n, sum.y / n, sum.z / n};
}

float Norm2(float x, float y) { return std::sqrt(x * x + y * y); }

PointCloud SortSlice(const PointCloud& slice) {
  struct Pair {
    float angle;
    Point3f point;
    bool operator<(const Pair& rhs) const { return angle < rhs.angle; }
  };
  const V3f c = Centroid(slice);
  std::vector<Pair> by_angle;
  by_angle.reserve(slice.size());
  for (const Point3f& p : slice) {
    const float dx = p.x - c.x, dy = p.y - c.y;
    if (Norm2(dx, dy) < kMinDistance) continue;
    by_angle.push_back(Pair{std::atan2(dy, dx), p});
  }
  std::sort(by_angle.begin(), by_angle.end());
  PointCloud result;
  for (const Pair& pair : by_angle) result.push_back(pair.point);
  return result;
}

void AddSlice(const PointCloud& slice, std::vector<float>* histogram) {
  if (slice.empty()) return;
  const V3f c = Centroid(slice);
  Point3f last = slice.front();
  for (const Point3f& p : slice) {
    const float dx = p.x - last.x, dy = p.y - last.y;
    const float ex = p.x - c.x, ey = p.y - c.y;
    const float distance = Norm2(dx, dy);
    const float direction_norm = Norm2(ex, ey);
    if (distance < kMinDistance || direction_norm < kMinDistance) continue;
    if (distance > kMaxDistance) {
      last = p;
      continue;
    }
    const float angle = std::atan2(dy, dx);
    // delta.normalized().dot(direction.normalized()): Eigen's normalized() divides by
    // sqrt(squaredNorm) when it is > 0.
    const float ndx = dx / distance, ndy = dy / distance;
    const float nex = ex / direction_norm, ney = ey / direction_norm;
    const float value = std::max(0.f, 1.f - std::abs(ndx * nex + ndy * ney));
    AddValueToHistogram(angle, value, histogram);
  }
}
}  // namespace

std::vector<float> ComputeHistogram(const PointCloud& cloud, int histogram_size) {
  std::vector<float> histogram(histogram_size, 0.f);
  std::map<int, PointCloud> slices;
  for (const Point3f& p : cloud) slices[RoundToInt(p.z / kSliceHeight)].push_back(p);
  for (const auto& slice : slices) AddSlice(SortSlice(slice.second), &histogram);
  return histogram;
}

}  // namespace oracle
