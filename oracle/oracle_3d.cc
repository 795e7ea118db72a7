// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
#include "oracle_3d.h"

#include <atomic>
#include <thread>

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>

namespace oracle {

// ------------------------------------------------------------- transforms ---
Qf AngleAxisVectorToRotationQuaternion(const V3f& aa) {
  float scale = 0.5f;
  float w = 1.f;
  constexpr double kCutoffAngle = 1e-8;
  const float squared_norm = (aa.x * aa.x + aa.y * aa.y) + aa.z * aa.z;
  if (squared_norm > kCutoffAngle) {
    const float norm = std::sqrt(squared_norm);
    scale = static_cast<float>(std::sin(norm / 2.) / norm);
    w = static_cast<float>(std::cos(norm / 2.));
  }
  return {w, scale * aa.x, scale * aa.y, scale * aa.z};
}

float GetAngle(const Rigid3f& t) {
  const float vec_norm = std::sqrt((t.q.x * t.q.x + t.q.y * t.q.y) + t.q.z * t.q.z);
  return 2.f * std::atan2(vec_norm, std::abs(t.q.w));
}

float GetYaw(const Qf& q) {
  const V3f direction = Rotate(q, V3f{1.f, 0.f, 0.f});
  return std::atan2(direction.y, direction.x);
}

namespace {
Rigid3f CastPose(const Pose3d& p) {
  Rigid3f r;
  r.t = {static_cast<float>(p.t[0]), static_cast<float>(p.t[1]), static_cast<float>(p.t[2])};
  r.q = {static_cast<float>(p.q.w), static_cast<float>(p.q.x), static_cast<float>(p.q.y),
         static_cast<float>(p.q.z)};
  return r;
}
Pose3d CastPose(const Rigid3f& r) {
  Pose3d p;
  p.t[0] = r.t.x; p.t[1] = r.t.y; p.t[2] = r.t.z;
  p.q = {r.q.w, r.q.x, r.q.y, r.q.z};
  return p;
}
float Norm3(const V3f& v) { return std::sqrt((v.x * v.x + v.y * v.y) + v.z * v.z); }
}  // namespace

// ------------------------------------------------------------- hybrid grid ---
HybridGridView::HybridGridView(float resolution, const Voxel* voxels, int64_t n)
    : resolution_(resolution), grid_size_(128) {
  if (n <= 0) return;
  Cell3i lo{voxels[0].x, voxels[0].y, voxels[0].z}, hi = lo;
  for (int64_t i = 0; i < n; ++i) {
    lo.x = std::min(lo.x, voxels[i].x); hi.x = std::max(hi.x, voxels[i].x);
    lo.y = std::min(lo.y, voxels[i].y); hi.y = std::max(hi.y, voxels[i].y);
    lo.z = std::min(lo.z, voxels[i].z); hi.z = std::max(hi.z, voxels[i].z);
  }
  // DynamicGrid::Grow (hybrid_grid.h:381-398): doubles until every written
  // index satisfies -size/2 <= i < size/2.
  auto fits = [&](int gs) {
    const int h = gs / 2;
    return lo.x >= -h && lo.y >= -h && lo.z >= -h && hi.x < h && hi.y < h && hi.z < h;
  };
  while (!fits(grid_size_)) grid_size_ *= 2;
  cells_.Reset(lo, hi);
  for (int64_t i = 0; i < n; ++i)
    *cells_.mutable_value(voxels[i].x, voxels[i].y, voxels[i].z) = voxels[i].value;
}

// ------------------------------------------------------------ real-time 3D ---
float RealTimeMatch3D(const HybridGridView& grid, const Pose3d& initial, const PointCloud3& cloud,
                      const double linear_window, const double angular_window, const double tw,
                      const double rw, Pose3d* pose_estimate, int64_t* num_candidates,
                      const int num_threads) {
  const float resolution = grid.resolution();
  // GenerateExhaustiveSearchTransforms (:55-95).
  const int linear_window_size = RoundToInt(linear_window / resolution);
  float max_scan_range = 3.f * resolution;
  for (const V3f& p : cloud) max_scan_range = std::max(Norm3(p), max_scan_range);
  const float kSafetyMargin = 1.f - 1e-3f;
  const float angular_step_size =
      kSafetyMargin * std::acos(1.f - (resolution * (resolution * 1.f)) /
                                          (2.f * (max_scan_range * (max_scan_range * 1.f))));
  const int angular_window_size = RoundToInt(angular_window / angular_step_size);
  const Rigid3f init = CastPose(initial);
  // The reference's loop is sequential (z outermost); slices of z are independent, so they may
  // run on several threads and are joined in z order with the loop's own rule (strict '>': the
  // first maximum wins).  num_threads = 1 is the loop as written.
  struct SliceBest { float score = -1.f; Pose3d pose; int64_t count = 0; };
  const int num_slices = 2 * linear_window_size + 1;
  std::vector<SliceBest> slices(num_slices);
  auto run_slice = [&](int z) {
    SliceBest& out = slices[z + linear_window_size];
    for (int y = -linear_window_size; y <= linear_window_size; ++y)
      for (int x = -linear_window_size; x <= linear_window_size; ++x)
        for (int rz = -angular_window_size; rz <= angular_window_size; ++rz)
          for (int ry = -angular_window_size; ry <= angular_window_size; ++ry)
            for (int rx = -angular_window_size; rx <= angular_window_size; ++rx) {
              const V3f angle_axis{rx * angular_step_size, ry * angular_step_size,
                                   rz * angular_step_size};
              Rigid3f transform;
              transform.t = {x * resolution, y * resolution, z * resolution};
              transform.q = AngleAxisVectorToRotationQuaternion(angle_axis);
              const Rigid3f candidate = MulSse(init, transform);
              // ScoreCandidate (:97-114).
              float score = 0.f;
              for (const V3f& p : cloud)
                score += grid.GetProbability(grid.GetCellIndex(Apply(candidate, p)));
              score /= static_cast<float>(cloud.size());
              const float angle = GetAngle(transform);
              const double t = Norm3(transform.t) * tw + angle * rw;
              score *= std::exp(-(t * (t * 1.)));
              ++out.count;
              if (score > out.score) {
                out.score = score;
                out.pose = CastPose(candidate);
              }
            }
  };
  if (num_threads <= 1) {
    for (int z = -linear_window_size; z <= linear_window_size; ++z) run_slice(z);
  } else {
    std::atomic<int> next{-linear_window_size};
    std::vector<std::thread> pool;
    for (int t = 0; t < num_threads; ++t)
      pool.emplace_back([&] {
        for (int z = next.fetch_add(1); z <= linear_window_size; z = next.fetch_add(1)) run_slice(z);
      });
    for (std::thread& th : pool) th.join();
  }
  float best_score = -1.f;
  int64_t count = 0;
  for (const SliceBest& sl : slices) {
    count += sl.count;
    if (sl.score > best_score) {
      best_score = sl.score;
      *pose_estimate = sl.pose;
    }
  }
  if (num_candidates) *num_candidates = count;
  return best_score;
}

// ------------------------------------------------------ precomputation 3D ---
PrecomputationGrid3D ConvertToPrecomputationGrid(const HybridGridView& grid) {
  PrecomputationGrid3D result;
  if (grid.cells().empty()) return result;
  result.Reset(grid.cells().lo(), grid.cells().hi());
  grid.cells().ForEachNonZero([&](const Cell3i& c, uint16_t v) {
    const int cell_value = RoundToInt((ValueToProbability(v) - kMinProbability) *
                                      (255.f / (kMaxProbability - kMinProbability)));
    assert(cell_value >= 0 && cell_value <= 255);
    *result.mutable_value(c.x, c.y, c.z) = static_cast<uint8_t>(cell_value);
  });
  return result;
}

PrecomputationGrid3D PrecomputeGrid(const PrecomputationGrid3D& grid, const bool half_resolution,
                                    const int shift) {
  PrecomputationGrid3D result;
  if (grid.empty()) return result;
  Cell3i lo = grid.lo(), hi = grid.hi();
  lo.x -= shift; lo.y -= shift; lo.z -= shift;
  if (half_resolution) {
    lo = {lo.x >> 1, lo.y >> 1, lo.z >> 1};
    hi = {hi.x >> 1, hi.y >> 1, hi.z >> 1};
  }
  result.Reset(lo, hi);
  grid.ForEachNonZero([&](const Cell3i& c, uint8_t v) {
    for (int i = 0; i != 8; ++i) {
      Cell3i t{c.x - shift * (i & 1), c.y - shift * ((i >> 1) & 1), c.z - shift * ((i >> 2) & 1)};
      if (half_resolution) t = {t.x >> 1, t.y >> 1, t.z >> 1};
      uint8_t* out = result.mutable_value(t.x, t.y, t.z);
      *out = std::max(v, *out);
    }
  });
  return result;
}

// ---------------------------------------------------- rotational matcher ---
std::vector<float> RotateHistogram(const std::vector<float>& histogram, const float angle) {
  const int size = static_cast<int>(histogram.size());
  if (size == 0) return histogram;
  // f32 * Index -> f32, then / M_PI in f64, stored as f32.
  const float rotate_by_buckets =
      static_cast<float>(static_cast<double>(-angle * static_cast<float>(size)) / M_PI);
  int full_buckets = RoundToInt(rotate_by_buckets - 0.5f);
  const float fraction = rotate_by_buckets - full_buckets;
  while (full_buckets < 0) full_buckets += size;
  std::vector<float> out(size);
  for (int i = 0; i != size; ++i) {
    const float h0 = histogram[(i + full_buckets) % size];
    const float h1 = histogram[(i + 1 + full_buckets) % size];
    out[i] = fraction * h1 + (1.f - fraction) * h0;
  }
  return out;
}

namespace {
// Sequential f32 reductions (Eigen's packet order is unpinned, see header).
float Dot(const std::vector<float>& a, const std::vector<float>& b) {
  float s = 0.f;
  for (size_t i = 0; i != a.size(); ++i) s += a[i] * b[i];
  return s;
}
float MatchHistograms(const std::vector<float>& submap, const std::vector<float>& scan) {
  const float scan_norm = std::sqrt(Dot(scan, scan));
  const float submap_norm = std::sqrt(Dot(submap, submap));
  const float normalization = scan_norm * submap_norm;
  if (normalization < 1e-3f) return 1.f;
  return Dot(submap, scan) / normalization;
}
}  // namespace

std::vector<float> RotationalMatch(const std::vector<float>& submap_histogram,
                                   const std::vector<float>& scan_histogram,
                                   const float initial_angle, const std::vector<float>& angles) {
  std::vector<float> result;
  result.reserve(angles.size());
  for (const float angle : angles)
    result.push_back(
        MatchHistograms(submap_histogram, RotateHistogram(scan_histogram, initial_angle + angle)));
  return result;
}

// ------------------------------------------------------------------ fast 3D ---
FastCorrelativeScanMatcher3D::FastCorrelativeScanMatcher3D(
    std::shared_ptr<HybridGridView> grid, std::shared_ptr<HybridGridView> low_resolution_grid,
    std::vector<float> histogram, const Fast3DOptions& options)
    : options_(options), resolution_(grid->resolution()), width_in_voxels_(grid->grid_size()),
      grid_(std::move(grid)), low_grid_(std::move(low_resolution_grid)),
      histogram_(std::move(histogram)) {
  // PrecomputationGridStack3D (:57-77).
  assert(options.branch_and_bound_depth >= 1 && options.full_resolution_depth >= 1);
  stack_.push_back(ConvertToPrecomputationGrid(*grid_));
  int last_width = 1;
  for (int depth = 1; depth != options.branch_and_bound_depth; ++depth) {
    const bool half_resolution = depth >= options.full_resolution_depth;
    const int next_width = 1 << depth;
    const int full_voxels_per_high_resolution_voxel =
        1 << std::max(0, depth - options.full_resolution_depth);
    const int shift = (next_width - last_width + (full_voxels_per_high_resolution_voxel - 1)) /
                      full_voxels_per_high_resolution_voxel;
    stack_.push_back(PrecomputeGrid(stack_.back(), half_resolution, shift));
    last_width = next_width;
  }
}

float FastCorrelativeScanMatcher3D::LowResolutionScore(const Search& sp,
                                                       const Rigid3f& pose) const {
  float score = 0.f;
  for (const V3f& p : *sp.low_resolution_cloud)
    score += low_grid_->GetProbability(low_grid_->GetCellIndex(Apply(pose, p)));
  return score / static_cast<float>(sp.low_resolution_cloud->size());
}

bool FastCorrelativeScanMatcher3D::Match(const Pose3d& global_node_pose,
                                         const Pose3d& global_submap_pose,
                                         const NodeData3D& data, const float min_score,
                                         Result3D* result, Stats3D* stats) const {
  const Search sp{RoundToInt(options_.linear_xy_search_window / resolution_),
                  RoundToInt(options_.linear_z_search_window / resolution_),
                  options_.angular_search_window, &data.low_resolution_point_cloud};
  return MatchWithSearchParameters(sp, CastPose(global_node_pose), CastPose(global_submap_pose),
                                   data, min_score, result, stats);
}

bool FastCorrelativeScanMatcher3D::MatchFullSubmap(const Qd& node_rotation,
                                                   const Qd& submap_rotation,
                                                   const NodeData3D& data, const float min_score,
                                                   Result3D* result, Stats3D* stats) const {
  float max_point_distance = 0.f;
  for (const V3f& p : data.high_resolution_point_cloud)
    max_point_distance = std::max(max_point_distance, Norm3(p));
  const int linear_window_size =
      (width_in_voxels_ + 1) / 2 + RoundToInt(max_point_distance / resolution_ + 0.5f);
  const Search sp{linear_window_size, linear_window_size, M_PI,
                  &data.low_resolution_point_cloud};
  Rigid3f node, submap;
  node.q = {static_cast<float>(node_rotation.w), static_cast<float>(node_rotation.x),
            static_cast<float>(node_rotation.y), static_cast<float>(node_rotation.z)};
  submap.q = {static_cast<float>(submap_rotation.w), static_cast<float>(submap_rotation.x),
              static_cast<float>(submap_rotation.y), static_cast<float>(submap_rotation.z)};
  return MatchWithSearchParameters(sp, node, submap, data, min_score, result, stats);
}

FastCorrelativeScanMatcher3D::DiscreteScan3D FastCorrelativeScanMatcher3D::DiscretizeScan(
    const Search& sp, const PointCloud3& cloud, const Rigid3f& pose,
    const float rotational_score) const {
  DiscreteScan3D scan;
  scan.pose = pose;
  scan.rotational_score = rotational_score;
  std::vector<Cell3i> full;
  full.reserve(cloud.size());
  for (const V3f& p : cloud) {
    const V3f t = Apply(pose, p);
    full.push_back({RoundToInt(t.x / resolution_), RoundToInt(t.y / resolution_),
                    RoundToInt(t.z / resolution_)});
  }
  const int full_resolution_depth =
      std::min(options_.full_resolution_depth, options_.branch_and_bound_depth);
  for (int i = 0; i != full_resolution_depth; ++i) scan.cell_indices_per_depth.push_back(full);
  const int low_resolution_depth = options_.branch_and_bound_depth - full_resolution_depth;
  const Cell3i start{-sp.linear_xy_window_size, -sp.linear_xy_window_size,
                     -sp.linear_z_window_size};
  for (int i = 0; i != low_resolution_depth; ++i) {
    const int e = i + 1;
    const Cell3i low_start{start.x >> e, start.y >> e, start.z >> e};
    scan.cell_indices_per_depth.emplace_back();
    for (const Cell3i& c : full) {
      scan.cell_indices_per_depth.back().push_back({((c.x + start.x) >> e) - low_start.x,
                                                    ((c.y + start.y) >> e) - low_start.y,
                                                    ((c.z + start.z) >> e) - low_start.z});
    }
  }
  return scan;
}

std::vector<FastCorrelativeScanMatcher3D::DiscreteScan3D>
FastCorrelativeScanMatcher3D::GenerateDiscreteScans(const Search& sp, const NodeData3D& data,
                                                    const Rigid3f& node,
                                                    const Rigid3f& submap) const {
  const PointCloud3& cloud = data.high_resolution_point_cloud;
  std::vector<DiscreteScan3D> result;
  float max_scan_range = 3.f * resolution_;
  for (const V3f& p : cloud) max_scan_range = std::max(Norm3(p), max_scan_range);
  const float kSafetyMargin = 1.f - 1e-2f;
  const float angular_step_size =
      kSafetyMargin * std::acos(1.f - (resolution_ * (resolution_ * 1.f)) /
                                          (2.f * (max_scan_range * (max_scan_range * 1.f))));
  const int angular_window_size = RoundToInt(sp.angular_search_window / angular_step_size);
  std::vector<float> angles;
  for (int rz = -angular_window_size; rz <= angular_window_size; ++rz)
    angles.push_back(rz * angular_step_size);
  const Rigid3f node_to_submap = MulSse(InverseRigid(submap), node);
  // gravity_alignment.inverse().cast<float>()  (Quaterniond::inverse: conj / squaredNorm)
  const Qd& g = data.gravity_alignment;
  const double n2 = (g.x * g.x + g.z * g.z) + (g.y * g.y + g.w * g.w);
  const Qf g_inv{static_cast<float>(g.w / n2), static_cast<float>(-g.x / n2),
                 static_cast<float>(-g.y / n2), static_cast<float>(-g.z / n2)};
  const std::vector<float> scores =
      RotationalMatch(histogram_, data.rotational_scan_matcher_histogram,
                      GetYaw(QuatMulSse(node_to_submap.q, g_inv)), angles);
  for (size_t i = 0; i != angles.size(); ++i) {
    if (scores[i] < options_.min_rotational_score) continue;
    const V3f angle_axis{0.f, 0.f, angles[i]};
    Rigid3f pose;
    pose.t = node_to_submap.t;
    pose.q = QuatMulSse(QuatMulSse(QuatInverseSse(submap.q),
                                   AngleAxisVectorToRotationQuaternion(angle_axis)),
                        node.q);
    result.push_back(DiscretizeScan(sp, cloud, pose, scores[i]));
  }
  return result;
}

static FILE* g_node_dump = nullptr;      // see MatchWithSearchParameters

void FastCorrelativeScanMatcher3D::ScoreCandidates(const int depth,
                                                   const std::vector<DiscreteScan3D>& scans,
                                                   std::vector<Candidate3D>* candidates,
                                                   Stats3D* stats) const {
  const int e = std::max(0, depth - options_.full_resolution_depth + 1);
  const PrecomputationGrid3D& g = stack_[depth];
  for (Candidate3D& c : *candidates) {
    int sum = 0;
    const std::vector<Cell3i>& cells = scans[c.scan_index].cell_indices_per_depth[depth];
    const Cell3i offset{c.offset.x >> e, c.offset.y >> e, c.offset.z >> e};
    for (const Cell3i& p : cells) sum += g.value(p.x + offset.x, p.y + offset.y, p.z + offset.z);
    c.score = ToProbability3D(sum / static_cast<float>(cells.size()));
  }
  if (stats) stats->candidates_scored += candidates->size();
  std::sort(candidates->begin(), candidates->end(), std::greater<Candidate3D>());
}

Rigid3f FastCorrelativeScanMatcher3D::GetPoseFromCandidate(
    const std::vector<DiscreteScan3D>& scans, const Candidate3D& c) const {
  Rigid3f translation;
  translation.t = {resolution_ * static_cast<float>(c.offset.x),
                   resolution_ * static_cast<float>(c.offset.y),
                   resolution_ * static_cast<float>(c.offset.z)};
  return MulSse(translation, scans[c.scan_index].pose);
}

FastCorrelativeScanMatcher3D::Candidate3D FastCorrelativeScanMatcher3D::BranchAndBound(
    const Search& sp, const std::vector<DiscreteScan3D>& scans,
    const std::vector<Candidate3D>& candidates, const int candidate_depth, float min_score,
    Stats3D* stats) const {
  const Candidate3D unsuccessful{0, {0, 0, 0}, -std::numeric_limits<float>::infinity(), 0.f};
  if (candidate_depth == 0) {
    for (const Candidate3D& c : candidates) {
      if (c.score <= min_score) return unsuccessful;
      const float low = LowResolutionScore(sp, GetPoseFromCandidate(scans, c));
      if (low >= options_.min_low_resolution_score) {
        Candidate3D best = c;
        best.low_resolution_score = low;
        return best;
      }
    }
    return unsuccessful;
  }
  Candidate3D best = unsuccessful;
  best.score = min_score;
  for (const Candidate3D& c : candidates) {
    if (c.score <= min_score) break;
    if (stats) ++stats->nodes_expanded;
    if (g_node_dump) {        // ORC_DUMP_NODES (design studies): scan, depth, offset of the expansion
      const int32_t rec[5] = {c.scan_index, candidate_depth, c.offset.x, c.offset.y, c.offset.z};
      fwrite(rec, sizeof rec, 1, g_node_dump);
    }
    std::vector<Candidate3D> children;
    const int half = 1 << (candidate_depth - 1);
    for (int z : {0, half}) {
      if (c.offset.z + z > sp.linear_z_window_size) break;
      for (int y : {0, half}) {
        if (c.offset.y + y > sp.linear_xy_window_size) break;
        for (int x : {0, half}) {
          if (c.offset.x + x > sp.linear_xy_window_size) break;
          children.push_back({c.scan_index, {c.offset.x + x, c.offset.y + y, c.offset.z + z},
                              -std::numeric_limits<float>::infinity(), 0.f});
        }
      }
    }
    ScoreCandidates(candidate_depth - 1, scans, &children, stats);
    best = std::max(best, BranchAndBound(sp, scans, children, candidate_depth - 1, best.score,
                                         stats));
  }
  return best;
}

bool FastCorrelativeScanMatcher3D::MatchWithSearchParameters(
    const Search& sp, const Rigid3f& node, const Rigid3f& submap, const NodeData3D& data,
    const float min_score, Result3D* result, Stats3D* stats) const {
  const std::vector<DiscreteScan3D> scans = GenerateDiscreteScans(sp, data, node, submap);
  // GenerateLowestResolutionCandidates (:297-330).
  const int max_depth = depth() - 1;
  const int step = 1 << max_depth;
  std::vector<Candidate3D> lowest;
  for (int s = 0; s != static_cast<int>(scans.size()); ++s)
    for (int z = -sp.linear_z_window_size; z <= sp.linear_z_window_size; z += step)
      for (int y = -sp.linear_xy_window_size; y <= sp.linear_xy_window_size; y += step)
        for (int x = -sp.linear_xy_window_size; x <= sp.linear_xy_window_size; x += step)
          lowest.push_back({s, {x, y, z}, -std::numeric_limits<float>::infinity(), 0.f});
  if (stats) {
    stats->num_scans = scans.size();
    stats->coarse_candidates = lowest.size();
  }
  ScoreCandidates(max_depth, scans, &lowest, stats);
  // ORC_DUMP_NODES=<file> (design studies, tools/prototypes): the discrete scans (cells per depth)
  // and every expanded node of this search, binary.  Not thread-safe: one search at a time.
  if (const char* path = getenv("ORC_DUMP_NODES")) {
    g_node_dump = fopen(path, "wb");
    if (g_node_dump) {
      const int32_t head[4] = {static_cast<int32_t>(scans.size()), depth(),
                               static_cast<int32_t>(scans.empty() ? 0 : scans[0].cell_indices_per_depth[0].size()),
                               options_.full_resolution_depth};
      fwrite(head, sizeof head, 1, g_node_dump);
      for (const DiscreteScan3D& scan : scans)
        for (int d = 0; d < depth(); ++d)
          fwrite(scan.cell_indices_per_depth[d].data(), sizeof(Cell3i),
                 scan.cell_indices_per_depth[d].size(), g_node_dump);
    }
  }
  const Candidate3D best = BranchAndBound(sp, scans, lowest, max_depth, min_score, stats);
  if (g_node_dump) {
    fclose(g_node_dump);
    g_node_dump = nullptr;
  }
  if (best.score > min_score) {
    result->score = best.score;
    result->pose_estimate = CastPose(GetPoseFromCandidate(scans, best));
    result->rotational_score = scans[best.scan_index].rotational_score;
    result->low_resolution_score = best.low_resolution_score;
    return true;
  }
  return false;
}

}  // namespace oracle
