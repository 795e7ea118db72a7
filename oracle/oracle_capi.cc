// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
// Flat C entry points so tests/ and bench.py's cpu_baseline leg can drive the
// oracle through ctypes.
#include <chrono>
#include <cstring>
#include <memory>
#include <thread>

#include "oracle_2d.h"
#include "oracle_3d.h"
#include "oracle_ceres_2d.h"
#include "oracle_ceres_3d.h"
#include "oracle_filters.h"

#include <map>
#include <tuple>

using namespace oracle;

namespace {
ProbabilityGridView MakeView(const uint16_t* cells, int nx, int ny, double res, double max_x,
                             double max_y) {
  ProbabilityGridView g;
  g.limits = MapLimits{res, max_x, max_y, nx, ny};
  g.cells = cells;
  return g;
}
PointCloud MakeCloud(const float* xyz, int n) {
  PointCloud c(n);
  for (int i = 0; i != n; ++i) c[i] = Point3f{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
  return c;
}
struct Fast2D {
  std::vector<uint16_t> cells;
  std::unique_ptr<FastCorrelativeScanMatcher2D> matcher;
};
}  // namespace

extern "C" {

// ---- tables / scalar helpers (pins for probability_values_test etc.) ----
void orc_value_tables(float* value_to_probability_65536, float* value_to_cost_65536,
                      float* grid_cost_65536) {
  std::memcpy(value_to_probability_65536, ValueToProbabilityTable().data(), 65536 * 4);
  std::memcpy(value_to_cost_65536, ValueToCorrespondenceCostTable().data(), 65536 * 4);
  std::memcpy(grid_cost_65536, GridCorrespondenceCostTable().data(), 65536 * 4);
}
int orc_probability_to_value(float p) { return ProbabilityToValue(p); }
int orc_correspondence_cost_to_value(float c) { return CorrespondenceCostToValue(c); }

// ---- search parameters ----
void orc_search_parameters(double lin, double ang, const float* xyz, int n, double res,
                           int* num_angular, double* step, int* num_scans, int* num_linear) {
  const SearchParameters sp(lin, ang, MakeCloud(xyz, n), res);
  *num_angular = sp.num_angular_perturbations;
  *step = sp.angular_perturbation_step_size;
  *num_scans = sp.num_scans;
  *num_linear = sp.linear_bounds.empty() ? 0 : sp.linear_bounds[0].max_x;
}

// GenerateRotatedScans with the testing ctor SearchParameters(nl, na, step, res).
void orc_generate_rotated_scans(const float* xyz, int n, int na, double step,
                                float* out_xyz /* (2na+1)*n*3 */) {
  const SearchParameters sp(0, na, step, 0.);
  const auto scans = GenerateRotatedScans(MakeCloud(xyz, n), sp);
  for (size_t s = 0; s != scans.size(); ++s)
    for (int i = 0; i != n; ++i) {
      out_xyz[(s * n + i) * 3 + 0] = scans[s][i].x;
      out_xyz[(s * n + i) * 3 + 1] = scans[s][i].y;
      out_xyz[(s * n + i) * 3 + 2] = scans[s][i].z;
    }
}

// Rotate (initial theta), generate `na` perturbations at `step`, discretise.
void orc_discretize_scans(const float* xyz, int n, double init_theta, int na, double step,
                          double res, double max_x, double max_y, int nx, int ny, float tx,
                          float ty, int* out_xy /* (2na+1)*n*2 */) {
  const SearchParameters sp(0, na, step, res);
  const PointCloud rotated = RotateCloudYaw(MakeCloud(xyz, n), static_cast<float>(init_theta));
  const auto scans = GenerateRotatedScans(rotated, sp);
  const auto d = DiscretizeScans(MapLimits{res, max_x, max_y, nx, ny}, scans, tx, ty);
  for (size_t s = 0; s != d.size(); ++s)
    for (int i = 0; i != n; ++i) {
      out_xy[(s * n + i) * 2 + 0] = d[s][i].x;
      out_xy[(s * n + i) * 2 + 1] = d[s][i].y;
    }
}

void orc_candidate2d(int nl, int na, double step, double res, int scan_index, int x_off,
                     int y_off, double* out_xyo) {
  const SearchParameters sp(nl, na, step, res);
  const Candidate2D c(scan_index, x_off, y_off, sp);
  out_xyo[0] = c.x; out_xyo[1] = c.y; out_xyo[2] = c.orientation;
}

float orc_grid_probability(const uint16_t* cells, int nx, int ny, int ix, int iy) {
  return MakeView(cells, nx, ny, 1., 0., 0.).GetProbability(Cell2i{ix, iy});
}

// ---- real-time 2D ----
double orc_rt2d_match(const uint16_t* cells, int nx, int ny, double res, double max_x,
                      double max_y, const double* init_xyt, const float* xyz, int n,
                      double lin, double ang, double tw, double rw, double* pose_xyt,
                      int64_t* num_candidates, float* all_scores, int all_scores_capacity) {
  const ProbabilityGridView g = MakeView(cells, nx, ny, res, max_x, max_y);
  Pose2d pose;
  MatchStats st;
  std::vector<float> scores;
  const double s = RealTimeMatch2D(g, Pose2d{init_xyt[0], init_xyt[1], init_xyt[2]},
                                   MakeCloud(xyz, n), lin, ang, tw, rw, &pose, &st,
                                   all_scores ? &scores : nullptr);
  pose_xyt[0] = pose.x; pose_xyt[1] = pose.y; pose_xyt[2] = pose.theta;
  if (num_candidates) *num_candidates = st.candidates_scored;
  if (all_scores) {
    const size_t m = std::min<size_t>(scores.size(), all_scores_capacity);
    std::memcpy(all_scores, scores.data(), m * 4);
  }
  return s;
}

// TSDF branch (a8'): tsd / weight planes + truncation distance + max weight.
double orc_rt2d_match_tsdf(const uint16_t* tsd_cells, const uint16_t* weight_cells, int nx,
                           int ny, double res, double max_x, double max_y,
                           float truncation_distance, float max_weight, const double* init_xyt,
                           const float* xyz, int n, double lin, double ang, double tw, double rw,
                           double* pose_xyt, int64_t* num_candidates, float* all_scores,
                           int all_scores_capacity) {
  const TsdfView g(MapLimits{res, max_x, max_y, nx, ny}, tsd_cells, weight_cells,
                   truncation_distance, max_weight);
  Pose2d pose;
  MatchStats st;
  std::vector<float> scores;
  const double s = RealTimeMatch2DTsdf(g, Pose2d{init_xyt[0], init_xyt[1], init_xyt[2]},
                                       MakeCloud(xyz, n), lin, ang, tw, rw, &pose, &st,
                                       all_scores ? &scores : nullptr);
  pose_xyt[0] = pose.x; pose_xyt[1] = pose.y; pose_xyt[2] = pose.theta;
  if (num_candidates) *num_candidates = st.candidates_scored;
  if (all_scores) {
    const size_t m = std::min<size_t>(scores.size(), all_scores_capacity);
    std::memcpy(all_scores, scores.data(), m * 4);
  }
  return s;
}

// TSDValueConverter (mapping/internal/2d/tsd_value_converter.h:39-67, .cc:22-33).
// kind 0: tsd in [-bound, bound]; kind 1: weight in [0, bound].
int orc_tsd_float_to_value(int kind, float bound, float x) {
  const float lo = kind == 0 ? -bound : 0.f, hi = bound;
  const float resolution = 32766.f / (hi - lo);
  const float clamped = x > hi ? hi : (x < lo ? lo : x);     // common::Clamp (math.h:31-42)
  return static_cast<int>(std::lround((clamped - lo) * resolution)) + 1;
}
float orc_tsd_value_to_float(int kind, float bound, int raw) {
  const float lo = kind == 0 ? -bound : 0.f, hi = bound;
  const TsdfView v(MapLimits{1., 0., 0., 1, 1}, nullptr, nullptr, 1.f, 1.f);
  (void)v;
  const uint16_t value = static_cast<uint16_t>(raw) & static_cast<uint16_t>(~kUpdateMarker);
  if (value == 0) return lo;                                  // unknown -> lower bound
  const float kScale = (hi - lo) / 32766.f;
  return value * kScale + (lo - kScale);
}

// ---- precomputation grid / fast 2D ----
void* orc_fast2d_create(const uint16_t* cells, int nx, int ny, double res, double max_x,
                        double max_y, int depth, double lin, double ang) {
  auto* f = new Fast2D;
  f->cells.assign(cells, cells + static_cast<size_t>(nx) * ny);
  f->matcher.reset(new FastCorrelativeScanMatcher2D(
      MakeView(f->cells.data(), nx, ny, res, max_x, max_y), depth, lin, ang));
  return f;
}
void orc_fast2d_destroy(void* h) { delete static_cast<Fast2D*>(h); }

void orc_fast2d_level_dims(void* h, int level, int* wx, int* wy) {
  const auto& g = static_cast<Fast2D*>(h)->matcher->level(level);
  *wx = g.wide_x(); *wy = g.wide_y();
}
void orc_fast2d_level_cells(void* h, int level, uint8_t* out) {
  const auto& g = static_cast<Fast2D*>(h)->matcher->level(level);
  std::memcpy(out, g.cells().data(), g.cells().size());
}
// PrecomputationGrid2D of an arbitrary width (reference test uses widths 3, 200).
void orc_precompute2d(const uint16_t* cells, int nx, int ny, int width, uint8_t* out) {
  const PrecomputationGrid2D g(MakeView(cells, nx, ny, 0.05, 0., 0.), width);
  std::memcpy(out, g.cells().data(), g.cells().size());
}

// The same over a Grid2D with other correspondence-cost bounds (a TSDF2D's tsd plane:
// min = -truncation_distance, max = truncation_distance).
void orc_precompute2d_range(const uint16_t* cells, int nx, int ny, int width, float min_cc,
                            float max_cc, uint8_t* out) {
  ProbabilityGridView view = MakeView(cells, nx, ny, 0.05, 0., 0.);
  view.min_correspondence_cost = min_cc;
  view.max_correspondence_cost = max_cc;
  const PrecomputationGrid2D g(view, width);
  std::memcpy(out, g.cells().data(), g.cells().size());
}

int orc_fast2d_match(void* h, const double* init_xyt, const float* xyz, int n,
                     int full_submap, float min_score, float* score, double* pose_xyt,
                     int64_t* stats4 /* candidates, scans, coarse, nodes */) {
  const auto& m = *static_cast<Fast2D*>(h)->matcher;
  Pose2d pose{0, 0, 0};
  MatchStats st;
  float sc = 0.f;
  const PointCloud cloud = MakeCloud(xyz, n);
  const bool ok = full_submap
      ? m.MatchFullSubmap(cloud, min_score, &sc, &pose, &st)
      : m.Match(Pose2d{init_xyt[0], init_xyt[1], init_xyt[2]}, cloud, min_score, &sc, &pose, &st);
  *score = sc;
  pose_xyt[0] = pose.x; pose_xyt[1] = pose.y; pose_xyt[2] = pose.theta;
  if (stats4) {
    stats4[0] = st.candidates_scored; stats4[1] = st.num_scans;
    stats4[2] = st.coarse_candidates; stats4[3] = st.nodes_expanded;
  }
  return ok ? 1 : 0;
}

// Prepared search: number of scans, discrete scans, shrunk bounds, coarse sums.
int orc_fast2d_prepare(void* h, const double* init_xyt, const float* xyz, int n,
                       int full_submap, int* num_scans, double* step, int* out_xy,
                       int64_t out_xy_capacity /* ints */, int* out_bounds,
                       int64_t bounds_capacity /* ints */, int* out_sums,
                       int64_t sums_capacity, int64_t* num_sums) {
  const auto& m = *static_cast<Fast2D*>(h)->matcher;
  Pose2d used;
  const auto p = m.Prepare(Pose2d{init_xyt[0], init_xyt[1], init_xyt[2]}, MakeCloud(xyz, n),
                           full_submap != 0, &used);
  *num_scans = static_cast<int>(p.discrete_scans.size());
  *step = p.angular_step;
  if (out_xy) {
    int64_t k = 0;
    for (const auto& s : p.discrete_scans)
      for (const Cell2i& c : s) {
        if (k + 2 > out_xy_capacity) return -1;
        out_xy[k++] = c.x; out_xy[k++] = c.y;
      }
  }
  if (out_bounds) {
    int64_t k = 0;
    for (const auto& b : p.bounds) {
      if (k + 4 > bounds_capacity) return -1;
      out_bounds[k++] = b.min_x; out_bounds[k++] = b.max_x;
      out_bounds[k++] = b.min_y; out_bounds[k++] = b.max_y;
    }
  }
  if (out_sums || num_sums) {
    const std::vector<int> sums = m.CoarseSums(p);
    if (num_sums) *num_sums = static_cast<int64_t>(sums.size());
    if (out_sums) {
      if (static_cast<int64_t>(sums.size()) > sums_capacity) return -1;
      std::memcpy(out_sums, sums.data(), sums.size() * sizeof(int));
    }
  }
  return 0;
}

// Batched full-submap matches over `num_threads` host threads (one submap per
// task) — the reference's thread-pool fan-out
// (constraints/constraint_builder_2d.cc:97-111), used as the CPU baseline.
void orc_fast2d_match_batch(void** handles, int num, const float* xyz, int n, float min_score,
                            int num_threads, int* found, float* scores, double* poses_xyt,
                            int64_t* candidates_total) {
  std::vector<int64_t> per(num, 0);
  auto work = [&](int t) {
    for (int i = t; i < num; i += num_threads) {
      int64_t st[4];
      const double init[3] = {0, 0, 0};
      found[i] = orc_fast2d_match(handles[i], init, xyz, n, 1, min_score, &scores[i],
                                  &poses_xyt[3 * i], st);
      per[i] = st[0];
    }
  };
  std::vector<std::thread> threads;
  for (int t = 0; t < num_threads; ++t) threads.emplace_back(work, t);
  for (auto& th : threads) th.join();
  int64_t total = 0;
  for (int64_t v : per) total += v;
  if (candidates_total) *candidates_total = total;
}

}  // extern "C"

// ============================================================== 3D ==========
namespace {
Pose3d MakePose3(const double* p7) {   // t xyz, q wxyz
  Pose3d p;
  p.t[0] = p7[0]; p.t[1] = p7[1]; p.t[2] = p7[2];
  p.q = {p7[3], p7[4], p7[5], p7[6]};
  return p;
}
void StorePose3(const Pose3d& p, double* out7) {
  out7[0] = p.t[0]; out7[1] = p.t[1]; out7[2] = p.t[2];
  out7[3] = p.q.w; out7[4] = p.q.x; out7[5] = p.q.y; out7[6] = p.q.z;
}
PointCloud3 MakeCloud3(const float* xyz, int n) {
  PointCloud3 c(n);
  for (int i = 0; i != n; ++i) c[i] = V3f{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
  return c;
}
struct Fast3D {
  std::unique_ptr<FastCorrelativeScanMatcher3D> matcher;
};
NodeData3D MakeNodeData(const double* gravity_wxyz, const float* hi, int nhi, const float* lo,
                        int nlo, const float* hist, int nh) {
  NodeData3D d;
  d.gravity_alignment = {gravity_wxyz[0], gravity_wxyz[1], gravity_wxyz[2], gravity_wxyz[3]};
  d.high_resolution_point_cloud = MakeCloud3(hi, nhi);
  d.low_resolution_point_cloud = MakeCloud3(lo, nlo);
  d.rotational_scan_matcher_histogram.assign(hist, hist + nh);
  return d;
}
}  // namespace

extern "C" {

int orc_grid3d_size(float resolution, const Voxel* voxels, int64_t n) {
  return HybridGridView(resolution, voxels, n).grid_size();
}

float orc_rt3d_match(float resolution, const Voxel* voxels, int64_t n, const double* init7,
                     const float* xyz, int npts, double lin, double ang, double tw, double rw,
                     double* pose7, int64_t* num_candidates) {
  const HybridGridView grid(resolution, voxels, n);
  Pose3d pose = MakePose3(init7);
  const float s = RealTimeMatch3D(grid, MakePose3(init7), MakeCloud3(xyz, npts), lin, ang, tw,
                                  rw, &pose, num_candidates);
  StorePose3(pose, pose7);
  return s;
}

// As orc_rt3d_match with the z slices of the window spread over host threads (joined in z order
// with the loop's first-maximum rule: the same result).
float orc_rt3d_match_mt(float resolution, const Voxel* voxels, int64_t n, const double* init7,
                        const float* xyz, int npts, double lin, double ang, double tw, double rw,
                        int num_threads, double* pose7, int64_t* num_candidates) {
  const HybridGridView grid(resolution, voxels, n);
  Pose3d pose = MakePose3(init7);
  const float s = RealTimeMatch3D(grid, MakePose3(init7), MakeCloud3(xyz, npts), lin, ang, tw,
                                  rw, &pose, num_candidates, num_threads);
  StorePose3(pose, pose7);
  return s;
}

void orc_rotational_match(const float* submap_hist, const float* scan_hist, int size,
                          float initial_angle, const float* angles, int n, float* out) {
  const std::vector<float> r =
      RotationalMatch(std::vector<float>(submap_hist, submap_hist + size),
                      std::vector<float>(scan_hist, scan_hist + size), initial_angle,
                      std::vector<float>(angles, angles + n));
  std::memcpy(out, r.data(), n * sizeof(float));
}

void* orc_fast3d_create(float resolution, const Voxel* voxels, int64_t n, float low_resolution,
                        const Voxel* low_voxels, int64_t nlow, const float* hist, int nh,
                        int depth, int full_resolution_depth, double min_rotational_score,
                        double min_low_resolution_score, double lin_xy, double lin_z,
                        double ang) {
  auto* f = new Fast3D;
  auto grid = std::make_shared<HybridGridView>(resolution, voxels, n);
  auto low = std::make_shared<HybridGridView>(low_resolution, low_voxels, nlow);
  const Fast3DOptions o{depth, full_resolution_depth, min_rotational_score,
                        min_low_resolution_score, lin_xy, lin_z, ang};
  f->matcher.reset(new FastCorrelativeScanMatcher3D(grid, low,
                                                    std::vector<float>(hist, hist + nh), o));
  return f;
}
void orc_fast3d_destroy(void* h) { delete static_cast<Fast3D*>(h); }

int64_t orc_fast3d_level_count(void* h, int depth) {
  int64_t c = 0;
  static_cast<Fast3D*>(h)->matcher->level(depth).ForEachNonZero(
      [&](const Cell3i&, uint8_t) { ++c; });
  return c;
}
void orc_fast3d_level_voxels(void* h, int depth, int* out_xyzv) {
  int64_t k = 0;
  static_cast<Fast3D*>(h)->matcher->level(depth).ForEachNonZero([&](const Cell3i& c, uint8_t v) {
    out_xyzv[4 * k] = c.x; out_xyzv[4 * k + 1] = c.y; out_xyzv[4 * k + 2] = c.z;
    out_xyzv[4 * k + 3] = v;
    ++k;
  });
}

// result9: score, pose7..., (pose at [1..7]), rotational_score, low_resolution_score
int orc_fast3d_match(void* h, int full_submap, const double* node7, const double* submap7,
                     const double* gravity_wxyz, const float* hi, int nhi, const float* lo,
                     int nlo, const float* hist, int nh, float min_score, double* result10,
                     int64_t* stats4) {
  const auto& m = *static_cast<Fast3D*>(h)->matcher;
  const NodeData3D data = MakeNodeData(gravity_wxyz, hi, nhi, lo, nlo, hist, nh);
  Result3D r{};
  Stats3D st;
  bool ok;
  if (full_submap) {
    ok = m.MatchFullSubmap({node7[3], node7[4], node7[5], node7[6]},
                           {submap7[3], submap7[4], submap7[5], submap7[6]}, data, min_score, &r,
                           &st);
  } else {
    ok = m.Match(MakePose3(node7), MakePose3(submap7), data, min_score, &r, &st);
  }
  if (ok) {
    result10[0] = r.score;
    StorePose3(r.pose_estimate, result10 + 1);
    result10[8] = r.rotational_score;
    result10[9] = r.low_resolution_score;
  }
  if (stats4) {
    stats4[0] = st.candidates_scored; stats4[1] = st.num_scans;
    stats4[2] = st.coarse_candidates; stats4[3] = st.nodes_expanded;
  }
  return ok ? 1 : 0;
}


// ---- CeresScanMatcher2D ----
// options5 = occupied_space_weight, translation_weight, rotation_weight, use_nonmonotonic_steps,
// max_num_iterations; summary5 = initial_cost, final_cost, successful, unsuccessful, termination.
void orc_ceres2d_match(const uint16_t* cells, int nx, int ny, double res, double max_x,
                       double max_y, const double* options5, const double* target_xy,
                       const double* init_xyt, const float* xyz, int n, double* pose_xyt,
                       double* summary5) {
  CeresOptions2D o;
  o.occupied_space_weight = options5[0]; o.translation_weight = options5[1];
  o.rotation_weight = options5[2]; o.use_nonmonotonic_steps = options5[3] != 0.;
  o.max_num_iterations = static_cast<int>(options5[4]);
  Pose2d out;
  CeresSummary2D sum;
  CeresScanMatcher2DMatch(o, target_xy, Pose2d{init_xyt[0], init_xyt[1], init_xyt[2]},
                          MakeCloud(xyz, n), MakeView(cells, nx, ny, res, max_x, max_y), &out, &sum);
  pose_xyt[0] = out.x; pose_xyt[1] = out.y; pose_xyt[2] = out.theta;
  summary5[0] = sum.initial_cost; summary5[1] = sum.final_cost;
  summary5[2] = sum.num_successful_steps; summary5[3] = sum.num_unsuccessful_steps;
  summary5[4] = sum.termination;
}
// Residuals (n + 3) and Jacobian ((n + 3) x 3) of the three residual blocks at `pose`.
void orc_ceres2d_residuals(const uint16_t* cells, int nx, int ny, double res, double max_x,
                           double max_y, const double* options5, const double* target_xy,
                           double target_angle, const double* pose_xyt, const float* xyz, int n,
                           double* residuals, double* jacobian) {
  CeresOptions2D o;
  o.occupied_space_weight = options5[0]; o.translation_weight = options5[1];
  o.rotation_weight = options5[2];
  std::vector<double> r, J;
  CeresResiduals2D(o, target_xy, target_angle, MakeCloud(xyz, n),
                   MakeView(cells, nx, ny, res, max_x, max_y), pose_xyt, &r, &J);
  std::memcpy(residuals, r.data(), r.size() * sizeof(double));
  std::memcpy(jacobian, J.data(), J.size() * sizeof(double));
}


// ---- voxel filters / rotational histogram ----
void orc_voxel_filter_flags(const float* xyz, int n, float resolution, uint8_t* used) {
  const std::vector<uint8_t> f = VoxelFilterFlags(MakeCloud(xyz, n), resolution);
  std::memcpy(used, f.data(), f.size());
}
int orc_adaptive_voxel_filter(const float* xyz, int n, float max_length, float min_num_points,
                              float max_range, float* out_xyz) {
  const PointCloud r = AdaptiveVoxelFilter(MakeCloud(xyz, n), max_length, min_num_points, max_range);
  for (size_t i = 0; i != r.size(); ++i) {
    out_xyz[3 * i] = r[i].x; out_xyz[3 * i + 1] = r[i].y; out_xyz[3 * i + 2] = r[i].z;
  }
  return static_cast<int>(r.size());
}
void orc_compute_histogram(const float* xyz, int n, int histogram_size, float* out) {
  const std::vector<float> h = ComputeHistogram(MakeCloud(xyz, n), histogram_size);
  std::memcpy(out, h.data(), h.size() * sizeof(float));
}

// ---- CeresScanMatcher3D ----
// options8 = translation weight, rotation weight, only_optimize_yaw, use_nonmonotonic_steps,
// max_num_iterations, then one occupied-space weight per pair (up to 3).  Pair k: clouds[k]
// (xyz, counts[k] points) against grid k (resolutions[k], voxels[k], voxel_counts[k]).
namespace {
struct Ceres3DArgs {
  CeresOptions3D options;
  std::vector<PointCloud3> clouds;
  std::vector<std::unique_ptr<HybridGridView>> grids;
  std::vector<std::unique_ptr<IntensityGridView>> intensity_grids;
  std::vector<std::vector<float>> intensities;
  std::vector<CloudAndGrid3D> pairs;
};
// Intensity part of the pairs (null `intensities`: none): per pair point intensities, intensity
// voxels (x, y, z, count, sum) and options3 = weight, huber_scale, intensity_threshold.
void AddIntensity3D(Ceres3DArgs* a, const float* const* intensities,
                    const IntensityVoxel* const* voxels, const int64_t* voxel_counts,
                    const double* options3, const float* resolutions) {
  if (intensities == nullptr) return;
  const size_t num = a->pairs.size();
  a->intensity_grids.resize(num);
  a->intensities.resize(num);
  for (size_t k = 0; k < num; ++k) {
    if (intensities[k] == nullptr) continue;
    a->intensities[k].assign(intensities[k], intensities[k] + a->clouds[k].size());
    a->intensity_grids[k].reset(new IntensityGridView(resolutions[k], voxels[k], voxel_counts[k]));
    a->pairs[k].intensity_hybrid_grid = a->intensity_grids[k].get();
    a->pairs[k].intensities = &a->intensities[k];
    a->pairs[k].intensity_weight = options3[3 * k];
    a->pairs[k].huber_scale = options3[3 * k + 1];
    a->pairs[k].intensity_threshold = static_cast<float>(options3[3 * k + 2]);
  }
}
void MakeCeres3D(const double* options8, int num_pairs, const float* const* clouds,
                 const int* counts, const float* resolutions, const Voxel* const* voxels,
                 const int64_t* voxel_counts, Ceres3DArgs* a) {
  a->options.translation_weight = options8[0];
  a->options.rotation_weight = options8[1];
  a->options.only_optimize_yaw = options8[2] != 0.;
  a->options.use_nonmonotonic_steps = options8[3] != 0.;
  a->options.max_num_iterations = static_cast<int>(options8[4]);
  a->clouds.resize(num_pairs);
  for (int k = 0; k < num_pairs; ++k) {
    a->options.occupied_space_weight.push_back(options8[5 + k]);
    a->clouds[k] = MakeCloud3(clouds[k], counts[k]);
    a->grids.emplace_back(new HybridGridView(resolutions[k], voxels[k], voxel_counts[k]));
  }
  for (int k = 0; k < num_pairs; ++k)
    a->pairs.push_back(CloudAndGrid3D{&a->clouds[k], a->grids[k].get()});
}
}  // namespace

void orc_ceres3d_match(const double* options8, int num_pairs, const float* const* clouds,
                       const int* counts, const float* resolutions, const Voxel* const* voxels,
                       const int64_t* voxel_counts, const double* target_xyz,
                       const double* init7, double* pose7, double* summary5) {
  Ceres3DArgs a;
  MakeCeres3D(options8, num_pairs, clouds, counts, resolutions, voxels, voxel_counts, &a);
  Pose3d pose = MakePose3(init7);
  CeresSummary2D sum;
  CeresScanMatcher3DMatch(a.options, target_xyz, MakePose3(init7), a.pairs, &pose, &sum);
  StorePose3(pose, pose7);
  summary5[0] = sum.initial_cost; summary5[1] = sum.final_cost;
  summary5[2] = sum.num_successful_steps; summary5[3] = sum.num_unsuccessful_steps;
  summary5[4] = sum.termination;
}

// residuals [sum(counts) + 6], jacobian [sum(counts) + 6][7] at pose7 (t, q = w x y z);
// the rotation residual targets target_q4.
void orc_ceres3d_residuals(const double* options8, int num_pairs, const float* const* clouds,
                           const int* counts, const float* resolutions,
                           const Voxel* const* voxels, const int64_t* voxel_counts,
                           const double* target_xyz, const double* target_q4, const double* pose7,
                           double* residuals, double* jacobian) {
  Ceres3DArgs a;
  MakeCeres3D(options8, num_pairs, clouds, counts, resolutions, voxels, voxel_counts, &a);
  std::vector<double> r, J;
  CeresResiduals3D(a.options, target_xyz, target_q4, a.pairs, pose7, pose7 + 3, &r, &J);
  std::memcpy(residuals, r.data(), r.size() * sizeof(double));
  std::memcpy(jacobian, J.data(), J.size() * sizeof(double));
}


// The same two entry points with IntensityCostFunction3D blocks (argument layout of
// refc_ceres3d_match in ref_ceres_wrapper.cc).  Residuals / Jacobian are the RAW blocks in
// CeresResidualBlocks3D order (before the Huber correction).
void orc_ceres3d_match_intensity(const double* options8, int num_pairs, const float* const* clouds,
                                 const int* counts, const float* resolutions,
                                 const Voxel* const* voxels, const int64_t* voxel_counts,
                                 const double* target_xyz, const double* init7, double* pose7,
                                 double* summary5, const float* const* intensities,
                                 const IntensityVoxel* const* intensity_voxels,
                                 const int64_t* intensity_voxel_counts,
                                 const double* intensity_options3) {
  Ceres3DArgs a;
  MakeCeres3D(options8, num_pairs, clouds, counts, resolutions, voxels, voxel_counts, &a);
  AddIntensity3D(&a, intensities, intensity_voxels, intensity_voxel_counts, intensity_options3,
                 resolutions);
  Pose3d pose = MakePose3(init7);
  CeresSummary2D sum;
  CeresScanMatcher3DMatch(a.options, target_xyz, MakePose3(init7), a.pairs, &pose, &sum);
  StorePose3(pose, pose7);
  summary5[0] = sum.initial_cost; summary5[1] = sum.final_cost;
  summary5[2] = sum.num_successful_steps; summary5[3] = sum.num_unsuccessful_steps;
  summary5[4] = sum.termination;
}
// InsertIntensitiesIntoGrid + IntensityHybridGrid::AddIntensity
// (mapping/3d/range_data_inserter_3d.cc:54-70, mapping/3d/hybrid_grid.h:552-556) onto a voxel
// list: returns above the threshold are skipped (`>`), the others add count += 1 and, in point
// order, sum += intensity (f32).  `voxels` holds `count_in` voxels on entry and the result sorted
// (z, y, x) on return (at most `capacity`); returns the number of voxels.
int64_t orc_insert_intensities(float resolution, const float* returns_xyz, const float* intensities,
                               int n, float intensity_threshold, IntensityVoxel* voxels,
                               int64_t count_in, int64_t capacity) {
  std::map<std::tuple<int, int, int>, std::pair<float, int>> cells;       // (z, y, x) -> sum, count
  for (int64_t k = 0; k < count_in; ++k)
    cells[std::make_tuple(voxels[k].z, voxels[k].y, voxels[k].x)] = {voxels[k].sum, voxels[k].count};
  const IntensityGridView index_of(resolution, nullptr, 0);
  if (intensities != nullptr) {                                           // (:57)
    for (int i = 0; i < n; ++i) {
      if (intensities[i] > intensity_threshold) continue;                 // (:59-61)
      const Cell3i c = index_of.GetCellIndex(V3f{returns_xyz[3 * i], returns_xyz[3 * i + 1],
                                                 returns_xyz[3 * i + 2]});
      auto& cell = cells[std::make_tuple(c.z, c.y, c.x)];
      cell.second += 1;
      cell.first += intensities[i];
    }
  }
  int64_t k = 0;
  for (const auto& kv : cells) {
    if (kv.second.second == 0) continue;
    if (k < capacity)
      voxels[k] = IntensityVoxel{std::get<2>(kv.first), std::get<1>(kv.first),
                                 std::get<0>(kv.first), kv.second.second, kv.second.first};
    ++k;
  }
  return k;
}

// IntensityCostFunction3D alone, like refc_intensity3d_residuals: residuals [n], jacobian [n][7].
void orc_intensity3d_residuals(double scaling_factor, float intensity_threshold, const float* xyz,
                               const float* intensities, int n, float resolution,
                               const IntensityVoxel* voxels, int64_t num_voxels,
                               const double* pose7, double* residuals, double* jacobian) {
  // one pair whose occupied-space block is dropped from the output: weight sqrt(n) * scaling
  const Voxel none{0, 0, 0, 0, 0};
  const HybridGridView empty(resolution, &none, 0);
  const PointCloud3 cloud = MakeCloud3(xyz, n);
  const IntensityGridView grid(resolution, voxels, num_voxels);
  const std::vector<float> in(intensities, intensities + n);
  CloudAndGrid3D pair{&cloud, &empty};
  pair.intensity_hybrid_grid = &grid;
  pair.intensities = &in;
  pair.intensity_weight = scaling_factor * std::sqrt(static_cast<double>(n));
  pair.intensity_threshold = intensity_threshold;
  CeresOptions3D o;
  o.occupied_space_weight = {1.};
  const double zero3[3] = {0, 0, 0}, ident[4] = {1, 0, 0, 0};
  std::vector<double> r, J;
  CeresResiduals3D(o, zero3, ident, {pair}, pose7, pose7 + 3, &r, &J);
  std::memcpy(residuals, r.data() + n, n * sizeof(double));
  std::memcpy(jacobian, J.data() + 7 * static_cast<size_t>(n), 7 * static_cast<size_t>(n) * sizeof(double));
}

}  // extern "C"
