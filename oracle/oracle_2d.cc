// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
#include "oracle_2d.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <deque>
#include <functional>

namespace oracle {

// ---------------------------------------------------------------- tables ---
const std::vector<float>& ValueToProbabilityTable() {
  static const std::vector<float> t = [] {
    std::vector<float> r;
    r.reserve(2 * 32768);
    for (int rep = 0; rep != 2; ++rep)
      for (int v = 0; v != 32768; ++v)
        r.push_back(SlowValueToBoundedFloat32768(v, kMinProbability, kMinProbability,
                                                 kMaxProbability));
    return r;
  }();
  return t;
}
const std::vector<float>& ValueToCorrespondenceCostTable() {
  static const std::vector<float> t = [] {
    std::vector<float> r;
    r.reserve(2 * 32768);
    for (int rep = 0; rep != 2; ++rep)
      for (int v = 0; v != 32768; ++v)
        r.push_back(SlowValueToBoundedFloat32768(v, kMaxCorrespondenceCost,
                                                 kMinCorrespondenceCost,
                                                 kMaxCorrespondenceCost));
    return r;
  }();
  return t;
}
const std::vector<float>& GridCorrespondenceCostTable() {
  // value_conversion_tables.cc:29-51: scale uses the literal 32766.f.
  static const std::vector<float> t = [] {
    std::vector<float> r;
    r.reserve(65536);
    const float lo = kMinCorrespondenceCost, hi = kMaxCorrespondenceCost;
    for (int v = 0; v != 65536; ++v) {
      const uint16_t m = static_cast<uint16_t>(v) & static_cast<uint16_t>(~kUpdateMarker);
      if (m == 0) {
        r.push_back(kMaxCorrespondenceCost);
      } else {
        const float scale = (hi - lo) / 32766.f;
        r.push_back(m * scale + (lo - scale));
      }
    }
    return r;
  }();
  return t;
}

// ------------------------------------------------------------ map limits ---
Cell2i MapLimits::GetCellIndex(float px, float py) const {
  // map_limits.h:73-75: x index from y, y index from x; f64 divide, lround.
  return {RoundToInt((max_y - static_cast<double>(py)) / resolution - 0.5),
          RoundToInt((max_x - static_cast<double>(px)) / resolution - 0.5)};
}

float ProbabilityGridView::GetCorrespondenceCost(const Cell2i& c) const {
  if (!limits.Contains(c)) return max_correspondence_cost;
  const uint16_t raw = cells[limits.num_x_cells * c.y + c.x];
  if (min_correspondence_cost == kMinCorrespondenceCost &&
      max_correspondence_cost == kMaxCorrespondenceCost)
    return GridCorrespondenceCostTable()[raw];
  // Any other Grid2D (a TSDF2D: [-truncation_distance, truncation_distance], tsdf_2d.cc:25-26):
  // the entry of the grid's own table, value_conversion_tables.cc:29-38 -- unknown (0) is
  // max_correspondence_cost (grid_2d.cc:60-66), the update marker is masked.
  const uint16_t value = raw & 0x7fffu;
  if (value == 0) return max_correspondence_cost;
  const float kScale = (max_correspondence_cost - min_correspondence_cost) / 32766.f;
  return value * kScale + (min_correspondence_cost - kScale);
}
float ProbabilityGridView::GetProbability(const Cell2i& c) const {
  // probability_grid.cc:78-82 uses the *global* 32768-entry table.
  if (!limits.Contains(c)) return kMinProbability;
  return 1.f - ValueToCorrespondenceCostTable()[cells[limits.num_x_cells * c.y + c.x]];
}

// ------------------------------------------------------ search parameters ---
SearchParameters::SearchParameters(const double linear_search_window,
                                   const double angular_search_window,
                                   const PointCloud& point_cloud, const double resolution)
    : resolution(resolution) {
  float max_scan_range = 3.f * resolution;  // f64 product narrowed to f32
  for (const Point3f& p : point_cloud) {
    const float range = std::sqrt(p.x * p.x + p.y * p.y);
    max_scan_range = std::max(range, max_scan_range);
  }
  const double kSafetyMargin = 1. - 1e-3;
  const float range_sq = max_scan_range * (max_scan_range * 1.f);  // Pow2<float>
  const double res_sq = resolution * (resolution * 1.);
  angular_perturbation_step_size =
      kSafetyMargin * std::acos(1. - res_sq / (2. * range_sq));
  num_angular_perturbations =
      std::ceil(angular_search_window / angular_perturbation_step_size);
  num_scans = 2 * num_angular_perturbations + 1;
  const int num_linear_perturbations = std::ceil(linear_search_window / resolution);
  linear_bounds.assign(num_scans,
                       LinearBounds{-num_linear_perturbations, num_linear_perturbations,
                                    -num_linear_perturbations, num_linear_perturbations});
}

SearchParameters::SearchParameters(const int num_linear_perturbations,
                                   const int num_angular_perturbations,
                                   const double angular_perturbation_step_size,
                                   const double resolution)
    : num_angular_perturbations(num_angular_perturbations),
      angular_perturbation_step_size(angular_perturbation_step_size),
      resolution(resolution),
      num_scans(2 * num_angular_perturbations + 1) {
  linear_bounds.assign(num_scans,
                       LinearBounds{-num_linear_perturbations, num_linear_perturbations,
                                    -num_linear_perturbations, num_linear_perturbations});
}

void SearchParameters::ShrinkToFit(const std::vector<DiscreteScan2D>& scans,
                                   const int num_x_cells, const int num_y_cells) {
  assert(static_cast<int>(scans.size()) == num_scans);
  for (int i = 0; i != num_scans; ++i) {
    int lo_x = 0, lo_y = 0, hi_x = 0, hi_y = 0;
    for (const Cell2i& c : scans[i]) {
      lo_x = std::min(lo_x, -c.x);
      lo_y = std::min(lo_y, -c.y);
      hi_x = std::max(hi_x, num_x_cells - 1 - c.x);
      hi_y = std::max(hi_y, num_y_cells - 1 - c.y);
    }
    LinearBounds& b = linear_bounds[i];
    b.min_x = std::max(b.min_x, lo_x);
    b.max_x = std::min(b.max_x, hi_x);
    b.min_y = std::max(b.min_y, lo_y);
    b.max_y = std::min(b.max_y, hi_y);
  }
}

PointCloud RotateCloudYaw(const PointCloud& cloud, const float angle) {
  const Qf q = QuatFromYaw(angle);
  PointCloud out;
  out.reserve(cloud.size());
  for (const Point3f& p : cloud) {
    const V3f r = Rotate(q, V3f{p.x, p.y, p.z});
    // Rigid3f::Rotation(...) * point = rotation * point + Vector3f::Zero().
    out.push_back(Point3f{r.x + 0.f, r.y + 0.f, r.z + 0.f});
  }
  return out;
}

std::vector<PointCloud> GenerateRotatedScans(const PointCloud& cloud,
                                             const SearchParameters& sp) {
  std::vector<PointCloud> rotated;
  rotated.reserve(sp.num_scans);
  double delta_theta = -sp.num_angular_perturbations * sp.angular_perturbation_step_size;
  for (int s = 0; s < sp.num_scans; ++s, delta_theta += sp.angular_perturbation_step_size) {
    rotated.push_back(RotateCloudYaw(cloud, static_cast<float>(delta_theta)));
  }
  return rotated;
}

std::vector<DiscreteScan2D> DiscretizeScans(const MapLimits& limits,
                                            const std::vector<PointCloud>& scans,
                                            const float tx, const float ty) {
  std::vector<DiscreteScan2D> out;
  out.reserve(scans.size());
  for (const PointCloud& scan : scans) {
    out.emplace_back();
    out.back().reserve(scan.size());
    for (const Point3f& p : scan) {
      // Affine2f(translation) * v: identity linear part, then + t, in f32.
      const float x = (1.f * p.x + 0.f * p.y) + tx;
      const float y = (0.f * p.x + 1.f * p.y) + ty;
      out.back().push_back(limits.GetCellIndex(x, y));
    }
  }
  return out;
}

// ------------------------------------------------------------ real-time 2D ---
namespace {

// RealTimeCorrelativeScanMatcher2D::Match (:117-149); `candidate_score` is
// ComputeCandidateScore for the grid type at hand.
template <typename CandidateScore>
double RealTimeMatchImpl(const MapLimits& limits, CandidateScore candidate_score,
                         const Pose2d& initial, const PointCloud& cloud,
                         const double linear_window, const double angular_window,
                         const double tw, const double rw, Pose2d* pose_estimate,
                         MatchStats* stats, std::vector<float>* all_scores) {
  const PointCloud rotated_cloud = RotateCloudYaw(cloud, static_cast<float>(initial.theta));
  const SearchParameters sp(linear_window, angular_window, rotated_cloud, limits.resolution);
  const std::vector<PointCloud> rotated_scans = GenerateRotatedScans(rotated_cloud, sp);
  const std::vector<DiscreteScan2D> scans = DiscretizeScans(
      limits, rotated_scans, static_cast<float>(initial.x), static_cast<float>(initial.y));
  // GenerateExhaustiveSearchCandidates (:83-115): scan, x, y nesting.
  std::vector<Candidate2D> candidates;
  for (int s = 0; s != sp.num_scans; ++s)
    for (int x = sp.linear_bounds[s].min_x; x <= sp.linear_bounds[s].max_x; ++x)
      for (int y = sp.linear_bounds[s].min_y; y <= sp.linear_bounds[s].max_y; ++y)
        candidates.emplace_back(s, x, y, sp);
  // ScoreCandidates (:151-176).
  for (Candidate2D& c : candidates) {
    c.score = candidate_score(scans[c.scan_index], c.x_index_offset, c.y_index_offset);
    const double t = std::hypot(c.x, c.y) * tw + std::abs(c.orientation) * rw;
    c.score *= std::exp(-(t * (t * 1.)));
  }
  if (stats) {
    stats->candidates_scored += candidates.size();
    stats->num_scans = sp.num_scans;
    stats->coarse_candidates = candidates.size();
  }
  if (all_scores) {
    all_scores->clear();
    for (const Candidate2D& c : candidates) all_scores->push_back(c.score);
  }
  const Candidate2D& best = *std::max_element(candidates.begin(), candidates.end());
  *pose_estimate = Pose2d{initial.x + best.x, initial.y + best.y,
                          initial.theta + best.orientation};
  return best.score;
}

}  // namespace

double RealTimeMatch2D(const ProbabilityGridView& grid, const Pose2d& initial,
                       const PointCloud& cloud, const double linear_window,
                       const double angular_window, const double tw, const double rw,
                       Pose2d* pose_estimate, MatchStats* stats,
                       std::vector<float>* all_scores) {
  // ComputeCandidateScore(ProbabilityGrid) (:61-75).
  const auto score = [&grid](const DiscreteScan2D& scan, const int dx, const int dy) {
    float acc = 0.f;
    for (const Cell2i& idx : scan) acc += grid.GetProbability(Cell2i{idx.x + dx, idx.y + dy});
    acc /= static_cast<float>(scan.size());
    return acc;
  };
  return RealTimeMatchImpl(grid.limits, score, initial, cloud, linear_window, angular_window, tw,
                           rw, pose_estimate, stats, all_scores);
}

// ---------------------------------------------------------------- TSDF2D ---
TsdfView::TsdfView(const MapLimits& l, const uint16_t* tsd, const uint16_t* weight,
                   const float truncation_distance, const float max_weight_in)
    : limits(l), tsd_cells(tsd), weight_cells(weight), max_tsd(truncation_distance),
      min_tsd(-truncation_distance), max_weight(max_weight_in) {}

namespace {
// ValueConversionTables (mapping/value_conversion_tables.cc:29-52).
float SlowValueToBoundedFloat(const uint16_t raw, const float unknown_result,
                              const float lower_bound, const float upper_bound) {
  const uint16_t value = raw & static_cast<uint16_t>(~kUpdateMarker);
  if (value == 0) return unknown_result;
  const float kScale = (upper_bound - lower_bound) / 32766.f;
  return value * kScale + (lower_bound - kScale);
}
}  // namespace

// TSDF2D::GetTSDAndWeight (mapping/internal/2d/tsdf_2d.cc:88-98) with the
// tables TSDValueConverter builds (tsd_value_converter.cc:22-33): unknown tsd
// -> min_tsd, unknown weight -> 0.
std::pair<float, float> TsdfView::GetTSDAndWeight(const Cell2i& c) const {
  if (limits.Contains(c)) {
    const size_t flat = static_cast<size_t>(limits.num_x_cells) * c.y + c.x;
    return {SlowValueToBoundedFloat(tsd_cells[flat], min_tsd, min_tsd, max_tsd),
            SlowValueToBoundedFloat(weight_cells[flat], 0.f, 0.f, max_weight)};
  }
  return {min_tsd, 0.f};
}

double RealTimeMatch2DTsdf(const TsdfView& tsdf, const Pose2d& initial, const PointCloud& cloud,
                           const double linear_window, const double angular_window,
                           const double tw, const double rw, Pose2d* pose_estimate,
                           MatchStats* stats, std::vector<float>* all_scores) {
  // ComputeCandidateScore(TSDF2D) (:38-59); GetMaxCorrespondenceCost() is the
  // truncation distance (tsdf_2d.cc:25-26).
  const auto score = [&tsdf](const DiscreteScan2D& scan, const int dx, const int dy) {
    float candidate_score = 0.f;
    float summed_weight = 0.f;
    for (const Cell2i& idx : scan) {
      const std::pair<float, float> tw_ = tsdf.GetTSDAndWeight(Cell2i{idx.x + dx, idx.y + dy});
      const float normalized_tsd_score = (tsdf.max_tsd - std::abs(tw_.first)) / tsdf.max_tsd;
      const float weight = tw_.second;
      candidate_score += normalized_tsd_score * weight;
      summed_weight += weight;
    }
    if (summed_weight == 0.f) return 0.f;
    candidate_score /= summed_weight;
    return candidate_score;
  };
  return RealTimeMatchImpl(tsdf.limits, score, initial, cloud, linear_window, angular_window, tw,
                           rw, pose_estimate, stats, all_scores);
}

// ------------------------------------------------------- precomputation 2D ---
namespace {

// Maximum of every length-`width` window [x0, x0+width) clipped to [0, n), for
// x0 in [-width+1, n-1]; out has n+width-1 entries (entry x0+width-1).
// Restates the effect of SlidingWindowMaximum (:41-74) + the three loops at
// :109-131 with a monotonic deque of (position, value).
template <typename Load>
void SlidingMax(const int n, const int width, Load load, float* out, const int out_stride) {
  std::deque<std::pair<int, float>> mono;  // values non-increasing front→back
  int next = 0;                            // next source position to push
  for (int x0 = -width + 1; x0 <= n - 1; ++x0) {
    const int hi = std::min(x0 + width - 1, n - 1);
    for (; next <= hi; ++next) {
      const float v = load(next);
      while (!mono.empty() && v > mono.back().second) mono.pop_back();
      mono.emplace_back(next, v);
    }
    while (mono.front().first < x0) mono.pop_front();
    out[(x0 + width - 1) * out_stride] = mono.front().second;
  }
}

}  // namespace

PrecomputationGrid2D::PrecomputationGrid2D(const ProbabilityGridView& grid, const int width)
    : offset_x_(-width + 1), offset_y_(-width + 1),
      wide_x_(grid.limits.num_x_cells + width - 1),
      wide_y_(grid.limits.num_y_cells + width - 1),
      min_score_(1.f - grid.max_correspondence_cost),
      max_score_(1.f - grid.min_correspondence_cost),
      cells_(static_cast<size_t>(wide_x_) * wide_y_) {
  const int nx = grid.limits.num_x_cells, ny = grid.limits.num_y_cells;
  const int stride = wide_x_;
  std::vector<float> intermediate(static_cast<size_t>(wide_x_) * ny);
  for (int y = 0; y != ny; ++y) {
    SlidingMax(nx, width,
               [&](int x) { return 1.f - std::abs(grid.GetCorrespondenceCost(Cell2i{x, y})); },
               &intermediate[static_cast<size_t>(y) * stride], 1);
  }
  std::vector<float> column(wide_y_);
  for (int x = 0; x != wide_x_; ++x) {
    SlidingMax(ny, width,
               [&](int y) { return intermediate[x + static_cast<size_t>(y) * stride]; },
               column.data(), 1);
    for (int y = 0; y != wide_y_; ++y) {
      // ComputeCellValue (:163-169).
      const int v = RoundToInt((column[y] - min_score_) * (255.f / (max_score_ - min_score_)));
      assert(v >= 0 && v <= 255);
      cells_[x + static_cast<size_t>(y) * stride] = static_cast<uint8_t>(v);
    }
  }
}

// ------------------------------------------------------------------ fast 2D ---
FastCorrelativeScanMatcher2D::FastCorrelativeScanMatcher2D(
    const ProbabilityGridView& grid, const int depth, const double linear_search_window,
    const double angular_search_window)
    : limits_(grid.limits), linear_search_window_(linear_search_window),
      angular_search_window_(angular_search_window) {
  assert(depth >= 1);
  stack_.reserve(depth);
  for (int i = 0; i != depth; ++i) stack_.emplace_back(grid, 1 << i);
}

bool FastCorrelativeScanMatcher2D::Match(const Pose2d& initial, const PointCloud& cloud,
                                         const float min_score, float* score, Pose2d* pose,
                                         MatchStats* stats) const {
  const SearchParameters sp(linear_search_window_, angular_search_window_, cloud,
                            limits_.resolution);
  return MatchWithSearchParameters(sp, initial, cloud, min_score, score, pose, stats);
}

bool FastCorrelativeScanMatcher2D::MatchFullSubmap(const PointCloud& cloud,
                                                   const float min_score, float* score,
                                                   Pose2d* pose, MatchStats* stats) const {
  const SearchParameters sp(1e6 * limits_.resolution, M_PI, cloud, limits_.resolution);
  // :219-222: centre = max - 0.5 * res * (num_y_cells, num_x_cells).
  const Pose2d center{limits_.max_x - 0.5 * limits_.resolution * limits_.num_y_cells,
                      limits_.max_y - 0.5 * limits_.resolution * limits_.num_x_cells, 0.};
  return MatchWithSearchParameters(sp, center, cloud, min_score, score, pose, stats);
}

FastCorrelativeScanMatcher2D::Prepared FastCorrelativeScanMatcher2D::Prepare(
    const Pose2d& initial_in, const PointCloud& cloud, const bool full_submap,
    Pose2d* used_initial) const {
  Pose2d initial = initial_in;
  SearchParameters sp = full_submap
      ? SearchParameters(1e6 * limits_.resolution, M_PI, cloud, limits_.resolution)
      : SearchParameters(linear_search_window_, angular_search_window_, cloud,
                         limits_.resolution);
  if (full_submap) {
    initial = Pose2d{limits_.max_x - 0.5 * limits_.resolution * limits_.num_y_cells,
                     limits_.max_y - 0.5 * limits_.resolution * limits_.num_x_cells, 0.};
  }
  if (used_initial) *used_initial = initial;
  const PointCloud rotated_cloud = RotateCloudYaw(cloud, static_cast<float>(initial.theta));
  const std::vector<PointCloud> rotated_scans = GenerateRotatedScans(rotated_cloud, sp);
  Prepared p;
  p.discrete_scans = DiscretizeScans(limits_, rotated_scans, static_cast<float>(initial.x),
                                     static_cast<float>(initial.y));
  sp.ShrinkToFit(p.discrete_scans, limits_.num_x_cells, limits_.num_y_cells);
  p.bounds = sp.linear_bounds;
  p.num_angular_perturbations = sp.num_angular_perturbations;
  p.angular_step = sp.angular_perturbation_step_size;
  return p;
}

std::vector<int> FastCorrelativeScanMatcher2D::CoarseSums(const Prepared& p) const {
  const int max_depth = depth() - 1;
  const int step = 1 << max_depth;
  const PrecomputationGrid2D& g = stack_[max_depth];
  std::vector<int> sums;
  for (size_t s = 0; s != p.discrete_scans.size(); ++s)
    for (int x = p.bounds[s].min_x; x <= p.bounds[s].max_x; x += step)
      for (int y = p.bounds[s].min_y; y <= p.bounds[s].max_y; y += step) {
        int sum = 0;
        for (const Cell2i& c : p.discrete_scans[s]) sum += g.GetValue(c.x + x, c.y + y);
        sums.push_back(sum);
      }
  return sums;
}

bool FastCorrelativeScanMatcher2D::MatchWithSearchParameters(
    SearchParameters sp, const Pose2d& initial, const PointCloud& cloud,
    const float min_score, float* score, Pose2d* pose, MatchStats* stats) const {
  const PointCloud rotated_cloud = RotateCloudYaw(cloud, static_cast<float>(initial.theta));
  const std::vector<PointCloud> rotated_scans = GenerateRotatedScans(rotated_cloud, sp);
  const std::vector<DiscreteScan2D> scans =
      DiscretizeScans(limits_, rotated_scans, static_cast<float>(initial.x),
                      static_cast<float>(initial.y));
  sp.ShrinkToFit(scans, limits_.num_x_cells, limits_.num_y_cells);

  std::vector<Candidate2D> lowest = GenerateLowestResolutionCandidates(sp);
  if (stats) {
    stats->num_scans = sp.num_scans;
    stats->coarse_candidates = lowest.size();
  }
  ScoreCandidates(stack_.back(), scans, &lowest, stats);
  const Candidate2D best =
      BranchAndBound(scans, sp, lowest, depth() - 1, min_score, stats);
  if (best.score > min_score) {
    *score = best.score;
    *pose = Pose2d{initial.x + best.x, initial.y + best.y, initial.theta + best.orientation};
    return true;
  }
  return false;
}

std::vector<Candidate2D> FastCorrelativeScanMatcher2D::GenerateLowestResolutionCandidates(
    const SearchParameters& sp) const {
  const int step = 1 << (depth() - 1);
  std::vector<Candidate2D> candidates;
  for (int s = 0; s != sp.num_scans; ++s)
    for (int x = sp.linear_bounds[s].min_x; x <= sp.linear_bounds[s].max_x; x += step)
      for (int y = sp.linear_bounds[s].min_y; y <= sp.linear_bounds[s].max_y; y += step)
        candidates.emplace_back(s, x, y, sp);
  return candidates;
}

void FastCorrelativeScanMatcher2D::ScoreCandidates(const PrecomputationGrid2D& grid,
                                                   const std::vector<DiscreteScan2D>& scans,
                                                   std::vector<Candidate2D>* candidates,
                                                   MatchStats* stats) const {
  for (Candidate2D& c : *candidates) {
    int sum = 0;
    for (const Cell2i& idx : scans[c.scan_index]) {
      sum += grid.GetValue(idx.x + c.x_index_offset, idx.y + c.y_index_offset);
    }
    c.score = grid.ToScore(sum / static_cast<float>(scans[c.scan_index].size()));
  }
  if (stats) stats->candidates_scored += candidates->size();
  std::sort(candidates->begin(), candidates->end(), std::greater<Candidate2D>());
}

Candidate2D FastCorrelativeScanMatcher2D::BranchAndBound(
    const std::vector<DiscreteScan2D>& scans, const SearchParameters& sp,
    const std::vector<Candidate2D>& candidates, const int candidate_depth, float min_score,
    MatchStats* stats) const {
  if (candidate_depth == 0) return *candidates.begin();
  Candidate2D best(0, 0, 0, sp);
  best.score = min_score;
  for (const Candidate2D& c : candidates) {
    if (c.score <= min_score) break;
    if (stats) ++stats->nodes_expanded;
    std::vector<Candidate2D> children;
    const int half = 1 << (candidate_depth - 1);
    for (int xo : {0, half}) {
      if (c.x_index_offset + xo > sp.linear_bounds[c.scan_index].max_x) break;
      for (int yo : {0, half}) {
        if (c.y_index_offset + yo > sp.linear_bounds[c.scan_index].max_y) break;
        children.emplace_back(c.scan_index, c.x_index_offset + xo, c.y_index_offset + yo, sp);
      }
    }
    ScoreCandidates(stack_[candidate_depth - 1], scans, &children, stats);
    best = std::max(best, BranchAndBound(scans, sp, children, candidate_depth - 1,
                                         best.score, stats));
  }
  return best;
}

}  // namespace oracle
