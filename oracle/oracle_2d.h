// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
// 2D correlative scan matching restated from
//   cartographer/mapping/internal/2d/scan_matching/correlative_scan_matcher_2d.{h,cc}
//   .../real_time_correlative_scan_matcher_2d.cc
//   .../fast_correlative_scan_matcher_2d.{h,cc}
// and the grid read accessors they use (mapping/2d/{map_limits.h,grid_2d.h,
// probability_grid.cc}).
#ifndef ORACLE_2D_H_
#define ORACLE_2D_H_

#include <cstdint>
#include <utility>
#include <vector>

#include "oracle_common.h"

namespace oracle {

struct Cell2i { int x, y; };
typedef std::vector<Cell2i> DiscreteScan2D;
struct Point3f { float x, y, z; };
typedef std::vector<Point3f> PointCloud;
struct Pose2d { double x, y, theta; };

// mapping/2d/map_limits.h:40-96.
struct MapLimits {
  double resolution;
  double max_x, max_y;
  int num_x_cells, num_y_cells;
  Cell2i GetCellIndex(float px, float py) const;   // map_limits.h:69-76
  bool Contains(const Cell2i& c) const {           // map_limits.h:85-90
    return 0 <= c.x && 0 <= c.y && c.x < num_x_cells && c.y < num_y_cells;
  }
};

// Read-only view of a ProbabilityGrid (grid_2d.h:53-57, probability_grid.cc:78-82).
struct ProbabilityGridView {
  MapLimits limits;
  const uint16_t* cells;  // row-major nx*iy+ix, 0 = unknown (grid_2d.h:113-116)
  float min_correspondence_cost = kMinCorrespondenceCost;
  float max_correspondence_cost = kMaxCorrespondenceCost;
  float GetCorrespondenceCost(const Cell2i& c) const;
  float GetProbability(const Cell2i& c) const;
};

// correlative_scan_matcher_2d.h:36-63, .cc:27-91.
struct SearchParameters {
  struct LinearBounds { int min_x, max_x, min_y, max_y; };
  SearchParameters(double linear_search_window, double angular_search_window,
                   const PointCloud& point_cloud, double resolution);
  SearchParameters(int num_linear_perturbations, int num_angular_perturbations,
                   double angular_perturbation_step_size, double resolution);
  void ShrinkToFit(const std::vector<DiscreteScan2D>& scans, int num_x_cells,
                   int num_y_cells);
  int num_angular_perturbations;
  double angular_perturbation_step_size;
  double resolution;
  int num_scans;
  std::vector<LinearBounds> linear_bounds;
};

PointCloud RotateCloudYaw(const PointCloud& cloud, float angle);  // TransformPointCloud by AngleAxisf(angle, Z)
std::vector<PointCloud> GenerateRotatedScans(const PointCloud& cloud,
                                             const SearchParameters& sp);  // .cc:93-109
std::vector<DiscreteScan2D> DiscretizeScans(const MapLimits& limits,
                                            const std::vector<PointCloud>& scans,
                                            float tx, float ty);  // .cc:111-127

// correlative_scan_matcher_2d.h:74-103.
struct Candidate2D {
  Candidate2D(int scan_index, int x_off, int y_off, const SearchParameters& sp)
      : scan_index(scan_index), x_index_offset(x_off), y_index_offset(y_off),
        x(-y_off * sp.resolution), y(-x_off * sp.resolution),
        orientation((scan_index - sp.num_angular_perturbations) *
                    sp.angular_perturbation_step_size) {}
  int scan_index, x_index_offset, y_index_offset;
  double x, y, orientation;
  float score = 0.f;
  bool operator<(const Candidate2D& o) const { return score < o.score; }
  bool operator>(const Candidate2D& o) const { return score > o.score; }
};

struct MatchStats {
  int64_t candidates_scored = 0;   // every ScoreCandidates element, all depths
  int64_t num_scans = 0;
  int64_t coarse_candidates = 0;
  int64_t nodes_expanded = 0;
};

// real_time_correlative_scan_matcher_2d.cc:117-176 (probability-grid branch).
double RealTimeMatch2D(const ProbabilityGridView& grid, const Pose2d& initial,
                       const PointCloud& cloud, double linear_window,
                       double angular_window, double translation_delta_cost_weight,
                       double rotation_delta_cost_weight, Pose2d* pose_estimate,
                       MatchStats* stats = nullptr,
                       std::vector<float>* all_scores = nullptr);

// TSDF2D read path (mapping/internal/2d/tsdf_2d.{h,cc}, tsd_value_converter.{h,cc}):
// two uint16 planes (tsd, weight), 0 = unknown, bit 15 = update marker.
struct TsdfView {
  TsdfView(const MapLimits& limits, const uint16_t* tsd_cells, const uint16_t* weight_cells,
           float truncation_distance, float max_weight);
  MapLimits limits;
  const uint16_t* tsd_cells;
  const uint16_t* weight_cells;
  float max_tsd, min_tsd, max_weight;
  std::pair<float, float> GetTSDAndWeight(const Cell2i& c) const;
};

// RealTimeCorrelativeScanMatcher2D::Match on a TSDF2D
// (real_time_correlative_scan_matcher_2d.cc:38-59,117-176).
double RealTimeMatch2DTsdf(const TsdfView& tsdf, const Pose2d& initial, const PointCloud& cloud,
                           double linear_window, double angular_window,
                           double translation_delta_cost_weight,
                           double rotation_delta_cost_weight, Pose2d* pose_estimate,
                           MatchStats* stats = nullptr,
                           std::vector<float>* all_scores = nullptr);

// fast_correlative_scan_matcher_2d.h:49-93, .cc:91-169.
class PrecomputationGrid2D {
 public:
  PrecomputationGrid2D(const ProbabilityGridView& grid, int width);
  int GetValue(int x, int y) const {
    const int lx = x - offset_x_, ly = y - offset_y_;
    if (static_cast<unsigned>(lx) >= static_cast<unsigned>(wide_x_) ||
        static_cast<unsigned>(ly) >= static_cast<unsigned>(wide_y_)) return 0;
    return cells_[lx + ly * wide_x_];
  }
  float ToScore(float value) const {
    return min_score_ + value * ((max_score_ - min_score_) / 255.f);
  }
  int wide_x() const { return wide_x_; }
  int wide_y() const { return wide_y_; }
  const std::vector<uint8_t>& cells() const { return cells_; }
 private:
  int offset_x_, offset_y_, wide_x_, wide_y_;
  float min_score_, max_score_;
  std::vector<uint8_t> cells_;
};

// fast_correlative_scan_matcher_2d.cc:171-378.
class FastCorrelativeScanMatcher2D {
 public:
  FastCorrelativeScanMatcher2D(const ProbabilityGridView& grid, int branch_and_bound_depth,
                               double linear_search_window, double angular_search_window);
  bool Match(const Pose2d& initial, const PointCloud& cloud, float min_score,
             float* score, Pose2d* pose, MatchStats* stats = nullptr) const;
  bool MatchFullSubmap(const PointCloud& cloud, float min_score, float* score,
                       Pose2d* pose, MatchStats* stats = nullptr) const;
  const PrecomputationGrid2D& level(int i) const { return stack_[i]; }
  int depth() const { return static_cast<int>(stack_.size()); }
  const MapLimits& limits() const { return limits_; }

  // Introspection for the parity tests: the prepared search (discrete scans,
  // shrunk bounds) and the scored lowest-resolution candidates in generation
  // order (before the sort).
  struct Prepared {
    std::vector<DiscreteScan2D> discrete_scans;
    std::vector<SearchParameters::LinearBounds> bounds;
    int num_angular_perturbations;
    double angular_step;
  };
  Prepared Prepare(const Pose2d& initial, const PointCloud& cloud, bool full_submap,
                   Pose2d* used_initial) const;
  std::vector<int> CoarseSums(const Prepared& p) const;

 private:
  bool MatchWithSearchParameters(SearchParameters sp, const Pose2d& initial,
                                 const PointCloud& cloud, float min_score, float* score,
                                 Pose2d* pose, MatchStats* stats) const;
  std::vector<Candidate2D> GenerateLowestResolutionCandidates(const SearchParameters& sp) const;
  void ScoreCandidates(const PrecomputationGrid2D& grid,
                       const std::vector<DiscreteScan2D>& scans,
                       std::vector<Candidate2D>* candidates, MatchStats* stats) const;
  Candidate2D BranchAndBound(const std::vector<DiscreteScan2D>& scans,
                             const SearchParameters& sp,
                             const std::vector<Candidate2D>& candidates, int depth,
                             float min_score, MatchStats* stats) const;
  MapLimits limits_;
  double linear_search_window_, angular_search_window_;
  std::vector<PrecomputationGrid2D> stack_;
};

}  // namespace oracle

#endif  // ORACLE_2D_H_
