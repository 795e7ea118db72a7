// ORACLE — TEST INFRASTRUCTURE ONLY.  C wrapper around five reference translation units that
// `make -C oracle ref` compiles unmodified from /root/reference (see ref_shims/README.md).
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "cartographer/common/fixed_ratio_sampler.h"
#include "cartographer/mapping/internal/2d/ray_to_pixel_mask.h"
#include "cartographer/mapping/2d/probability_grid.h"
#include "cartographer/mapping/internal/2d/scan_matching/fast_correlative_scan_matcher_2d.h"
#include "cartographer/mapping/internal/2d/scan_matching/real_time_correlative_scan_matcher_2d.h"
#include "cartographer/mapping/internal/2d/tsd_value_converter.h"
#include "cartographer/mapping/internal/2d/tsdf_2d.h"
#include "cartographer/mapping/probability_values.h"
#include "cartographer/mapping/value_conversion_tables.h"

extern "C" {

// RayToPixelMask: returns the number of pixels; out gets min(count, capacity) (x, y) pairs.
int ref_ray_to_pixel_mask(int bx, int by, int ex, int ey, int subpixel_scale, int32_t* out_xy,
                          int capacity) {
  const std::vector<Eigen::Array2i> mask = cartographer::mapping::RayToPixelMask(
      Eigen::Array2i(bx, by), Eigen::Array2i(ex, ey), subpixel_scale);
  const int n = static_cast<int>(mask.size());
  for (int i = 0; i < n && i < capacity; ++i) {
    out_xy[2 * i] = mask[i].x();
    out_xy[2 * i + 1] = mask[i].y();
  }
  return n;
}

// kValueToProbability / kValueToCorrespondenceCost, 65536 entries each.
void ref_value_tables(float* value_to_probability, float* value_to_correspondence_cost) {
  for (int v = 0; v != 65536; ++v) {
    value_to_probability[v] = cartographer::mapping::ValueToProbability(static_cast<uint16_t>(v));
    value_to_correspondence_cost[v] =
        cartographer::mapping::ValueToCorrespondenceCost(static_cast<uint16_t>(v));
  }
}

int ref_probability_to_value(float p) { return cartographer::mapping::ProbabilityToValue(p); }
int ref_correspondence_cost_to_value(float c) {
  return cartographer::mapping::CorrespondenceCostToValue(c);
}

// ComputeLookupTableToApplyCorrespondenceCostOdds(Odds(probability)) / ...ApplyOdds, 32768 each.
void ref_odds_tables(float probability, uint16_t* correspondence_cost_table,
                     uint16_t* probability_table) {
  const float odds = cartographer::mapping::Odds(probability);
  const std::vector<uint16_t> cc =
      cartographer::mapping::ComputeLookupTableToApplyCorrespondenceCostOdds(odds);
  const std::vector<uint16_t> pr = cartographer::mapping::ComputeLookupTableToApplyOdds(odds);
  std::memcpy(correspondence_cost_table, cc.data(), 32768 * sizeof(uint16_t));
  std::memcpy(probability_table, pr.data(), 32768 * sizeof(uint16_t));
}

// ValueConversionTables::GetConversionTable(unknown_result, lower, upper), 65536 entries.
void ref_conversion_table(float unknown_result, float lower_bound, float upper_bound, float* out) {
  cartographer::mapping::ValueConversionTables tables;
  const std::vector<float>* t = tables.GetConversionTable(unknown_result, lower_bound, upper_bound);
  std::memcpy(out, t->data(), 65536 * sizeof(float));
}

// TSDValueConverter(max_tsd, max_weight): kind 0 = TSDToValue / ValueToTSD, 1 = weights.
int ref_tsd_float_to_value(int kind, float max_tsd, float max_weight, float x) {
  cartographer::mapping::ValueConversionTables tables;
  const cartographer::mapping::TSDValueConverter c(max_tsd, max_weight, &tables);
  return kind == 0 ? c.TSDToValue(x) : c.WeightToValue(x);
}
float ref_tsd_value_to_float(int kind, float max_tsd, float max_weight, int value) {
  cartographer::mapping::ValueConversionTables tables;
  const cartographer::mapping::TSDValueConverter c(max_tsd, max_weight, &tables);
  return kind == 0 ? c.ValueToTSD(static_cast<uint16_t>(value))
                   : c.ValueToWeight(static_cast<uint16_t>(value));
}

// FixedRatioSampler(ratio): the outcome of `count` consecutive Pulse() calls.
void ref_fixed_ratio_sampler(double ratio, int count, uint8_t* out) {
  cartographer::common::FixedRatioSampler sampler(ratio);
  for (int i = 0; i != count; ++i) out[i] = sampler.Pulse() ? 1 : 0;
}

}  // extern "C"

// ---- the 2D scan matchers: the reference's correlative_scan_matcher_2d.cc,
// fast_correlative_scan_matcher_2d.cc and real_time_correlative_scan_matcher_2d.cc, compiled
// unmodified against the stand-in data types of ref_shims/ (README there). ----------------------
namespace {

namespace cm = cartographer::mapping;
namespace sm = cartographer::mapping::scan_matching;

cartographer::sensor::PointCloud MakeCloud(const float* xyz, int n) {
  cartographer::sensor::PointCloud cloud;
  for (int i = 0; i != n; ++i)
    cloud.push_back({Eigen::Vector3f(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2])});
  return cloud;
}

struct RefFast2D {
  std::vector<uint16_t> cells;
  cm::ValueConversionTables tables;
  std::unique_ptr<cm::ProbabilityGrid> grid;
  std::unique_ptr<sm::FastCorrelativeScanMatcher2D> matcher;
  sm::proto::FastCorrelativeScanMatcherOptions2D options;
};

}  // namespace

extern "C" {

void* ref_fast2d_create(const uint16_t* cells, int nx, int ny, double resolution, double max_x,
                        double max_y, int depth, double linear_search_window,
                        double angular_search_window) {
  auto* f = new RefFast2D;
  f->cells.assign(cells, cells + static_cast<size_t>(nx) * ny);
  f->grid.reset(new cm::ProbabilityGrid(
      cm::MapLimits(resolution, Eigen::Vector2d(max_x, max_y), cm::CellLimits(nx, ny)),
      f->cells.data(), &f->tables));
  f->options.set_linear_search_window(linear_search_window);
  f->options.set_angular_search_window(angular_search_window);
  f->options.set_branch_and_bound_depth(depth);
  f->matcher.reset(new sm::FastCorrelativeScanMatcher2D(*f->grid, f->options));
  return f;
}
void ref_fast2d_destroy(void* h) { delete static_cast<RefFast2D*>(h); }

// Match / MatchFullSubmap: returns found; *score and pose_xyt[3] only written when found.
int ref_fast2d_match(void* h, int full_submap, const double* init_xyt, const float* xyz, int n,
                     float min_score, float* score, double* pose_xyt) {
  const RefFast2D* f = static_cast<RefFast2D*>(h);
  const cartographer::sensor::PointCloud cloud = MakeCloud(xyz, n);
  float s = 0.f;
  cartographer::transform::Rigid2d pose;
  const bool found =
      full_submap
          ? f->matcher->MatchFullSubmap(cloud, min_score, &s, &pose)
          : f->matcher->Match(cartographer::transform::Rigid2d({init_xyt[0], init_xyt[1]},
                                                               init_xyt[2]),
                              cloud, min_score, &s, &pose);
  if (found) {
    *score = s;
    pose_xyt[0] = pose.translation().x();
    pose_xyt[1] = pose.translation().y();
    pose_xyt[2] = pose.rotation().angle();
  }
  return found ? 1 : 0;
}

// PrecomputationGrid2D of `width` = 2^level: values for x in [-width+1, nx), y likewise,
// row-major (ny + width - 1) x (nx + width - 1) like the oracle's precompute2d.
void ref_precompute2d(const uint16_t* cells, int nx, int ny, int width, uint8_t* out) {
  cm::ValueConversionTables tables;
  const cm::ProbabilityGrid grid(cm::MapLimits(1., Eigen::Vector2d(0., 0.), cm::CellLimits(nx, ny)),
                                 cells, &tables);
  std::vector<float> reusable;
  const sm::PrecomputationGrid2D pre(grid, grid.limits().cell_limits(), width, &reusable);
  const int wx = nx + width - 1;
  for (int y = -width + 1; y < ny; ++y)
    for (int x = -width + 1; x < nx; ++x)
      out[static_cast<size_t>(y + width - 1) * wx + (x + width - 1)] =
          static_cast<uint8_t>(pre.GetValue(Eigen::Array2i(x, y)));
}

// RealTimeCorrelativeScanMatcher2D::Match on a probability grid, or on a TSDF2D when
// weight_cells != nullptr.
double ref_rt2d_match(const uint16_t* cells, const uint16_t* weight_cells, int nx, int ny,
                      double resolution, double max_x, double max_y, float truncation_distance,
                      float max_weight, const double* init_xyt, const float* xyz, int n, double lin,
                      double ang, double translation_weight, double rotation_weight,
                      double* pose_xyt) {
  cm::ValueConversionTables tables;
  const cm::MapLimits limits(resolution, Eigen::Vector2d(max_x, max_y), cm::CellLimits(nx, ny));
  std::unique_ptr<cm::Grid2D> grid;
  if (weight_cells) {
    grid.reset(new cm::TSDF2D(limits, cells, weight_cells, truncation_distance, max_weight, &tables));
  } else {
    grid.reset(new cm::ProbabilityGrid(limits, cells, &tables));
  }
  sm::proto::RealTimeCorrelativeScanMatcherOptions options;
  options.set_linear_search_window(lin);
  options.set_angular_search_window(ang);
  options.set_translation_delta_cost_weight(translation_weight);
  options.set_rotation_delta_cost_weight(rotation_weight);
  const sm::RealTimeCorrelativeScanMatcher2D matcher(options);
  cartographer::transform::Rigid2d pose;
  const double score = matcher.Match(
      cartographer::transform::Rigid2d({init_xyt[0], init_xyt[1]}, init_xyt[2]), MakeCloud(xyz, n),
      *grid, &pose);
  pose_xyt[0] = pose.translation().x();
  pose_xyt[1] = pose.translation().y();
  pose_xyt[2] = pose.rotation().angle();
  return score;
}

}  // extern "C"
