// ORACLE — TEST INFRASTRUCTURE ONLY.  C wrapper around the reference translation units that
// `make -C oracle ref` compiles unmodified from /root/reference (see ref_shims/README.md).
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "cartographer/common/fixed_ratio_sampler.h"
#include "cartographer/mapping/internal/2d/ray_to_pixel_mask.h"
#include "cartographer/mapping/2d/probability_grid.h"
#include "cartographer/mapping/2d/probability_grid_range_data_inserter_2d.h"
#include "cartographer/mapping/internal/2d/scan_matching/fast_correlative_scan_matcher_2d.h"
#include "cartographer/mapping/internal/2d/scan_matching/real_time_correlative_scan_matcher_2d.h"
#include "cartographer/mapping/3d/hybrid_grid.h"
#include "cartographer/mapping/3d/range_data_inserter_3d.h"
#include "cartographer/mapping/internal/3d/scan_matching/fast_correlative_scan_matcher_3d.h"
#include "cartographer/mapping/internal/3d/scan_matching/real_time_correlative_scan_matcher_3d.h"
#include "cartographer/mapping/internal/3d/scan_matching/rotational_scan_matcher.h"
#include "cartographer/mapping/internal/2d/tsd_value_converter.h"
#include "cartographer/mapping/internal/2d/tsdf_2d.h"
#include "cartographer/mapping/internal/2d/tsdf_range_data_inserter_2d.h"
#include "cartographer/mapping/probability_values.h"
#include "cartographer/mapping/value_conversion_tables.h"
#include "cartographer/sensor/internal/voxel_filter.h"

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

// Raw uint16 cells enter the REAL grid classes through their proto constructors
// (grid_2d.cc:77-98, probability_grid.cc:31-35, tsdf_2d.cc:35-47).
cm::proto::Grid2D MakeGridProto(const uint16_t* cells, int nx, int ny, double resolution,
                                double max_x, double max_y, float min_cc, float max_cc) {
  cm::proto::Grid2D proto;
  *proto.mutable_limits() = cm::ToProto(
      cm::MapLimits(resolution, Eigen::Vector2d(max_x, max_y), cm::CellLimits(nx, ny)));
  proto.mutable_cells()->assign(cells, cells + static_cast<size_t>(nx) * ny);
  proto.set_min_correspondence_cost(min_cc);
  proto.set_max_correspondence_cost(max_cc);
  return proto;
}
std::unique_ptr<cm::ProbabilityGrid> MakeProbabilityGrid(const uint16_t* cells, int nx, int ny,
                                                         double resolution, double max_x,
                                                         double max_y,
                                                         cm::ValueConversionTables* tables) {
  cm::proto::Grid2D proto = MakeGridProto(cells, nx, ny, resolution, max_x, max_y,
                                          cm::kMinCorrespondenceCost, cm::kMaxCorrespondenceCost);
  proto.mutable_probability_grid_2d();
  return std::make_unique<cm::ProbabilityGrid>(proto, tables);
}
std::unique_ptr<cm::TSDF2D> MakeTsdf(const uint16_t* cells, const uint16_t* weight_cells, int nx,
                                     int ny, double resolution, double max_x, double max_y,
                                     float truncation_distance, float max_weight,
                                     cm::ValueConversionTables* tables) {
  cm::proto::Grid2D proto = MakeGridProto(cells, nx, ny, resolution, max_x, max_y,
                                          -truncation_distance, truncation_distance);
  proto.mutable_tsdf_2d()->set_truncation_distance(truncation_distance);
  proto.mutable_tsdf_2d()->set_max_weight(max_weight);
  proto.mutable_tsdf_2d()->mutable_weight_cells()->assign(
      weight_cells, weight_cells + static_cast<size_t>(nx) * ny);
  return std::make_unique<cm::TSDF2D>(proto, tables);
}

struct RefFast2D {
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
  f->grid = MakeProbabilityGrid(cells, nx, ny, resolution, max_x, max_y, &f->tables);
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

// The same over the reference's TSDF2D (fast_correlative_scan_matcher_2d.cc:91-108 takes any
// Grid2D: 1 - |cost| with cost in [-truncation_distance, truncation_distance]).
void ref_precompute2d_tsdf(const uint16_t* cells, const uint16_t* weight_cells, int nx, int ny,
                           int width, float truncation_distance, float max_weight, uint8_t* out) {
  cm::ValueConversionTables tables;
  const auto grid = MakeTsdf(cells, weight_cells, nx, ny, 1., 0., 0., truncation_distance,
                             max_weight, &tables);
  std::vector<float> reusable;
  const sm::PrecomputationGrid2D pre(*grid, grid->limits().cell_limits(), width, &reusable);
  const int wx = nx + width - 1;
  for (int y = -width + 1; y < ny; ++y)
    for (int x = -width + 1; x < nx; ++x)
      out[static_cast<size_t>(y + width - 1) * wx + (x + width - 1)] =
          static_cast<uint8_t>(pre.GetValue(Eigen::Array2i(x, y)));
}

// PrecomputationGrid2D of `width` = 2^level: values for x in [-width+1, nx), y likewise,
// row-major (ny + width - 1) x (nx + width - 1) like the oracle's precompute2d.
void ref_precompute2d(const uint16_t* cells, int nx, int ny, int width, uint8_t* out) {
  cm::ValueConversionTables tables;
  const auto grid = MakeProbabilityGrid(cells, nx, ny, 1., 0., 0., &tables);
  std::vector<float> reusable;
  const sm::PrecomputationGrid2D pre(*grid, grid->limits().cell_limits(), width, &reusable);
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
  std::unique_ptr<cm::Grid2D> grid;
  if (weight_cells) {
    grid = MakeTsdf(cells, weight_cells, nx, ny, resolution, max_x, max_y, truncation_distance,
                    max_weight, &tables);
  } else {
    grid = MakeProbabilityGrid(cells, nx, ny, resolution, max_x, max_y, &tables);
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

// ---- the 2D grid and its range-data inserter: grid_2d.cc, probability_grid.cc and
// probability_grid_range_data_inserter_2d.cc compiled unmodified (GrowLimits, ApplyLookupTable,
// FinishUpdate, CastRays over the reference's own RayToPixelMask). ----------------------------
namespace {
struct RefGrid2D {
  cm::ValueConversionTables tables;
  std::unique_ptr<cm::ProbabilityGrid> grid;
};
}  // namespace

extern "C" {

void* ref_grid2d_create(double resolution, double max_x, double max_y, int nx, int ny,
                        const uint16_t* cells) {
  auto* g = new RefGrid2D;
  if (cells) {
    g->grid = MakeProbabilityGrid(cells, nx, ny, resolution, max_x, max_y, &g->tables);
  } else {
    g->grid = std::make_unique<cm::ProbabilityGrid>(
        cm::MapLimits(resolution, Eigen::Vector2d(max_x, max_y), cm::CellLimits(nx, ny)),
        &g->tables);
  }
  return g;
}
void ref_grid2d_destroy(void* h) { delete static_cast<RefGrid2D*>(h); }

// limits4: resolution, max_x, max_y; cells2: num_x_cells, num_y_cells.
void ref_grid2d_get_limits(void* h, double* limits3, int32_t* cells2) {
  const cm::MapLimits& l = static_cast<RefGrid2D*>(h)->grid->limits();
  limits3[0] = l.resolution(); limits3[1] = l.max().x(); limits3[2] = l.max().y();
  cells2[0] = l.cell_limits().num_x_cells; cells2[1] = l.cell_limits().num_y_cells;
}
// Raw cells through ToProto() (grid_2d.cc:167-183).
void ref_grid2d_download(void* h, uint16_t* out) {
  const cm::proto::Grid2D proto = static_cast<RefGrid2D*>(h)->grid->ToProto();
  for (int i = 0; i != proto.cells_size(); ++i) out[i] = static_cast<uint16_t>(proto.cells()[i]);
}
// ProbabilityGridRangeDataInserter2D::Insert (:123-132).  origin_xy and the points are in the
// map frame, like sensor::RangeData.
void ref_grid2d_insert(void* h, const float* origin_xy, const float* returns_xyz, int num_returns,
                       const float* misses_xyz, int num_misses, double hit_probability,
                       double miss_probability, int insert_free_space) {
  cm::proto::ProbabilityGridRangeDataInserterOptions2D options;
  options.set_hit_probability(hit_probability);
  options.set_miss_probability(miss_probability);
  options.set_insert_free_space(insert_free_space != 0);
  const cm::ProbabilityGridRangeDataInserter2D inserter(options);
  cartographer::sensor::RangeData range_data;
  range_data.origin = Eigen::Vector3f(origin_xy[0], origin_xy[1], 0.f);
  range_data.returns = MakeCloud(returns_xyz, num_returns);
  range_data.misses = MakeCloud(misses_xyz, num_misses);
  inserter.Insert(range_data, static_cast<RefGrid2D*>(h)->grid.get());
}
// Replaces the grid by ComputeCroppedGrid() (probability_grid.cc:90-106).
void ref_grid2d_crop(void* h) {
  auto* g = static_cast<RefGrid2D*>(h);
  std::unique_ptr<cm::Grid2D> cropped = g->grid->ComputeCroppedGrid();
  g->grid.reset(static_cast<cm::ProbabilityGrid*>(cropped.release()));
}
// SetProbability (probability_grid.cc:37-46; aborts on a known cell) / GetProbability (:78-83).
void ref_grid2d_set_probability(void* h, int ix, int iy, float probability) {
  static_cast<RefGrid2D*>(h)->grid->SetProbability(Eigen::Array2i(ix, iy), probability);
}
float ref_grid2d_get_probability(void* h, int ix, int iy) {
  return static_cast<RefGrid2D*>(h)->grid->GetProbability(Eigen::Array2i(ix, iy));
}
// MapLimits::GetCellIndex (map_limits.h:69-76) for n points: out (x, y) pairs.
void ref_map_limits_cell_index(double resolution, double max_x, double max_y, const float* xy,
                               int n, int32_t* out_xy) {
  const cm::MapLimits limits(resolution, Eigen::Vector2d(max_x, max_y), cm::CellLimits(1, 1));
  for (int i = 0; i != n; ++i) {
    const Eigen::Array2i c = limits.GetCellIndex(Eigen::Vector2f(xy[2 * i], xy[2 * i + 1]));
    out_xy[2 * i] = c.x(); out_xy[2 * i + 1] = c.y();
  }
}

}  // extern "C"

// ---- TSDF2D filled by the reference's own TSDFRangeDataInserter2D (tsdf_range_data_inserter_2d.cc,
// normal_estimation_2d.cc): what the reference's TSDF test fixtures are made with. --------------
namespace {
struct RefTsdf {
  cm::ValueConversionTables tables;
  std::unique_ptr<cm::TSDF2D> grid;
};
}  // namespace

extern "C" {

void* ref_tsdf_create(double resolution, double max_x, double max_y, int nx, int ny,
                      float truncation_distance, float max_weight) {
  auto* g = new RefTsdf;
  g->grid = std::make_unique<cm::TSDF2D>(
      cm::MapLimits(resolution, Eigen::Vector2d(max_x, max_y), cm::CellLimits(nx, ny)),
      truncation_distance, max_weight, &g->tables);
  return g;
}
void ref_tsdf_destroy(void* h) { delete static_cast<RefTsdf*>(h); }
void ref_tsdf_get_limits(void* h, double* limits3, int32_t* cells2) {
  const cm::MapLimits& l = static_cast<RefTsdf*>(h)->grid->limits();
  limits3[0] = l.resolution(); limits3[1] = l.max().x(); limits3[2] = l.max().y();
  cells2[0] = l.cell_limits().num_x_cells; cells2[1] = l.cell_limits().num_y_cells;
}
void ref_tsdf_download(void* h, uint16_t* tsd_cells, uint16_t* weight_cells) {
  const cm::proto::Grid2D proto = static_cast<RefTsdf*>(h)->grid->ToProto();
  for (int i = 0; i != proto.cells_size(); ++i) {
    tsd_cells[i] = static_cast<uint16_t>(proto.cells()[i]);
    weight_cells[i] = static_cast<uint16_t>(proto.tsdf_2d().weight_cells()[i]);
  }
}
// options8: truncation_distance, maximum_weight, update_free_space, num_normal_samples,
// sample_radius, project_sdf_distance_to_scan_normal, update_weight_range_exponent,
// angle kernel bandwidth, distance kernel bandwidth (9 values).
void ref_tsdf_insert(void* h, const float* origin_xyz, const float* returns_xyz, int num_returns,
                     const double* options9) {
  cm::proto::TSDFRangeDataInserterOptions2D o;
  o.set_truncation_distance(options9[0]);
  o.set_maximum_weight(options9[1]);
  o.set_update_free_space(options9[2] != 0.);
  o.mutable_normal_estimation_options()->set_num_normal_samples(static_cast<int>(options9[3]));
  o.mutable_normal_estimation_options()->set_sample_radius(options9[4]);
  o.set_project_sdf_distance_to_scan_normal(options9[5] != 0.);
  o.set_update_weight_range_exponent(static_cast<int>(options9[6]));
  o.set_update_weight_angle_scan_normal_to_ray_kernel_bandwidth(options9[7]);
  o.set_update_weight_distance_cell_to_hit_kernel_bandwidth(options9[8]);
  const cm::TSDFRangeDataInserter2D inserter(o);
  cartographer::sensor::RangeData range_data;
  range_data.origin = Eigen::Vector3f(origin_xyz[0], origin_xyz[1], origin_xyz[2]);
  range_data.returns = MakeCloud(returns_xyz, num_returns);
  inserter.Insert(range_data, static_cast<RefTsdf*>(h)->grid.get());
}

}  // extern "C"

// ---- the 3D scan matchers: real_time_correlative_scan_matcher_3d.cc, precomputation_grid_3d.cc,
// rotational_scan_matcher.cc, low_resolution_matcher.cc, fast_correlative_scan_matcher_3d.cc and
// the header-only mapping/3d/hybrid_grid.h, all compiled unmodified.  Same argument layout as the
// orc_*3d functions of oracle_capi.cc so that one Python caller drives both. --------------------
namespace {

struct RefVoxel { int32_t x, y, z; uint16_t value; uint16_t pad; };   // = oracle::Voxel

std::unique_ptr<cm::HybridGrid> MakeHybridGrid(float resolution, const RefVoxel* voxels,
                                               int64_t n) {
  auto grid = std::make_unique<cm::HybridGrid>(resolution);
  for (int64_t i = 0; i != n; ++i)
    *grid->mutable_value(Eigen::Array3i(voxels[i].x, voxels[i].y, voxels[i].z)) = voxels[i].value;
  return grid;
}

cartographer::transform::Rigid3d MakeRigid3d(const double* p7) {   // t xyz, q wxyz
  return cartographer::transform::Rigid3d(Eigen::Vector3d(p7[0], p7[1], p7[2]),
                                          Eigen::Quaterniond(p7[3], p7[4], p7[5], p7[6]));
}
void StoreRigid3d(const cartographer::transform::Rigid3d& p, double* out7) {
  out7[0] = p.translation().x(); out7[1] = p.translation().y(); out7[2] = p.translation().z();
  out7[3] = p.rotation().w(); out7[4] = p.rotation().x(); out7[5] = p.rotation().y();
  out7[6] = p.rotation().z();
}
Eigen::VectorXf MakeHistogram(const float* h, int n) {
  Eigen::VectorXf v = Eigen::VectorXf::Zero(n);
  for (int i = 0; i != n; ++i) v[i] = h[i];
  return v;
}
sm::proto::FastCorrelativeScanMatcherOptions3D MakeOptions3D(int depth, int full_resolution_depth,
                                                             double min_rotational_score,
                                                             double min_low_resolution_score,
                                                             double lin_xy, double lin_z,
                                                             double ang) {
  sm::proto::FastCorrelativeScanMatcherOptions3D o;
  o.set_branch_and_bound_depth(depth);
  o.set_full_resolution_depth(full_resolution_depth);
  o.set_min_rotational_score(min_rotational_score);
  o.set_min_low_resolution_score(min_low_resolution_score);
  o.set_linear_xy_search_window(lin_xy);
  o.set_linear_z_search_window(lin_z);
  o.set_angular_search_window(ang);
  return o;
}

struct RefFast3D {
  std::unique_ptr<cm::HybridGrid> grid, low_grid;
  Eigen::VectorXf histogram;
  sm::proto::FastCorrelativeScanMatcherOptions3D options;
  std::unique_ptr<sm::FastCorrelativeScanMatcher3D> matcher;
  std::unique_ptr<sm::PrecomputationGridStack3D> stack;   // built on demand for level dumps
};

}  // namespace

extern "C" {

// DynamicGrid::grid_size() after every voxel has been written (Grow(), hybrid_grid.h:381-398).
int ref_grid3d_size(float resolution, const void* voxels, int64_t n) {
  return MakeHybridGrid(resolution, static_cast<const RefVoxel*>(voxels), n)->grid_size();
}

// Round trip through the real HybridGrid: its iterator's (cell index, value) pairs in iteration
// order; returns the count, writes min(count, capacity) rows of (x, y, z, value).
int64_t ref_grid3d_iterate(float resolution, const void* voxels, int64_t n, int32_t* out_xyzv,
                           int64_t capacity) {
  const auto grid = MakeHybridGrid(resolution, static_cast<const RefVoxel*>(voxels), n);
  int64_t k = 0;
  for (auto it = cm::HybridGrid::Iterator(*grid); !it.Done(); it.Next(), ++k) {
    if (k >= capacity) continue;
    const Eigen::Array3i c = it.GetCellIndex();
    out_xyzv[4 * k] = c.x(); out_xyzv[4 * k + 1] = c.y(); out_xyzv[4 * k + 2] = c.z();
    out_xyzv[4 * k + 3] = it.GetValue();
  }
  return k;
}

// HybridGridBase::GetCellIndex (hybrid_grid.h:428-433) for n points.
void ref_grid3d_cell_index(float resolution, const float* xyz, int n, int32_t* out_xyz) {
  const cm::HybridGrid grid(resolution);
  for (int i = 0; i != n; ++i) {
    const Eigen::Array3i c =
        grid.GetCellIndex(Eigen::Vector3f(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
    out_xyz[3 * i] = c.x(); out_xyz[3 * i + 1] = c.y(); out_xyz[3 * i + 2] = c.z();
  }
}

// A mutable HybridGrid driven by the reference's own RangeDataInserter3D
// (range_data_inserter_3d.cc:27-114).
void* ref_hgrid_create(float resolution) { return new cm::HybridGrid(resolution); }
void ref_hgrid_destroy(void* h) { delete static_cast<cm::HybridGrid*>(h); }
int ref_hgrid_size(void* h) { return static_cast<cm::HybridGrid*>(h)->grid_size(); }
void ref_hgrid_set_probability(void* h, int x, int y, int z, float probability) {
  static_cast<cm::HybridGrid*>(h)->SetProbability(Eigen::Array3i(x, y, z), probability);
}
float ref_hgrid_get_probability(void* h, int x, int y, int z) {
  return static_cast<cm::HybridGrid*>(h)->GetProbability(Eigen::Array3i(x, y, z));
}
void ref_hgrid_insert(void* h, const float* origin_xyz, const float* returns_xyz, int num_returns,
                      double hit_probability, double miss_probability,
                      int num_free_space_voxels) {
  cm::proto::RangeDataInserterOptions3D options;
  options.set_hit_probability(hit_probability);
  options.set_miss_probability(miss_probability);
  options.set_num_free_space_voxels(num_free_space_voxels);
  const cm::RangeDataInserter3D inserter(options);
  cartographer::sensor::RangeData range_data;
  range_data.origin = Eigen::Vector3f(origin_xyz[0], origin_xyz[1], origin_xyz[2]);
  range_data.returns = MakeCloud(returns_xyz, num_returns);
  inserter.Insert(range_data, static_cast<cm::HybridGrid*>(h), nullptr);
}
}  // extern "C" (reopened below)
// (the reference never iterates an IntensityHybridGrid, so AverageIntensityData has no
// operator== for IsDefaultValue, hybrid_grid.h:55-58; found by ADL at instantiation)
namespace cartographer {
namespace mapping {
inline bool operator==(const AverageIntensityData& a, const AverageIntensityData& b) {
  return a.sum == b.sum && a.count == b.count;
}
}  // namespace mapping
}  // namespace cartographer
extern "C" {
// The reference's IntensityHybridGrid (hybrid_grid.h:543-571) filled by its own
// RangeDataInserter3D::Insert with an intensity grid (range_data_inserter_3d.cc:54-70,109-112).
void* ref_igrid_create(float resolution) { return new cm::IntensityHybridGrid(resolution); }
void ref_igrid_destroy(void* h) { delete static_cast<cm::IntensityHybridGrid*>(h); }
void ref_hgrid_insert_with_intensities(void* h, void* ih, const float* origin_xyz,
                                       const float* returns_xyz, const float* intensities,
                                       int num_returns, double hit_probability,
                                       double miss_probability, int num_free_space_voxels,
                                       double intensity_threshold) {
  cm::proto::RangeDataInserterOptions3D options;
  options.set_hit_probability(hit_probability);
  options.set_miss_probability(miss_probability);
  options.set_num_free_space_voxels(num_free_space_voxels);
  options.set_intensity_threshold(intensity_threshold);
  const cm::RangeDataInserter3D inserter(options);
  cartographer::sensor::RangeData range_data;
  range_data.origin = Eigen::Vector3f(origin_xyz[0], origin_xyz[1], origin_xyz[2]);
  std::vector<cartographer::sensor::RangefinderPoint> points;
  for (int i = 0; i != num_returns; ++i)
    points.push_back({Eigen::Vector3f(returns_xyz[3 * i], returns_xyz[3 * i + 1], returns_xyz[3 * i + 2])});
  range_data.returns =
      intensities ? cartographer::sensor::PointCloud(
                        points, std::vector<float>(intensities, intensities + num_returns))
                  : cartographer::sensor::PointCloud(points);
  inserter.Insert(range_data, static_cast<cm::HybridGrid*>(h),
                  static_cast<cm::IntensityHybridGrid*>(ih));
}
// (x, y, z, count) rows + sums in iteration order; returns the count, writes min(count, capacity).
int64_t ref_igrid_voxels(void* ih, int32_t* out_xyzc, float* out_sum, int64_t capacity) {
  const auto* grid = static_cast<cm::IntensityHybridGrid*>(ih);
  int64_t k = 0;
  for (auto it = cm::IntensityHybridGrid::Iterator(*grid); !it.Done(); it.Next(), ++k) {
    if (k >= capacity) continue;
    const Eigen::Array3i c = it.GetCellIndex();
    out_xyzc[4 * k] = c.x(); out_xyzc[4 * k + 1] = c.y(); out_xyzc[4 * k + 2] = c.z();
    out_xyzc[4 * k + 3] = it.GetValue().count;
    out_sum[k] = it.GetValue().sum;
  }
  return k;
}

// (x, y, z, value) rows in iteration order; returns the count, writes min(count, capacity).
int64_t ref_hgrid_voxels(void* h, int32_t* out_xyzv, int64_t capacity) {
  const auto* grid = static_cast<cm::HybridGrid*>(h);
  int64_t k = 0;
  for (auto it = cm::HybridGrid::Iterator(*grid); !it.Done(); it.Next(), ++k) {
    if (k >= capacity) continue;
    const Eigen::Array3i c = it.GetCellIndex();
    out_xyzv[4 * k] = c.x(); out_xyzv[4 * k + 1] = c.y(); out_xyzv[4 * k + 2] = c.z();
    out_xyzv[4 * k + 3] = it.GetValue();
  }
  return k;
}

float ref_rt3d_match(float resolution, const void* voxels, int64_t n, const double* init7,
                     const float* xyz, int npts, double lin, double ang, double tw, double rw,
                     double* pose7, int64_t* num_candidates) {
  const auto grid = MakeHybridGrid(resolution, static_cast<const RefVoxel*>(voxels), n);
  sm::proto::RealTimeCorrelativeScanMatcherOptions options;
  options.set_linear_search_window(lin);
  options.set_angular_search_window(ang);
  options.set_translation_delta_cost_weight(tw);
  options.set_rotation_delta_cost_weight(rw);
  const sm::RealTimeCorrelativeScanMatcher3D matcher(options);
  cartographer::transform::Rigid3d pose = MakeRigid3d(init7);
  const float score = matcher.Match(MakeRigid3d(init7), MakeCloud(xyz, npts), *grid, &pose);
  StoreRigid3d(pose, pose7);
  if (num_candidates) *num_candidates = -1;   // the reference does not count them
  return score;
}

void ref_rotational_match(const float* submap_hist, const float* scan_hist, int size,
                          float initial_angle, const float* angles, int n, float* out) {
  const Eigen::VectorXf submap = MakeHistogram(submap_hist, size);
  const sm::RotationalScanMatcher matcher(&submap);
  const std::vector<float> r = matcher.Match(MakeHistogram(scan_hist, size), initial_angle,
                                             std::vector<float>(angles, angles + n));
  std::memcpy(out, r.data(), n * sizeof(float));
}

// RotationalScanMatcher::ComputeHistogram (rotational_scan_matcher.cc:164-177).
void ref_compute_histogram(const float* xyz, int n, int histogram_size, float* out) {
  const Eigen::VectorXf h = sm::RotationalScanMatcher::ComputeHistogram(MakeCloud(xyz, n),
                                                                        histogram_size);
  for (int i = 0; i != histogram_size; ++i) out[i] = h[i];
}

void* ref_fast3d_create(float resolution, const void* voxels, int64_t n, float low_resolution,
                        const void* low_voxels, int64_t nlow, const float* hist, int nh, int depth,
                        int full_resolution_depth, double min_rotational_score,
                        double min_low_resolution_score, double lin_xy, double lin_z, double ang) {
  auto* f = new RefFast3D;
  f->grid = MakeHybridGrid(resolution, static_cast<const RefVoxel*>(voxels), n);
  f->low_grid = MakeHybridGrid(low_resolution, static_cast<const RefVoxel*>(low_voxels), nlow);
  f->histogram = MakeHistogram(hist, nh);
  f->options = MakeOptions3D(depth, full_resolution_depth, min_rotational_score,
                             min_low_resolution_score, lin_xy, lin_z, ang);
  f->matcher.reset(new sm::FastCorrelativeScanMatcher3D(*f->grid, f->low_grid.get(),
                                                        &f->histogram, f->options));
  return f;
}
void ref_fast3d_destroy(void* h) { delete static_cast<RefFast3D*>(h); }

namespace {
const sm::PrecomputationGrid3D& Level(void* h, int depth) {
  auto* f = static_cast<RefFast3D*>(h);
  if (!f->stack) f->stack.reset(new sm::PrecomputationGridStack3D(*f->grid, f->options));
  return f->stack->Get(depth);
}
}  // namespace

int64_t ref_fast3d_level_count(void* h, int depth) {
  int64_t c = 0;
  for (auto it = sm::PrecomputationGrid3D::Iterator(Level(h, depth)); !it.Done(); it.Next()) ++c;
  return c;
}
void ref_fast3d_level_voxels(void* h, int depth, int* out_xyzv) {   // iteration order
  int64_t k = 0;
  for (auto it = sm::PrecomputationGrid3D::Iterator(Level(h, depth)); !it.Done();
       it.Next(), ++k) {
    const Eigen::Array3i c = it.GetCellIndex();
    out_xyzv[4 * k] = c.x(); out_xyzv[4 * k + 1] = c.y(); out_xyzv[4 * k + 2] = c.z();
    out_xyzv[4 * k + 3] = it.GetValue();
  }
}

int ref_fast3d_match(void* h, int full_submap, const double* node7, const double* submap7,
                     const double* gravity_wxyz, const float* hi, int nhi, const float* lo, int nlo,
                     const float* hist, int nh, float min_score, double* result10,
                     int64_t* stats4) {
  const auto& m = *static_cast<RefFast3D*>(h)->matcher;
  cm::TrajectoryNode::Data data;
  data.gravity_alignment =
      Eigen::Quaterniond(gravity_wxyz[0], gravity_wxyz[1], gravity_wxyz[2], gravity_wxyz[3]);
  data.high_resolution_point_cloud = MakeCloud(hi, nhi);
  data.low_resolution_point_cloud = MakeCloud(lo, nlo);
  data.rotational_scan_matcher_histogram = MakeHistogram(hist, nh);
  std::unique_ptr<sm::FastCorrelativeScanMatcher3D::Result> r;
  if (full_submap) {
    r = m.MatchFullSubmap(Eigen::Quaterniond(node7[3], node7[4], node7[5], node7[6]),
                          Eigen::Quaterniond(submap7[3], submap7[4], submap7[5], submap7[6]), data,
                          min_score);
  } else {
    r = m.Match(MakeRigid3d(node7), MakeRigid3d(submap7), data, min_score);
  }
  if (r != nullptr) {
    result10[0] = r->score;
    StoreRigid3d(r->pose_estimate, result10 + 1);
    result10[8] = r->rotational_score;
    result10[9] = r->low_resolution_score;
  }
  if (stats4) stats4[0] = stats4[1] = stats4[2] = stats4[3] = -1;   // not counted by the reference
  return r != nullptr ? 1 : 0;
}


// sensor::VoxelFilter(PointCloud, resolution) (sensor/internal/voxel_filter.cc:132-152): the
// reference's own randomised reservoir filter; `out_xyz` receives the kept points in order.
int ref_voxel_filter(const float* xyz, int n, float resolution, float* out_xyz) {
  const cartographer::sensor::PointCloud r =
      cartographer::sensor::VoxelFilter(MakeCloud(xyz, n), resolution);
  for (size_t i = 0; i != r.size(); ++i) {
    out_xyz[3 * i] = r[i].position.x();
    out_xyz[3 * i + 1] = r[i].position.y();
    out_xyz[3 * i + 2] = r[i].position.z();
  }
  return static_cast<int>(r.size());
}
// sensor::AdaptiveVoxelFilter (voxel_filter.cc:193-198).
int ref_adaptive_voxel_filter(const float* xyz, int n, float max_length, float min_num_points,
                              float max_range, float* out_xyz) {
  cartographer::sensor::proto::AdaptiveVoxelFilterOptions options;
  options.set_max_length(max_length);
  options.set_min_num_points(min_num_points);
  options.set_max_range(max_range);
  const cartographer::sensor::PointCloud r =
      cartographer::sensor::AdaptiveVoxelFilter(MakeCloud(xyz, n), options);
  for (size_t i = 0; i != r.size(); ++i) {
    out_xyz[3 * i] = r[i].position.x();
    out_xyz[3 * i + 1] = r[i].position.y();
    out_xyz[3 * i + 2] = r[i].position.z();
  }
  return static_cast<int>(r.size());
}

}  // extern "C"
