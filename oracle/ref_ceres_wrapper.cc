// ORACLE — TEST INFRASTRUCTURE ONLY.  C wrapper around the reference's Ceres-side translation
// units, compiled unmodified from /root/reference by `make -C oracle ref_ceres` against the
// stand-in ceres/ headers of ref_shims/ (a small dense solver of ours with Ceres' interface:
// ref_shims/ceres/ceres.h says what of Ceres is restated there):
//   2d/scan_matching/occupied_space_cost_function_2d.cc, ceres_scan_matcher_2d.cc,
//   tsdf_match_cost_function_2d.cc, translation_/rotation_delta_cost_functor_2d.h,
//   3d/scan_matching/ceres_scan_matcher_3d.cc, occupied_space_cost_function_3d.h,
//   interpolated_grid.h, intensity_cost_function_3d.{h,cc}, translation_/rotation_delta_cost_functor_3d.h,
//   3d/rotation_parameterization.h, optimization/ceres_pose.cc, common/internal/ceres_solver_options.cc.
// Same argument layout as the orc_ceres* functions of oracle_capi.cc, so one Python caller
// drives the restatement and the reference.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "cartographer/mapping/2d/probability_grid.h"
#include "cartographer/mapping/3d/hybrid_grid.h"
#include "cartographer/mapping/internal/2d/scan_matching/ceres_scan_matcher_2d.h"
#include "cartographer/mapping/internal/2d/scan_matching/occupied_space_cost_function_2d.h"
#include "cartographer/mapping/internal/2d/scan_matching/rotation_delta_cost_functor_2d.h"
#include "cartographer/mapping/internal/2d/scan_matching/translation_delta_cost_functor_2d.h"
#include "cartographer/mapping/internal/3d/scan_matching/ceres_scan_matcher_3d.h"
#include "cartographer/mapping/internal/3d/scan_matching/intensity_cost_function_3d.h"
#include "cartographer/mapping/internal/3d/scan_matching/occupied_space_cost_function_3d.h"
#include "cartographer/mapping/internal/3d/scan_matching/rotation_delta_cost_functor_3d.h"
#include "cartographer/mapping/internal/3d/scan_matching/translation_delta_cost_functor_3d.h"
#include "cartographer/mapping/probability_values.h"
#include "cartographer/mapping/value_conversion_tables.h"

namespace {
namespace cm = cartographer::mapping;
namespace sm = cartographer::mapping::scan_matching;

cartographer::sensor::PointCloud MakeCloud(const float* xyz, int n, const float* intensities = nullptr) {
  std::vector<cartographer::sensor::RangefinderPoint> points;
  for (int i = 0; i != n; ++i)
    points.push_back({Eigen::Vector3f(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2])});
  if (intensities == nullptr) return cartographer::sensor::PointCloud(std::move(points));
  return cartographer::sensor::PointCloud(std::move(points),
                                          std::vector<float>(intensities, intensities + n));
}
std::unique_ptr<cm::ProbabilityGrid> MakeProbabilityGrid(const uint16_t* cells, int nx, int ny,
                                                         double resolution, double max_x,
                                                         double max_y,
                                                         cm::ValueConversionTables* tables) {
  cm::proto::Grid2D proto;
  *proto.mutable_limits() = cm::ToProto(
      cm::MapLimits(resolution, Eigen::Vector2d(max_x, max_y), cm::CellLimits(nx, ny)));
  proto.mutable_cells()->assign(cells, cells + static_cast<size_t>(nx) * ny);
  proto.set_min_correspondence_cost(cm::kMinCorrespondenceCost);
  proto.set_max_correspondence_cost(cm::kMaxCorrespondenceCost);
  proto.mutable_probability_grid_2d();
  return std::make_unique<cm::ProbabilityGrid>(proto, tables);
}
struct RefVoxelC { int32_t x, y, z; uint16_t value; uint16_t pad; };   // = oracle::Voxel
std::unique_ptr<cm::HybridGrid> MakeHybridGrid(float resolution, const RefVoxelC* v, int64_t n) {
  auto grid = std::make_unique<cm::HybridGrid>(resolution);
  for (int64_t i = 0; i != n; ++i)
    *grid->mutable_value(Eigen::Array3i(v[i].x, v[i].y, v[i].z)) = v[i].value;
  return grid;
}
// Intensity voxels: (x, y, z, count) + sum, as AverageIntensityData holds them.
struct RefIntensityVoxel { int32_t x, y, z; int32_t count; float sum; };
std::unique_ptr<cm::IntensityHybridGrid> MakeIntensityGrid(float resolution,
                                                           const RefIntensityVoxel* v, int64_t n) {
  auto grid = std::make_unique<cm::IntensityHybridGrid>(resolution);
  for (int64_t i = 0; i != n; ++i) {
    cm::AverageIntensityData* cell = grid->mutable_value(Eigen::Array3i(v[i].x, v[i].y, v[i].z));
    cell->count = v[i].count;
    cell->sum = v[i].sum;
  }
  return grid;
}
void Append(const ceres::CostFunction& f, const std::vector<const double*>& params,
            const std::vector<int>& column_of_block, int num_columns, std::vector<double>* r,
            std::vector<double>* J) {
  const int nr = f.num_residuals();
  std::vector<double> res(nr);
  std::vector<std::vector<double>> jac(params.size());
  std::vector<double*> jp(params.size());
  for (size_t b = 0; b != params.size(); ++b) {
    jac[b].assign(static_cast<size_t>(nr) * f.parameter_block_sizes()[b], 0.);
    jp[b] = jac[b].data();
  }
  f.Evaluate(params.data(), res.data(), jp.data());
  const size_t row0 = r->size();
  r->insert(r->end(), res.begin(), res.end());
  J->resize((row0 + nr) * num_columns, 0.);
  for (size_t b = 0; b != params.size(); ++b) {
    const int sz = f.parameter_block_sizes()[b];
    for (int k = 0; k != nr; ++k)
      for (int c = 0; c != sz; ++c)
        (*J)[(row0 + k) * num_columns + column_of_block[b] + c] = jac[b][k * sz + c];
  }
}
void StoreSummary(const ceres::Solver::Summary& s, double* summary5) {
  summary5[0] = s.initial_cost; summary5[1] = s.final_cost;
  // Ceres counts iteration 0 as a successful step; the restatement's summary does not.
  summary5[2] = s.num_successful_steps - 1; summary5[3] = s.num_unsuccessful_steps;
  summary5[4] = s.termination_type == ceres::CONVERGENCE ? 0
                : s.termination_type == ceres::NO_CONVERGENCE ? 1 : 2;
}
sm::proto::CeresScanMatcherOptions3D MakeOptions3D(const double* options8, int num_pairs) {
  sm::proto::CeresScanMatcherOptions3D o;
  o.set_translation_weight(options8[0]);
  o.set_rotation_weight(options8[1]);
  o.set_only_optimize_yaw(options8[2] != 0.);
  o.mutable_ceres_solver_options()->set_use_nonmonotonic_steps(options8[3] != 0.);
  o.mutable_ceres_solver_options()->set_max_num_iterations(static_cast<int>(options8[4]));
  o.mutable_ceres_solver_options()->set_num_threads(1);
  for (int k = 0; k != num_pairs; ++k) o.add_occupied_space_weight(options8[5 + k]);
  return o;
}
}  // namespace

extern "C" {

// options5 = occupied_space_weight, translation_weight, rotation_weight, use_nonmonotonic_steps,
// max_num_iterations; summary5 = initial_cost, final_cost, successful, unsuccessful, termination.
void refc_ceres2d_match(const uint16_t* cells, int nx, int ny, double res, double max_x,
                        double max_y, const double* options5, const double* target_xy,
                        const double* init_xyt, const float* xyz, int n, double* pose_xyt,
                        double* summary5) {
  cm::ValueConversionTables tables;
  const auto grid = MakeProbabilityGrid(cells, nx, ny, res, max_x, max_y, &tables);
  sm::proto::CeresScanMatcherOptions2D o;
  o.set_occupied_space_weight(options5[0]);
  o.set_translation_weight(options5[1]);
  o.set_rotation_weight(options5[2]);
  o.mutable_ceres_solver_options()->set_use_nonmonotonic_steps(options5[3] != 0.);
  o.mutable_ceres_solver_options()->set_max_num_iterations(static_cast<int>(options5[4]));
  o.mutable_ceres_solver_options()->set_num_threads(1);
  const sm::CeresScanMatcher2D matcher(o);
  const cartographer::sensor::PointCloud cloud = MakeCloud(xyz, n);
  cartographer::transform::Rigid2d pose;
  ceres::Solver::Summary summary;
  matcher.Match(Eigen::Vector2d(target_xy[0], target_xy[1]),
                cartographer::transform::Rigid2d({init_xyt[0], init_xyt[1]}, init_xyt[2]), cloud,
                *grid, &pose, &summary);
  pose_xyt[0] = pose.translation().x(); pose_xyt[1] = pose.translation().y();
  pose_xyt[2] = pose.rotation().angle();
  StoreSummary(summary, summary5);
}

// Residuals (n + 3) and Jacobian ((n + 3) x 3) of the three residual blocks
// CeresScanMatcher2D::Match builds (ceres_scan_matcher_2d.cc:74-103), evaluated at `pose`.
void refc_ceres2d_residuals(const uint16_t* cells, int nx, int ny, double res, double max_x,
                            double max_y, const double* options5, const double* target_xy,
                            double target_angle, const double* pose_xyt, const float* xyz, int n,
                            double* residuals, double* jacobian) {
  cm::ValueConversionTables tables;
  const auto grid = MakeProbabilityGrid(cells, nx, ny, res, max_x, max_y, &tables);
  const cartographer::sensor::PointCloud cloud = MakeCloud(xyz, n);
  std::vector<double> r, J;
  const std::vector<const double*> p{pose_xyt};
  std::unique_ptr<ceres::CostFunction> occupied(sm::CreateOccupiedSpaceCostFunction2D(
      options5[0] / std::sqrt(static_cast<double>(cloud.size())), cloud, *grid));
  Append(*occupied, p, {0}, 3, &r, &J);
  std::unique_ptr<ceres::CostFunction> translation(
      sm::TranslationDeltaCostFunctor2D::CreateAutoDiffCostFunction(
          options5[1], Eigen::Vector2d(target_xy[0], target_xy[1])));
  Append(*translation, p, {0}, 3, &r, &J);
  std::unique_ptr<ceres::CostFunction> rotation(
      sm::RotationDeltaCostFunctor2D::CreateAutoDiffCostFunction(options5[2], target_angle));
  Append(*rotation, p, {0}, 3, &r, &J);
  std::memcpy(residuals, r.data(), r.size() * sizeof(double));
  std::memcpy(jacobian, J.data(), J.size() * sizeof(double));
}

// options8 = translation_weight, rotation_weight, only_optimize_yaw, use_nonmonotonic_steps,
// max_num_iterations, occupied_space_weight[0..2].  intensity (optional, per pair; null pointers
// = no intensity grid): point intensities, intensity voxels, options3 = weight, huber_scale,
// intensity_threshold.
void refc_ceres3d_match(const double* options8, int num_pairs, const float* const* clouds,
                        const int* counts, const float* resolutions,
                        const RefVoxelC* const* voxels, const int64_t* voxel_counts,
                        const double* target_xyz, const double* init7, double* pose7,
                        double* summary5, const float* const* intensities,
                        const RefIntensityVoxel* const* intensity_voxels,
                        const int64_t* intensity_voxel_counts, const double* intensity_options3) {
  sm::proto::CeresScanMatcherOptions3D o = MakeOptions3D(options8, num_pairs);
  std::vector<cartographer::sensor::PointCloud> pcs;
  std::vector<std::unique_ptr<cm::HybridGrid>> grids;
  std::vector<std::unique_ptr<cm::IntensityHybridGrid>> igrids;
  for (int k = 0; k != num_pairs; ++k) {
    const bool has_intensity = intensities != nullptr && intensities[k] != nullptr;
    pcs.push_back(MakeCloud(clouds[k], counts[k], has_intensity ? intensities[k] : nullptr));
    grids.push_back(MakeHybridGrid(resolutions[k], voxels[k], voxel_counts[k]));
    igrids.push_back(has_intensity ? MakeIntensityGrid(resolutions[k], intensity_voxels[k],
                                                       intensity_voxel_counts[k])
                                   : nullptr);
    if (intensities != nullptr) {
      auto* io = o.add_intensity_cost_function_options();
      io->set_weight(intensity_options3[3 * k]);
      io->set_huber_scale(intensity_options3[3 * k + 1]);
      io->set_intensity_threshold(static_cast<float>(intensity_options3[3 * k + 2]));
    }
  }
  std::vector<sm::PointCloudAndHybridGridsPointers> pairs;
  for (int k = 0; k != num_pairs; ++k)
    pairs.push_back(sm::PointCloudAndHybridGridsPointers{&pcs[k], grids[k].get(), igrids[k].get()});
  const sm::CeresScanMatcher3D matcher(o);
  const cartographer::transform::Rigid3d init(
      Eigen::Vector3d(init7[0], init7[1], init7[2]),
      Eigen::Quaterniond(init7[3], init7[4], init7[5], init7[6]));
  cartographer::transform::Rigid3d pose;
  ceres::Solver::Summary summary;
  matcher.Match(Eigen::Vector3d(target_xyz[0], target_xyz[1], target_xyz[2]), init, pairs, &pose,
                &summary);
  pose7[0] = pose.translation().x(); pose7[1] = pose.translation().y();
  pose7[2] = pose.translation().z();
  pose7[3] = pose.rotation().w(); pose7[4] = pose.rotation().x(); pose7[5] = pose.rotation().y();
  pose7[6] = pose.rotation().z();
  StoreSummary(summary, summary5);
}

// residuals [sum(counts) + 6], jacobian [sum(counts) + 6][7] at pose7 (t, q = w x y z): the
// occupied-space blocks of every pair, then translation, then rotation (target target_q4).
void refc_ceres3d_residuals(const double* options8, int num_pairs, const float* const* clouds,
                            const int* counts, const float* resolutions,
                            const RefVoxelC* const* voxels, const int64_t* voxel_counts,
                            const double* target_xyz, const double* target_q4,
                            const double* pose7, double* residuals, double* jacobian) {
  std::vector<double> r, J;
  const std::vector<const double*> tq{pose7, pose7 + 3};
  for (int k = 0; k != num_pairs; ++k) {
    const cartographer::sensor::PointCloud cloud = MakeCloud(clouds[k], counts[k]);
    const auto grid = MakeHybridGrid(resolutions[k], voxels[k], voxel_counts[k]);
    std::unique_ptr<ceres::CostFunction> f(sm::OccupiedSpaceCostFunction3D::CreateAutoDiffCostFunction(
        options8[5 + k] / std::sqrt(static_cast<double>(cloud.size())), cloud, *grid));
    Append(*f, tq, {0, 3}, 7, &r, &J);
  }
  std::unique_ptr<ceres::CostFunction> translation(
      sm::TranslationDeltaCostFunctor3D::CreateAutoDiffCostFunction(
          options8[0], Eigen::Vector3d(target_xyz[0], target_xyz[1], target_xyz[2])));
  Append(*translation, {pose7}, {0}, 7, &r, &J);
  std::unique_ptr<ceres::CostFunction> rotation(
      sm::RotationDeltaCostFunctor3D::CreateAutoDiffCostFunction(
          options8[1], Eigen::Quaterniond(target_q4[0], target_q4[1], target_q4[2], target_q4[3])));
  Append(*rotation, {pose7 + 3}, {3}, 7, &r, &J);
  std::memcpy(residuals, r.data(), r.size() * sizeof(double));
  std::memcpy(jacobian, J.data(), J.size() * sizeof(double));
}

// IntensityCostFunction3D alone: residuals [n] and jacobian [n][7] at pose7, BEFORE the Huber
// loss (intensity_cost_function_3d.h:66-85).
void refc_intensity3d_residuals(double scaling_factor, float intensity_threshold, const float* xyz,
                                const float* intensities, int n, float resolution,
                                const RefIntensityVoxel* voxels, int64_t num_voxels,
                                const double* pose7, double* residuals, double* jacobian) {
  const cartographer::sensor::PointCloud cloud = MakeCloud(xyz, n, intensities);
  const auto grid = MakeIntensityGrid(resolution, voxels, num_voxels);
  std::unique_ptr<ceres::CostFunction> f(sm::IntensityCostFunction3D::CreateAutoDiffCostFunction(
      scaling_factor, intensity_threshold, cloud, *grid));
  std::vector<double> r, J;
  Append(*f, {pose7, pose7 + 3}, {0, 3}, 7, &r, &J);
  std::memcpy(residuals, r.data(), r.size() * sizeof(double));
  std::memcpy(jacobian, J.data(), J.size() * sizeof(double));
}

}  // extern "C"
