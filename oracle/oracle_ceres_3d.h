// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
// CeresScanMatcher3D::Match restated, probability grids only:
//   cartographer/mapping/internal/3d/scan_matching/ceres_scan_matcher_3d.cc:90-156
//   .../occupied_space_cost_function_3d.h:66-97, interpolated_grid.h:36-151
//   .../translation_delta_cost_functor_3d.h:43-50, rotation_delta_cost_functor_3d.h:43-55
//   mapping/internal/3d/rotation_parameterization.h:27-39 (YawOnlyQuaternionPlus)
// (IntensityCostFunction3D and its Huber loss are out of scope like IntensityHybridGrid.)
//
// PARITY UNPINNED against Ceres itself (third party, absent from /root/reference: ceres-solver
// at 58c5edae2f7c4d2533fe8a975c1f5f0b892dfd3e, bazel/repositories.bzl:134-144).  Restated from
// its published algorithm, as for 2D (oracle_ceres_2d.h), plus what 3D adds:
//   * the pose is two parameter blocks, translation[3] and rotation[4] = (w, x, y, z) with a
//     LocalParameterization: ceres::QuaternionParameterization (Plus(x, d) = (cos|d|,
//     sin|d| d/|d|) * x; its Jacobian at d = 0 is the 4 x 3 matrix of local_parameterization.cc)
//     or, with only_optimize_yaw, AutoDiffLocalParameterization<YawOnlyQuaternionPlus, 4, 1>
//     (Plus(x, d) = (sqrt(1 - c^2), 0, 0, c) * x with c = clamp(d, -0.5, 0.5); Jacobian at 0 =
//     d/dd, what the Jets compute);
//   * the minimizer works in the tangent space: Jacobians of the residual blocks w.r.t. the
//     seven ambient parameters are multiplied by the Plus-Jacobian (Program::Evaluate), Jacobi
//     scaling, LM and the step live in 6 (or 4) dimensions, the candidate is Plus(x, step);
//     step_norm and x_norm are ambient norms (trust_region_minimizer.cc);
//   * automatic differentiation replaced by closed-form derivatives of the same expressions:
//     Eigen's quaternion-times-vector formula v + w (2 u x v) + u x (2 u x v) differentiated as
//     written (NOT the derivative of a normalised rotation: Jets see the formula), the
//     piecewise cubic of InterpolatedGrid differentiated through normalized_{x,y,z}.
// What pins it: the reference's own CeresScanMatcher3DTest fixture and expectations
// (ceres_scan_matcher_3d_test.cc:36-128: five initial poses, pose within 3e-2, final cost
// within 1e-2 of 0) without the intensity term, and finite differences of the residuals
// (tests/test_ceres_3d.py).
#ifndef ORACLE_CERES_3D_H_
#define ORACLE_CERES_3D_H_

#include <vector>

#include "oracle_3d.h"
#include "oracle_ceres_2d.h"

namespace oracle {

struct CeresOptions3D {            // proto::CeresScanMatcherOptions3D
  std::vector<double> occupied_space_weight;   // one per (point cloud, grid) pair
  double translation_weight = 5.;
  double rotation_weight = 4e2;
  bool only_optimize_yaw = false;
  bool use_nonmonotonic_steps = false;
  int max_num_iterations = 12;
};

// IntensityHybridGrid read side (mapping/3d/hybrid_grid.h:543-571): AverageIntensityData
// {sum, count} per voxel, GetIntensity = sum / count (f32 / int) or 0 where nothing was added.
struct IntensityVoxel { int32_t x, y, z; int32_t count; float sum; };
class IntensityGridView {
 public:
  IntensityGridView(float resolution, const IntensityVoxel* voxels, int64_t n);
  float resolution() const { return resolution_; }
  Cell3i GetCellIndex(const V3f& p) const {      // HybridGridBase::GetCellIndex
    return {RoundToInt(p.x / resolution_), RoundToInt(p.y / resolution_),
            RoundToInt(p.z / resolution_)};
  }
  float GetIntensity(const Cell3i& c) const { return cells_.value(c.x, c.y, c.z); }
 private:
  float resolution_;
  Brick<float> cells_;     // the quotient as the reference evaluates it on every read
};

struct CloudAndGrid3D {            // CeresScanMatcher3D::PointCloudAndHybridGridsPointers
  const PointCloud3* point_cloud;
  const HybridGridView* hybrid_grid;
  // IntensityCostFunction3D (SM3/intensity_cost_function_3d.h, ceres_scan_matcher_3d.cc:118-137):
  // null grid = no intensity residual block for this pair (what ConstraintBuilder3D passes).
  const IntensityGridView* intensity_hybrid_grid = nullptr;
  const std::vector<float>* intensities = nullptr;        // PointCloud::intensities()
  double intensity_weight = 0., huber_scale = 0.;          // IntensityCostFunctionOptions
  float intensity_threshold = 0.f;
};

// InterpolatedGrid::GetInterpolatedValue at (x, y, z) and its gradient (interpolated_grid.h).
double InterpolatedProbability(const HybridGridView& grid, double x, double y, double z,
                               double gradient[3]);
double InterpolatedIntensity(const IntensityGridView& grid, double x, double y, double z,
                             double gradient[3]);

// Residual blocks in the order CeresScanMatcher3D::Match adds them: per pair its occupied-space
// block and (with an intensity grid) its intensity block, then translation, then rotation.
// `huber_a` > 0: the block carries ceres::HuberLoss(a) -- applied to the BLOCK's squared norm s
// (residual_block.cc): cost 1/2 rho(s), residuals and Jacobian scaled by sqrt(rho'(s))
// (corrector.cc with rho'' <= 0).  CeresResiduals3D returns the RAW residuals and Jacobian.
struct ResidualBlock3D { size_t begin, end; double huber_a; };
std::vector<ResidualBlock3D> CeresResidualBlocks3D(const std::vector<CloudAndGrid3D>& pairs);

// All residuals (sum of cloud sizes + 6) at translation[3], rotation[4] (w, x, y, z) and, when
// `jacobian` is non-null, their Jacobian w.r.t. the 7 ambient parameters (row-major, 7 columns).
void CeresResiduals3D(const CeresOptions3D& options, const double target_translation[3],
                      const double target_rotation[4], const std::vector<CloudAndGrid3D>& pairs,
                      const double translation[3], const double rotation[4],
                      std::vector<double>* residuals, std::vector<double>* jacobian);

// CeresScanMatcher3D::Match.
void CeresScanMatcher3DMatch(const CeresOptions3D& options, const double target_translation[3],
                             const Pose3d& initial_pose_estimate,
                             const std::vector<CloudAndGrid3D>& pairs, Pose3d* pose_estimate,
                             CeresSummary2D* summary);

}  // namespace oracle

#endif  // ORACLE_CERES_3D_H_
