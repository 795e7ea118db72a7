// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
// CeresScanMatcher2D::Match restated:
//   cartographer/mapping/internal/2d/scan_matching/ceres_scan_matcher_2d.cc:63-107
//   .../occupied_space_cost_function_2d.cc:39-108
//   .../translation_delta_cost_functor_2d.h:42-47, rotation_delta_cost_functor_2d.h:42-45
//
// PARITY UNPINNED against Ceres itself.  The least-squares solver is a third-party
// dependency absent from /root/reference (ceres-solver at 58c5edae2f7c4d2533fe8a975c1f5f0b892dfd3e,
// bazel/repositories.bzl:134-144); its published algorithm is restated here:
//   * ceres::BiCubicInterpolator / CubicHermiteSpline (include/ceres/cubic_interpolation.h):
//     Catmull-Rom splines over the 4x4 neighbourhood, rows first then the column;
//   * automatic differentiation replaced by the closed-form derivatives of the same
//     expressions (what Jets compute);
//   * TrustRegionMinimizer + LevenbergMarquardtStrategy (internal/ceres/trust_region_minimizer.cc,
//     levenberg_marquardt_strategy.cc, trust_region_step_evaluator.cc) with Solver::Options
//     defaults: initial radius 1e4, min_relative_decrease 1e-3, min / max LM diagonal 1e-6 / 1e32,
//     function / gradient / parameter tolerances 1e-6 / 1e-10 / 1e-8, Jacobi scaling from the
//     first Jacobian, radius update 1 / max(1/3, 1 - (2 rho - 1)^3), non-monotonic steps over a
//     window of 5 when requested.  The 3-unknown damped system is solved through its normal
//     equations (Ceres: DENSE_QR of the augmented Jacobian; same solution up to rounding).
// What pins it: the reference's own CeresScanMatcherTest (ceres_scan_matcher_2d_test.cc:34-112,
// four poses IsNearly to 1e-2, final cost ~0: tests/test_ceres_2d.py) and the reference's own
// occupied_space_cost_function_2d.cc, delta functors and ceres_scan_matcher_2d.cc compiled in
// place over the stand-in ceres/ headers of ref_shims (oracle/_ref/libref_ceres.so, `make
// ref_ceres`): residuals and Jacobians equal to 1e-12 on random poses, Match() to 1e-9 with the
// same step counts (tests/test_reference_ref_ceres.py).  The stand-in solver is ours too (Jets +
// Householder QR, written independently of this file): Ceres' own iterates stay unpinned.
#ifndef ORACLE_CERES_2D_H_
#define ORACLE_CERES_2D_H_

#include <vector>

#include "oracle_2d.h"

namespace oracle {

struct CeresOptions2D {            // proto::CeresScanMatcherOptions2D
  double occupied_space_weight = 1.;
  double translation_weight = 10.;
  double rotation_weight = 40.;
  bool use_nonmonotonic_steps = false;
  int max_num_iterations = 20;
};

enum CeresTermination { kCeresConvergence = 0, kCeresNoConvergence = 1, kCeresFailure = 2 };

struct CeresSummary2D {
  double initial_cost = 0., final_cost = 0.;
  int num_successful_steps = 0, num_unsuccessful_steps = 0;
  int termination = kCeresNoConvergence;
};

// Residuals (n + 3) and, when `jacobian` is non-null, the (n + 3) x 3 Jacobian (row-major)
// of the three residual blocks at `pose` = (x, y, theta).
void CeresResiduals2D(const CeresOptions2D& options, const double target_translation[2],
                      double target_angle, const PointCloud& cloud,
                      const ProbabilityGridView& grid, const double pose[3],
                      std::vector<double>* residuals, std::vector<double>* jacobian);

// CeresScanMatcher2D::Match.
void CeresScanMatcher2DMatch(const CeresOptions2D& options, const double target_translation[2],
                             const Pose2d& initial_pose_estimate, const PointCloud& cloud,
                             const ProbabilityGridView& grid, Pose2d* pose_estimate,
                             CeresSummary2D* summary);

}  // namespace oracle

#endif  // ORACLE_CERES_2D_H_
