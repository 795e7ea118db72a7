// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_ceres_2d.h for what is restated and what
// pins it).
#include "oracle_ceres_2d.h"

#include <algorithm>
#include <climits>
#include <cmath>

namespace oracle {
namespace {

constexpr int kPadding = INT_MAX / 4;   // occupied_space_cost_function_2d.cc:78

// GridArrayAdapter::GetValue (occupied_space_cost_function_2d.cc:85-93).
double AdapterValue(const ProbabilityGridView& grid, int row, int column) {
  const int rows = grid.limits.num_y_cells + 2 * kPadding;
  const int cols = grid.limits.num_x_cells + 2 * kPadding;
  if (row < kPadding || column < kPadding || row >= rows - kPadding || column >= cols - kPadding)
    return static_cast<double>(kMaxCorrespondenceCost);
  return static_cast<double>(grid.GetCorrespondenceCost(Cell2i{column - kPadding, row - kPadding}));
}

// ceres::CubicHermiteSpline<1> (cubic_interpolation.h): Catmull-Rom through p1, p2.
void CubicHermiteSpline(double p0, double p1, double p2, double p3, double x, double* f,
                        double* dfdx) {
  const double a = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3);
  const double b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3);
  const double c = 0.5 * (-p0 + p2);
  const double d = p1;
  if (f) *f = d + x * (c + x * (b + x * a));
  if (dfdx) *dfdx = c + x * (2.0 * b + 3.0 * a * x);
}

// ceres::BiCubicInterpolator::Evaluate(r, c, f, dfdr, dfdc).
void BiCubic(const ProbabilityGridView& grid, double r, double c, double* f, double* dfdr,
             double* dfdc) {
  const int row = static_cast<int>(std::floor(r));
  const int col = static_cast<int>(std::floor(c));
  double fr[4], dfr[4];
  for (int k = 0; k < 4; ++k) {
    const int rr = row - 1 + k;
    CubicHermiteSpline(AdapterValue(grid, rr, col - 1), AdapterValue(grid, rr, col),
                       AdapterValue(grid, rr, col + 1), AdapterValue(grid, rr, col + 2), c - col,
                       &fr[k], &dfr[k]);
  }
  CubicHermiteSpline(fr[0], fr[1], fr[2], fr[3], r - row, f, dfdr);
  CubicHermiteSpline(dfr[0], dfr[1], dfr[2], dfr[3], r - row, dfdc, nullptr);
}

// Solves the symmetric positive definite 3x3 system A x = b (Cholesky); false if not SPD.
bool SolveSpd3(const double A[3][3], const double b[3], double x[3]) {
  double L[3][3] = {{0}};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j <= i; ++j) {
      double sum = A[i][j];
      for (int k = 0; k < j; ++k) sum -= L[i][k] * L[j][k];
      if (i == j) {
        if (!(sum > 0.)) return false;
        L[i][i] = std::sqrt(sum);
      } else {
        L[i][j] = sum / L[j][j];
      }
    }
  }
  double y[3];
  for (int i = 0; i < 3; ++i) {
    double sum = b[i];
    for (int k = 0; k < i; ++k) sum -= L[i][k] * y[k];
    y[i] = sum / L[i][i];
  }
  for (int i = 2; i >= 0; --i) {
    double sum = y[i];
    for (int k = i + 1; k < 3; ++k) sum -= L[k][i] * x[k];
    x[i] = sum / L[i][i];
  }
  return std::isfinite(x[0]) && std::isfinite(x[1]) && std::isfinite(x[2]);
}

// What a pass over the residual blocks leaves behind for the minimizer: cost = 1/2 |r|^2,
// gradient g = J^T r and H = J^T J (3 x 3), all of the UNSCALED Jacobian.
struct Evaluation {
  double cost = 0.;
  double g[3] = {0, 0, 0};
  double H[3][3] = {{0}};
};

}  // namespace

void CeresResiduals2D(const CeresOptions2D& options, const double target_translation[2],
                      double target_angle, const PointCloud& cloud,
                      const ProbabilityGridView& grid, const double pose[3],
                      std::vector<double>* residuals, std::vector<double>* jacobian) {
  const size_t n = cloud.size();
  residuals->assign(n + 3, 0.);
  if (jacobian) jacobian->assign(3 * (n + 3), 0.);
  // occupied_space_weight / sqrt(point_cloud.size())  (ceres_scan_matcher_2d.cc:78-80)
  const double scaling = options.occupied_space_weight / std::sqrt(static_cast<double>(n));
  const double c = std::cos(pose[2]), s = std::sin(pose[2]);
  const double res = grid.limits.resolution;
  for (size_t i = 0; i < n; ++i) {
    // transform * (x, y, 1): rotation | translation (occupied_space_cost_function_2d.cc:48-62)
    const double px = static_cast<double>(cloud[i].x), py = static_cast<double>(cloud[i].y);
    const double wx = c * px + -s * py + pose[0] * 1.;
    const double wy = s * px + c * py + pose[1] * 1.;
    const double r = (grid.limits.max_x - wx) / res - 0.5 + static_cast<double>(kPadding);
    const double cc = (grid.limits.max_y - wy) / res - 0.5 + static_cast<double>(kPadding);
    double f, dfdr, dfdc;
    BiCubic(grid, r, cc, &f, &dfdr, &dfdc);
    (*residuals)[i] = scaling * f;
    if (jacobian) {
      const double dwx_dt = -s * px - c * py, dwy_dt = c * px - s * py;
      double* row = jacobian->data() + 3 * i;
      row[0] = scaling * (dfdr * (-1. / res));
      row[1] = scaling * (dfdc * (-1. / res));
      row[2] = scaling * (dfdr * (-dwx_dt / res) + dfdc * (-dwy_dt / res));
    }
  }
  // TranslationDeltaCostFunctor2D, RotationDeltaCostFunctor2D.
  (*residuals)[n] = options.translation_weight * (pose[0] - target_translation[0]);
  (*residuals)[n + 1] = options.translation_weight * (pose[1] - target_translation[1]);
  (*residuals)[n + 2] = options.rotation_weight * (pose[2] - target_angle);
  if (jacobian) {
    (*jacobian)[3 * n + 0] = options.translation_weight;
    (*jacobian)[3 * (n + 1) + 1] = options.translation_weight;
    (*jacobian)[3 * (n + 2) + 2] = options.rotation_weight;
  }
}

void CeresScanMatcher2DMatch(const CeresOptions2D& options, const double target_translation[2],
                             const Pose2d& initial_pose_estimate, const PointCloud& cloud,
                             const ProbabilityGridView& grid, Pose2d* pose_estimate,
                             CeresSummary2D* summary) {
  // Solver::Options defaults of the pinned Ceres (include/ceres/solver.h).
  const double kFunctionTolerance = 1e-6, kGradientTolerance = 1e-10, kParameterTolerance = 1e-8;
  const double kMinRelativeDecrease = 1e-3, kMinLmDiagonal = 1e-6, kMaxLmDiagonal = 1e32;
  const double kMaxRadius = 1e16, kMinRadius = 1e-32;
  const int kMaxConsecutiveInvalidSteps = 5;
  const int max_consecutive_nonmonotonic_steps = options.use_nonmonotonic_steps ? 5 : 0;

  const double target_angle = initial_pose_estimate.theta;
  std::vector<double> r, J;
  auto evaluate = [&](const double x[3], bool with_jacobian, Evaluation* e) {
    CeresResiduals2D(options, target_translation, target_angle, cloud, grid, x, &r,
                     with_jacobian ? &J : nullptr);
    *e = Evaluation();
    for (size_t i = 0; i < r.size(); ++i) {
      e->cost += r[i] * r[i];
      if (with_jacobian) {
        const double* row = J.data() + 3 * i;
        for (int a = 0; a < 3; ++a) {
          e->g[a] += row[a] * r[i];
          for (int b = 0; b < 3; ++b) e->H[a][b] += row[a] * row[b];
        }
      }
    }
    e->cost *= 0.5;
  };

  double x[3] = {initial_pose_estimate.x, initial_pose_estimate.y, initial_pose_estimate.theta};
  Evaluation at_x;
  evaluate(x, true, &at_x);
  CeresSummary2D local;
  CeresSummary2D& sum = summary ? *summary : local;
  sum = CeresSummary2D();
  sum.initial_cost = at_x.cost;
  // Jacobi scaling from the first Jacobian: 1 / (1 + ||column||).
  double scale[3];
  for (int a = 0; a < 3; ++a) scale[a] = 1. / (1. + std::sqrt(at_x.H[a][a]));
  double x_cost = at_x.cost;
  double x_norm = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  auto gradient_max_norm = [](const Evaluation& e) {
    return std::max(std::fabs(e.g[0]), std::max(std::fabs(e.g[1]), std::fabs(e.g[2])));
  };

  // Levenberg-Marquardt strategy state.
  double radius = 1e4, decrease_factor = 2.;
  bool reuse_diagonal = false;
  double diagonal[3] = {0, 0, 0};
  // TrustRegionStepEvaluator state (trust_region_step_evaluator.cc).
  double minimum_cost = x_cost, current_cost = x_cost, reference_cost = x_cost,
         candidate_cost_eval = x_cost;
  double accumulated_reference_model_cost_change = 0., accumulated_candidate_model_cost_change = 0.;
  int num_consecutive_nonmonotonic_steps = 0;
  int num_consecutive_invalid_steps = 0;

  // The parameters handed back are those of the lowest cost seen (they differ from the last
  // iterate only with non-monotonic steps): FinalizeIterationAndCheckIfMinimizerCanContinue.
  double best_x[3] = {x[0], x[1], x[2]};
  double best_cost = x_cost;

  sum.termination = kCeresNoConvergence;
  bool done = gradient_max_norm(at_x) <= kGradientTolerance;   // iteration 0
  if (done) sum.termination = kCeresConvergence;
  bool last_step_successful = false;
  for (int iteration = 1; !done; ++iteration) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue of the previous iteration.
    if (iteration - 1 >= options.max_num_iterations) { sum.termination = kCeresNoConvergence; break; }
    if (last_step_successful && gradient_max_norm(at_x) <= kGradientTolerance) {
      sum.termination = kCeresConvergence;
      break;
    }
    if (radius < kMinRadius) { sum.termination = kCeresConvergence; break; }
    last_step_successful = false;

    // ---- LevenbergMarquardtStrategy::ComputeStep on the column-scaled Jacobian -----------
    double Hs[3][3], gs[3];
    for (int a = 0; a < 3; ++a) {
      gs[a] = at_x.g[a] * scale[a];
      for (int b = 0; b < 3; ++b) Hs[a][b] = at_x.H[a][b] * scale[a] * scale[b];
    }
    if (!reuse_diagonal) {
      for (int a = 0; a < 3; ++a)
        diagonal[a] = std::min(std::max(Hs[a][a], kMinLmDiagonal), kMaxLmDiagonal);
    }
    double A[3][3], step[3];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) A[a][b] = Hs[a][b] + (a == b ? diagonal[a] / radius : 0.);
    // min |J y - r|^2 + |D y|^2, step = -y.
    bool solved = SolveSpd3(A, gs, step);
    for (int a = 0; a < 3; ++a) step[a] = -step[a];
    reuse_diagonal = true;
    // model_cost_change = -(J step)^T (r + J step / 2) = -(step^T g + step^T H step / 2)
    double model_cost_change = 0.;
    if (solved) {
      double sg = 0., sHs = 0.;
      for (int a = 0; a < 3; ++a) {
        sg += step[a] * gs[a];
        for (int b = 0; b < 3; ++b) sHs += step[a] * Hs[a][b] * step[b];
      }
      model_cost_change = -(sg + 0.5 * sHs);
    }
    if (!solved || !(model_cost_change > 0.)) {
      // HandleInvalidStep: treated as an unsuccessful iteration; LM halves the radius.
      if (++num_consecutive_invalid_steps >= kMaxConsecutiveInvalidSteps) {
        sum.termination = kCeresFailure;
        break;
      }
      radius *= 0.5;
      reuse_diagonal = false;
      ++sum.num_unsuccessful_steps;
      continue;
    }
    num_consecutive_invalid_steps = 0;
    double delta[3], candidate[3];
    for (int a = 0; a < 3; ++a) {
      delta[a] = step[a] * scale[a];
      candidate[a] = x[a] + delta[a];
    }
    Evaluation at_candidate;
    evaluate(candidate, true, &at_candidate);   // (Ceres evaluates the Jacobian only if accepted)
    const double candidate_cost = at_candidate.cost;

    // ParameterToleranceReached / FunctionToleranceReached: checked before the step is taken.
    const double step_norm =
        std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
    if (step_norm <= kParameterTolerance * (x_norm + kParameterTolerance)) {
      sum.termination = kCeresConvergence;
      break;
    }
    const double cost_change = x_cost - candidate_cost;
    if (std::fabs(cost_change) <= kFunctionTolerance * x_cost) {
      sum.termination = kCeresConvergence;
      break;
    }
    // TrustRegionStepEvaluator::StepQuality.
    const double relative_decrease_now = (current_cost - candidate_cost) / model_cost_change;
    const double historical_relative_decrease =
        (reference_cost - candidate_cost) /
        (accumulated_reference_model_cost_change + model_cost_change);
    const double relative_decrease = std::max(relative_decrease_now, historical_relative_decrease);
    if (relative_decrease > kMinRelativeDecrease) {
      // HandleSuccessfulStep.
      for (int a = 0; a < 3; ++a) x[a] = candidate[a];
      x_norm = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
      at_x = at_candidate;
      x_cost = candidate_cost;
      last_step_successful = true;
      ++sum.num_successful_steps;
      if (x_cost < best_cost) {
        best_cost = x_cost;
        for (int a = 0; a < 3; ++a) best_x[a] = x[a];
      }
      // LevenbergMarquardtStrategy::StepAccepted.
      radius = radius / std::max(1. / 3., 1. - std::pow(2. * relative_decrease - 1., 3));
      radius = std::min(kMaxRadius, radius);
      decrease_factor = 2.;
      reuse_diagonal = false;
      // TrustRegionStepEvaluator::StepAccepted.
      current_cost = candidate_cost;
      accumulated_candidate_model_cost_change += model_cost_change;
      accumulated_reference_model_cost_change += model_cost_change;
      if (candidate_cost < minimum_cost) {
        minimum_cost = candidate_cost;
        num_consecutive_nonmonotonic_steps = 0;
        candidate_cost_eval = candidate_cost;
        accumulated_candidate_model_cost_change = 0.;
      } else {
        ++num_consecutive_nonmonotonic_steps;
        if (candidate_cost > candidate_cost_eval) {
          candidate_cost_eval = candidate_cost;
          accumulated_candidate_model_cost_change = 0.;
        }
      }
      if (num_consecutive_nonmonotonic_steps == max_consecutive_nonmonotonic_steps) {
        reference_cost = candidate_cost_eval;
        accumulated_reference_model_cost_change = accumulated_candidate_model_cost_change;
      }
    } else {
      // HandleUnsuccessfulStep: LevenbergMarquardtStrategy::StepRejected.
      radius = radius / decrease_factor;
      decrease_factor *= 2.;
      reuse_diagonal = true;
      ++sum.num_unsuccessful_steps;
    }
  }
  sum.final_cost = best_cost;
  *pose_estimate = Pose2d{best_x[0], best_x[1], best_x[2]};
}

}  // namespace oracle
