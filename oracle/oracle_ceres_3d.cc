// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_ceres_3d.h for what is restated and what
// pins it).
#include "oracle_ceres_3d.h"

#include <limits>

#include <algorithm>
#include <cmath>

namespace oracle {
namespace {

constexpr int kMaxLocal = 6;

// Cholesky solve of a k x k SPD system (k <= 6).
bool SolveSpd(int k, const double A[kMaxLocal][kMaxLocal], const double* b, double* x) {
  double L[kMaxLocal][kMaxLocal] = {{0}};
  for (int i = 0; i < k; ++i) {
    for (int j = 0; j <= i; ++j) {
      double sum = A[i][j];
      for (int m = 0; m < j; ++m) sum -= L[i][m] * L[j][m];
      if (i == j) {
        if (!(sum > 0.)) return false;
        L[i][i] = std::sqrt(sum);
      } else {
        L[i][j] = sum / L[j][j];
      }
    }
  }
  double y[kMaxLocal];
  for (int i = 0; i < k; ++i) {
    double sum = b[i];
    for (int m = 0; m < i; ++m) sum -= L[i][m] * y[m];
    y[i] = sum / L[i][i];
  }
  for (int i = k - 1; i >= 0; --i) {
    double sum = y[i];
    for (int m = i + 1; m < k; ++m) sum -= L[m][i] * x[m];
    x[i] = sum / L[i][i];
  }
  for (int i = 0; i < k; ++i)
    if (!std::isfinite(x[i])) return false;
  return true;
}

// ceres::QuaternionProduct / common::QuaternionProduct (math.h:76-85), (w, x, y, z).
void QuaternionProduct(const double z[4], const double w[4], double zw[4]) {
  zw[0] = z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3];
  zw[1] = z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2];
  zw[2] = z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1];
  zw[3] = z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0];
}

void Cross(const double a[3], const double b[3], double out[3]) {
  out[0] = a[1] * b[2] - a[2] * b[1];
  out[1] = a[2] * b[0] - a[0] * b[2];
  out[2] = a[0] * b[1] - a[1] * b[0];
}

struct Evaluation {   // cost = 1/2 |r|^2, g = J^T r, H = J^T J of the LOCAL Jacobian
  double cost = 0.;
  double g[kMaxLocal] = {0};
  double H[kMaxLocal][kMaxLocal] = {{0}};
};

// Jacobian of Plus at delta = 0: rotation[4] x local rotation dims.
int PlusJacobian(bool yaw_only, const double q[4], double jac[4][3]) {
  if (yaw_only) {
    // d/dd [(sqrt(1 - d^2), 0, 0, d) * q] at 0 = (0, 0, 0, 1) * q
    jac[0][0] = -q[3]; jac[1][0] = -q[2]; jac[2][0] = q[1]; jac[3][0] = q[0];
    return 1;
  }
  // QuaternionParameterization::ComputeJacobian (local_parameterization.cc).
  jac[0][0] = -q[1]; jac[0][1] = -q[2]; jac[0][2] = -q[3];
  jac[1][0] = q[0];  jac[1][1] = q[3];  jac[1][2] = -q[2];
  jac[2][0] = -q[3]; jac[2][1] = q[0];  jac[2][2] = q[1];
  jac[3][0] = q[2];  jac[3][1] = -q[1]; jac[3][2] = q[0];
  return 3;
}

void Plus(bool yaw_only, const double q[4], const double* delta, double out[4]) {
  double q_delta[4];
  if (yaw_only) {
    const double c = std::min(std::max(delta[0], -0.5), 0.5);
    q_delta[0] = std::sqrt(1. - c * c); q_delta[1] = 0.; q_delta[2] = 0.; q_delta[3] = c;
  } else {
    const double norm = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
    if (!(norm > 0.)) {
      for (int k = 0; k < 4; ++k) out[k] = q[k];
      return;
    }
    const double s = std::sin(norm) / norm;
    q_delta[0] = std::cos(norm); q_delta[1] = s * delta[0]; q_delta[2] = s * delta[1];
    q_delta[3] = s * delta[2];
  }
  QuaternionProduct(q_delta, q, out);
}

}  // namespace

namespace {
// InterpolatedGrid<GridType>::GetInterpolatedValue (interpolated_grid.h:52-93) for either grid:
// `cell_of` = GetCellIndex, `value_at` = GetProbability / GetIntensity.
template <typename Grid, typename ValueAt>
double Interpolated(const Grid& grid, ValueAt value_at, double x, double y, double z,
                    double gradient[3]) {
  const float res = grid.resolution();
  // CenterOfLowerVoxel (:113-130): centre of the cell containing the point (f32), moved to the
  // next lower centre where it lies above the coordinate.
  const Cell3i at = grid.GetCellIndex(V3f{static_cast<float>(x), static_cast<float>(y),
                                          static_cast<float>(z)});
  float cx = static_cast<float>(at.x) * res, cy = static_cast<float>(at.y) * res,
        cz = static_cast<float>(at.z) * res;
  if (cx > x) cx -= res;
  if (cy > y) cy -= res;
  if (cz > z) cz -= res;
  const double x1 = cx, y1 = cy, z1 = cz;
  const double x2 = cx + res, y2 = cy + res, z2 = cz + res;     // f32 additions (:100-110)
  const Cell3i i1 = grid.GetCellIndex(V3f{cx, cy, cz});
  const auto q = [&](int dx, int dy, int dz) {
    return static_cast<double>(value_at(Cell3i{i1.x + dx, i1.y + dy, i1.z + dz}));
  };
  const double q111 = q(0, 0, 0), q112 = q(0, 0, 1), q121 = q(0, 1, 0), q122 = q(0, 1, 1);
  const double q211 = q(1, 0, 0), q212 = q(1, 0, 1), q221 = q(1, 1, 0), q222 = q(1, 1, 1);
  const double nx = (x - x1) / (x2 - x1), ny = (y - y1) / (y2 - y1), nz = (z - z1) / (z2 - z1);
  const double nxx = nx * nx, nxxx = nx * nxx, nyy = ny * ny, nyyy = ny * nyy, nzz = nz * nz,
               nzzz = nz * nzz;
  // A (2 t^3 - 3 t^2 + 1) + B (-2 t^3 + 3 t^2), first in z, then y, then x (:78-91).
  const auto blend = [](double a, double b, double t2, double t3) {
    return (a - b) * t3 * 2. + (b - a) * t2 * 3. + a;
  };
  const auto dblend = [](double a, double b, double t, double t2) {   // d/dt of blend
    return (a - b) * t2 * 6. + (b - a) * t * 6.;
  };
  const double q11 = blend(q111, q112, nzz, nzzz), q12 = blend(q121, q122, nzz, nzzz);
  const double q21 = blend(q211, q212, nzz, nzzz), q22 = blend(q221, q222, nzz, nzzz);
  const double q1 = blend(q11, q12, nyy, nyyy), q2 = blend(q21, q22, nyy, nyyy);
  const double value = blend(q1, q2, nxx, nxxx);
  if (gradient) {
    const double d11 = dblend(q111, q112, nz, nzz), d12 = dblend(q121, q122, nz, nzz);
    const double d21 = dblend(q211, q212, nz, nzz), d22 = dblend(q221, q222, nz, nzz);
    // d/dnz passes linearly through the y and x blends.
    const double q1_z = blend(d11, d12, nyy, nyyy), q2_z = blend(d21, d22, nyy, nyyy);
    const double q1_y = dblend(q11, q12, ny, nyy), q2_y = dblend(q21, q22, ny, nyy);
    gradient[0] = dblend(q1, q2, nx, nxx) / (x2 - x1);
    gradient[1] = blend(q1_y, q2_y, nxx, nxxx) / (y2 - y1);
    gradient[2] = blend(q1_z, q2_z, nxx, nxxx) / (z2 - z1);
  }
  return value;
}
}  // namespace

double InterpolatedProbability(const HybridGridView& grid, double x, double y, double z,
                               double gradient[3]) {
  return Interpolated(grid, [&](const Cell3i& c) { return grid.GetProbability(c); }, x, y, z,
                      gradient);
}
double InterpolatedIntensity(const IntensityGridView& grid, double x, double y, double z,
                             double gradient[3]) {
  return Interpolated(grid, [&](const Cell3i& c) { return grid.GetIntensity(c); }, x, y, z,
                      gradient);
}

IntensityGridView::IntensityGridView(float resolution, const IntensityVoxel* voxels, int64_t n)
    : resolution_(resolution) {
  if (n <= 0) return;
  Cell3i lo{voxels[0].x, voxels[0].y, voxels[0].z}, hi = lo;
  for (int64_t i = 1; i < n; ++i) {
    lo.x = std::min(lo.x, voxels[i].x); hi.x = std::max(hi.x, voxels[i].x);
    lo.y = std::min(lo.y, voxels[i].y); hi.y = std::max(hi.y, voxels[i].y);
    lo.z = std::min(lo.z, voxels[i].z); hi.z = std::max(hi.z, voxels[i].z);
  }
  cells_.Reset(lo, hi);
  for (int64_t i = 0; i < n; ++i)
    if (voxels[i].count != 0)       // GetIntensity: cell.sum / cell.count (hybrid_grid.h:563-570)
      *cells_.mutable_value(voxels[i].x, voxels[i].y, voxels[i].z) =
          voxels[i].sum / voxels[i].count;
}

std::vector<ResidualBlock3D> CeresResidualBlocks3D(const std::vector<CloudAndGrid3D>& pairs) {
  std::vector<ResidualBlock3D> blocks;
  size_t row = 0;
  for (const CloudAndGrid3D& p : pairs) {
    const size_t n = p.point_cloud->size();
    blocks.push_back({row, row + n, 0.});
    row += n;
    if (p.intensity_hybrid_grid) {
      blocks.push_back({row, row + n, p.huber_scale});
      row += n;
    }
  }
  blocks.push_back({row, row + 3, 0.});
  blocks.push_back({row + 3, row + 6, 0.});
  return blocks;
}

void CeresResiduals3D(const CeresOptions3D& options, const double target_translation[3],
                      const double target_rotation[4], const std::vector<CloudAndGrid3D>& pairs,
                      const double translation[3], const double rotation[4],
                      std::vector<double>* residuals, std::vector<double>* jacobian) {
  size_t total = 6;
  for (const CloudAndGrid3D& p : pairs)
    total += p.point_cloud->size() * (p.intensity_hybrid_grid ? 2 : 1);
  residuals->assign(total, 0.);
  if (jacobian) jacobian->assign(total * 7, 0.);
  const double w = rotation[0];
  const double u[3] = {rotation[1], rotation[2], rotation[3]};
  size_t row = 0;
  for (size_t k = 0; k < pairs.size(); ++k) {
    const PointCloud3& cloud = *pairs[k].point_cloud;
    const HybridGridView& grid = *pairs[k].hybrid_grid;
    const double scaling =
        options.occupied_space_weight[k] / std::sqrt(static_cast<double>(cloud.size()));
    // One residual block per grid of the pair: occupied space, then (optional) intensity.
    for (int term = 0; term < (pairs[k].intensity_hybrid_grid ? 2 : 1); ++term) {
      const bool intensity_term = term == 1;
      const double term_scaling =
          intensity_term ? pairs[k].intensity_weight / std::sqrt(static_cast<double>(cloud.size()))
                         : scaling;
      for (size_t i = 0; i < cloud.size(); ++i, ++row) {
        if (intensity_term && (*pairs[k].intensities)[i] > pairs[k].intensity_threshold)
          continue;                       // residual[i] = T(0.f) (intensity_cost_function_3d.h:69-71)
        const double v[3] = {static_cast<double>(cloud[i].x), static_cast<double>(cloud[i].y),
                             static_cast<double>(cloud[i].z)};
        // Eigen: uv = 2 (u x v); rotated = v + w uv + u x uv; world = rotated + translation.
        double uv[3], uuv[3];
        Cross(u, v, uv);
        for (int a = 0; a < 3; ++a) uv[a] += uv[a];
        Cross(u, uv, uuv);
        double world[3];
        for (int a = 0; a < 3; ++a) world[a] = ((v[a] + w * uv[a]) + uuv[a]) + translation[a];
        double grad[3];
        double sign;                      // d residual / d interpolated value, over the scaling
        if (intensity_term) {
          const double interpolated = InterpolatedIntensity(
              *pairs[k].intensity_hybrid_grid, world[0], world[1], world[2],
              jacobian ? grad : nullptr);
          (*residuals)[row] =
              term_scaling * (interpolated - static_cast<double>((*pairs[k].intensities)[i]));
          sign = 1.;
        } else {
          const double probability = InterpolatedProbability(grid, world[0], world[1], world[2],
                                                             jacobian ? grad : nullptr);
          (*residuals)[row] = term_scaling * (1. - probability);
          sign = -1.;
        }
        if (jacobian) {
          const double sc = sign * term_scaling;
          double* J = jacobian->data() + 7 * row;
          for (int a = 0; a < 3; ++a) J[a] = sc * grad[a];              // d world / d t = I
          // d world / d w = uv
          J[3] = sc * (grad[0] * uv[0] + grad[1] * uv[1] + grad[2] * uv[2]);
          for (int c = 0; c < 3; ++c) {
            // d world / d u_c = w 2 (e_c x v) + e_c x uv + u x (2 e_c x v)
            double e[3] = {0., 0., 0.};
            e[c] = 1.;
            double ev[3], euv[3], uev[3];
            Cross(e, v, ev);
            for (int a = 0; a < 3; ++a) ev[a] += ev[a];
            Cross(e, uv, euv);
            Cross(u, ev, uev);
            double d = 0.;
            for (int a = 0; a < 3; ++a) d += grad[a] * ((w * ev[a] + euv[a]) + uev[a]);
            J[4 + c] = sc * d;
          }
        }
      }
    }
  }
  // TranslationDeltaCostFunctor3D.
  for (int a = 0; a < 3; ++a, ++row) {
    (*residuals)[row] = options.translation_weight * (translation[a] - target_translation[a]);
    if (jacobian) (*jacobian)[7 * row + a] = options.translation_weight;
  }
  // RotationDeltaCostFunctor3D: vector part of target^-1 * rotation.
  const double inv[4] = {target_rotation[0], -target_rotation[1], -target_rotation[2],
                         -target_rotation[3]};
  double delta[4];
  QuaternionProduct(inv, rotation, delta);
  const double rows[3][4] = {{inv[1], inv[0], -inv[3], inv[2]},
                             {inv[2], inv[3], inv[0], -inv[1]},
                             {inv[3], -inv[2], inv[1], inv[0]}};
  for (int a = 0; a < 3; ++a, ++row) {
    (*residuals)[row] = options.rotation_weight * delta[1 + a];
    if (jacobian)
      for (int c = 0; c < 4; ++c) (*jacobian)[7 * row + 3 + c] = options.rotation_weight * rows[a][c];
  }
}

void CeresScanMatcher3DMatch(const CeresOptions3D& options, const double target_translation[3],
                             const Pose3d& initial_pose_estimate,
                             const std::vector<CloudAndGrid3D>& pairs, Pose3d* pose_estimate,
                             CeresSummary2D* summary) {
  // Solver::Options defaults of the pinned Ceres (include/ceres/solver.h).
  const double kFunctionTolerance = 1e-6, kGradientTolerance = 1e-10, kParameterTolerance = 1e-8;
  const double kMinRelativeDecrease = 1e-3, kMinLmDiagonal = 1e-6, kMaxLmDiagonal = 1e32;
  const double kMaxRadius = 1e16, kMinRadius = 1e-32;
  const int kMaxConsecutiveInvalidSteps = 5;
  const int max_consecutive_nonmonotonic_steps = options.use_nonmonotonic_steps ? 5 : 0;
  const bool yaw_only = options.only_optimize_yaw;
  const int K = 3 + (yaw_only ? 1 : 3);

  const double target_rotation[4] = {initial_pose_estimate.q.w, initial_pose_estimate.q.x,
                                     initial_pose_estimate.q.y, initial_pose_estimate.q.z};
  std::vector<double> r, J;
  const std::vector<ResidualBlock3D> blocks = CeresResidualBlocks3D(pairs);
  const auto evaluate = [&](const double x[7], Evaluation* e) {
    CeresResiduals3D(options, target_translation, target_rotation, pairs, x, x + 3, &r, &J);
    double plus[4][3];
    const int kr = PlusJacobian(yaw_only, x + 3, plus);
    *e = Evaluation();
    double local[kMaxLocal];
    for (const ResidualBlock3D& block : blocks) {
      // ResidualBlock::Evaluate: rho over the block's squared norm; Corrector (rho'' <= 0 for
      // Huber): residuals and Jacobian times sqrt(rho'), cost 1/2 rho.
      double s = 0.;
      for (size_t i = block.begin; i < block.end; ++i) s += r[i] * r[i];
      double rho0 = s, rho1 = 1.;
      if (block.huber_a > 0.) {
        const double a_ = block.huber_a, b_ = a_ * a_;
        if (s > b_) {
          const double root = std::sqrt(s);
          rho0 = 2. * a_ * root - b_;
          rho1 = std::max(std::numeric_limits<double>::min(), a_ / root);
        }
      }
      e->cost += rho0;
      for (size_t i = block.begin; i < block.end; ++i) {
        const double* row = J.data() + 7 * i;
        for (int a = 0; a < 3; ++a) local[a] = row[a];
        for (int c = 0; c < kr; ++c) {
          double sum = 0.;
          for (int m = 0; m < 4; ++m) sum += row[3 + m] * plus[m][c];
          local[3 + c] = sum;
        }
        for (int a = 0; a < K; ++a) {
          e->g[a] += rho1 * (local[a] * r[i]);
          for (int b = 0; b < K; ++b) e->H[a][b] += rho1 * (local[a] * local[b]);
        }
      }
    }
    e->cost *= 0.5;
  };
  const auto norm7 = [](const double x[7]) {
    double s = 0.;
    for (int a = 0; a < 7; ++a) s += x[a] * x[a];
    return std::sqrt(s);
  };

  double x[7] = {initial_pose_estimate.t[0], initial_pose_estimate.t[1], initial_pose_estimate.t[2],
                 target_rotation[0], target_rotation[1], target_rotation[2], target_rotation[3]};
  Evaluation at_x;
  evaluate(x, &at_x);
  CeresSummary2D local_summary;
  CeresSummary2D& sum = summary ? *summary : local_summary;
  sum = CeresSummary2D();
  sum.initial_cost = at_x.cost;
  double scale[kMaxLocal];
  for (int a = 0; a < K; ++a) scale[a] = 1. / (1. + std::sqrt(at_x.H[a][a]));
  double x_cost = at_x.cost;
  double x_norm = norm7(x);
  // TrustRegionMinimizer::ComputeGradientMaxNorm (trust_region_minimizer.cc): the gradient
  // PROJECTED through the parameterization, |x - Plus(x, -g)|_inf over the ambient coordinates.
  // For the Euclidean translation block that is |g|; for the rotation block it is not.  (`x`
  // is always the point at_x was evaluated at.)
  const auto gradient_max_norm = [K, &x, &options](const Evaluation& e) {
    double m = 0.;
    for (int a = 0; a < 3; ++a) m = std::max(m, std::fabs(e.g[a]));
    double negative[3] = {0., 0., 0.}, projected[4];
    for (int a = 3; a < K; ++a) negative[a - 3] = -e.g[a];
    Plus(options.only_optimize_yaw, x + 3, negative, projected);
    for (int a = 0; a < 4; ++a) m = std::max(m, std::fabs(x[3 + a] - projected[a]));
    return m;
  };

  double radius = 1e4, decrease_factor = 2.;
  bool reuse_diagonal = false;
  double diagonal[kMaxLocal] = {0};
  double minimum_cost = x_cost, current_cost = x_cost, reference_cost = x_cost,
         candidate_cost_eval = x_cost;
  double accumulated_reference_model_cost_change = 0., accumulated_candidate_model_cost_change = 0.;
  int num_consecutive_nonmonotonic_steps = 0;
  int num_consecutive_invalid_steps = 0;
  double best_x[7];
  for (int a = 0; a < 7; ++a) best_x[a] = x[a];
  double best_cost = x_cost;

  sum.termination = kCeresNoConvergence;
  bool done = gradient_max_norm(at_x) <= kGradientTolerance;
  if (done) sum.termination = kCeresConvergence;
  bool last_step_successful = false;
  for (int iteration = 1; !done; ++iteration) {
    if (iteration - 1 >= options.max_num_iterations) { sum.termination = kCeresNoConvergence; break; }
    if (last_step_successful && gradient_max_norm(at_x) <= kGradientTolerance) {
      sum.termination = kCeresConvergence;
      break;
    }
    if (radius < kMinRadius) { sum.termination = kCeresConvergence; break; }
    last_step_successful = false;

    double Hs[kMaxLocal][kMaxLocal], gs[kMaxLocal];
    for (int a = 0; a < K; ++a) {
      gs[a] = at_x.g[a] * scale[a];
      for (int b = 0; b < K; ++b) Hs[a][b] = at_x.H[a][b] * scale[a] * scale[b];
    }
    if (!reuse_diagonal)
      for (int a = 0; a < K; ++a)
        diagonal[a] = std::min(std::max(Hs[a][a], kMinLmDiagonal), kMaxLmDiagonal);
    double A[kMaxLocal][kMaxLocal], step[kMaxLocal];
    for (int a = 0; a < K; ++a)
      for (int b = 0; b < K; ++b) A[a][b] = Hs[a][b] + (a == b ? diagonal[a] / radius : 0.);
    const bool solved = SolveSpd(K, A, gs, step);
    for (int a = 0; a < K; ++a) step[a] = -step[a];
    reuse_diagonal = true;
    double model_cost_change = 0.;
    if (solved) {
      double sg = 0., sHs = 0.;
      for (int a = 0; a < K; ++a) {
        sg += step[a] * gs[a];
        for (int b = 0; b < K; ++b) sHs += step[a] * Hs[a][b] * step[b];
      }
      model_cost_change = -(sg + 0.5 * sHs);
    }
    if (!solved || !(model_cost_change > 0.)) {
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
    double delta[kMaxLocal], candidate[7];
    for (int a = 0; a < K; ++a) delta[a] = step[a] * scale[a];
    for (int a = 0; a < 3; ++a) candidate[a] = x[a] + delta[a];     // identity parameterization
    Plus(yaw_only, x + 3, delta + 3, candidate + 3);
    Evaluation at_candidate;
    evaluate(candidate, &at_candidate);
    const double candidate_cost = at_candidate.cost;

    double step_sq = 0.;
    for (int a = 0; a < 7; ++a) step_sq += (x[a] - candidate[a]) * (x[a] - candidate[a]);
    const double step_norm = std::sqrt(step_sq);
    if (step_norm <= kParameterTolerance * (x_norm + kParameterTolerance)) {
      sum.termination = kCeresConvergence;
      break;
    }
    const double cost_change = x_cost - candidate_cost;
    if (std::fabs(cost_change) <= kFunctionTolerance * x_cost) {
      sum.termination = kCeresConvergence;
      break;
    }
    const double relative_decrease_now = (current_cost - candidate_cost) / model_cost_change;
    const double historical_relative_decrease =
        (reference_cost - candidate_cost) /
        (accumulated_reference_model_cost_change + model_cost_change);
    const double relative_decrease = std::max(relative_decrease_now, historical_relative_decrease);
    if (relative_decrease > kMinRelativeDecrease) {
      for (int a = 0; a < 7; ++a) x[a] = candidate[a];
      x_norm = norm7(x);
      at_x = at_candidate;
      x_cost = candidate_cost;
      last_step_successful = true;
      ++sum.num_successful_steps;
      if (x_cost < best_cost) {
        best_cost = x_cost;
        for (int a = 0; a < 7; ++a) best_x[a] = x[a];
      }
      radius = radius / std::max(1. / 3., 1. - std::pow(2. * relative_decrease - 1., 3));
      radius = std::min(kMaxRadius, radius);
      decrease_factor = 2.;
      reuse_diagonal = false;
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
      radius = radius / decrease_factor;
      decrease_factor *= 2.;
      reuse_diagonal = true;
      ++sum.num_unsuccessful_steps;
    }
  }
  sum.final_cost = best_cost;
  for (int a = 0; a < 3; ++a) pose_estimate->t[a] = best_x[a];
  pose_estimate->q = Qd{best_x[3], best_x[4], best_x[5], best_x[6]};
}

}  // namespace oracle
