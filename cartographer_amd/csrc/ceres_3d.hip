// CeresScanMatcher3D::Match on gfx950: the refinement step that follows
// the correlative match in both 3D callers
//   mapping/internal/3d/local_trajectory_builder_3d.cc (after the real-time matcher)
//   mapping/internal/constraints/constraint_builder_3d.cc:255-263 (after the fast matcher)
//
// Reference: SM3/ceres_scan_matcher_3d.cc:90-156, SM3/occupied_space_cost_function_3d.h:66-97,
// SM3/interpolated_grid.h:36-151 (piecewise cubic with vanishing derivatives at the voxel
// centres), SM3/translation_delta_cost_functor_3d.h:43-50, SM3/rotation_delta_cost_functor_3d.h:
// 43-55, mapping/internal/3d/rotation_parameterization.h:27-39, SM3/intensity_cost_function_3d.h:
// 37-91 (optional per pair: InterpolatedGrid over the IntensityHybridGrid, ceres::HuberLoss on
// the block).
// The least-squares solver is Ceres (third party, absent from the reference tree): what runs
// here is its published trust-region Levenberg-Marquardt loop with Solver::Options defaults in
// the tangent space of the pose (translation: identity; rotation: QuaternionParameterization,
// or YawOnlyQuaternionPlus with only_optimize_yaw), restated the same way as
// oracle/oracle_ceres_3d.cc (which documents every rule taken from Ceres and what pins it).
//
// One workgroup per match.  Per evaluation every thread takes a few points of every cloud:
// Eigen's quaternion-times-vector in f64, the eight probabilities around the point (uint16 ->
// the table's f32 expression -> f64, gathered from a dense brick in HBM / L2), the cubic blend
// and its gradient, the residual and its six (or four) local partials, accumulated as cost,
// J^T r and J^T J in f64 (28 numbers per thread), reduced across the wavefront and across the
// four wavefronts through LDS in a FIXED order -- results do not depend on scheduling.  Every
// thread carries the (identical) minimizer state.
#include <cmath>

#include "scan_matching_3d.h"

namespace cmx {
namespace {

constexpr int kCeres3DThreads = 256;
constexpr int kMaxLocal = 6;
constexpr int kMaxPairs = 3;
constexpr int kNumSums = 1 + kMaxLocal + kMaxLocal * (kMaxLocal + 1) / 2;   // 28

struct Ceres3DPair {
  Brick grid;              // uint16 values
  float resolution;
  int n;
  const float* xyz;
  double scaling;          // occupied_space_weight / sqrt(n)
  // IntensityCostFunction3D of this pair (has_intensity != 0)
  int has_intensity;
  float intensity_threshold;
  Brick igrid;             // f32: AverageIntensityData sum / count, 0 elsewhere
  const float* intensities;
  double iscaling;         // weight / sqrt(n)
  double huber_a;          // HuberLoss(a): rho over the block's squared norm
};

struct Ceres3DProblem {
  Ceres3DPair pair[kMaxPairs];
  int num_pairs;
  int yaw_only, use_nonmonotonic_steps, max_num_iterations;
  double translation_weight, rotation_weight;
  double target[3];        // target translation
  double init[7];          // initial pose: t, q = (w, x, y, z); q is also the rotation target
  double* out;             // [12]: pose (7), initial cost, final cost, successful, unsuccessful,
                           //       termination
};

// HybridGrid::GetProbability: value 0 -> kMinProbability (hybrid_grid.h:521-523).
__device__ __forceinline__ double Probability(const Brick& b, int x, int y, int z) {
  return static_cast<double>(ValueToProbabilityDev(BrickValueU16(b, x, y, z)));
}

// InterpolatedGrid::GetInterpolatedValue and its gradient (SM3/interpolated_grid.h:57-92,
// 99-130).
__device__ __forceinline__ double Intensity(const Brick& b, int x, int y, int z) {
  const int ix = x - b.lo_x, iy = y - b.lo_y, iz = z - b.lo_z;
  const bool inside = static_cast<unsigned>(ix) < static_cast<unsigned>(b.nx) &&
                      static_cast<unsigned>(iy) < static_cast<unsigned>(b.ny) &&
                      static_cast<unsigned>(iz) < static_cast<unsigned>(b.nz);
  const float v = static_cast<const float*>(b.cells)[
      inside ? (static_cast<size_t>(iz) * b.ny + iy) * b.nx + ix : 0];
  return inside ? static_cast<double>(v) : 0.;
}

template <bool kIntensity>
__device__ __forceinline__ double Interpolate(const Ceres3DPair& p, double x, double y, double z,
                                              double gradient[3]) {
  const float res = p.resolution;
  const Brick& grid = kIntensity ? p.igrid : p.grid;
  const auto value = [&](int cx_, int cy_, int cz_) {
    return kIntensity ? Intensity(grid, cx_, cy_, cz_) : Probability(grid, cx_, cy_, cz_);
  };
  const int3 at = CellIndex3(F3{static_cast<float>(x), static_cast<float>(y), static_cast<float>(z)},
                             res);
  float cx = static_cast<float>(at.x) * res, cy = static_cast<float>(at.y) * res,
        cz = static_cast<float>(at.z) * res;
  if (static_cast<double>(cx) > x) cx -= res;
  if (static_cast<double>(cy) > y) cy -= res;
  if (static_cast<double>(cz) > z) cz -= res;
  const double x1 = cx, y1 = cy, z1 = cz;
  const double x2 = cx + res, y2 = cy + res, z2 = cz + res;     // f32 additions
  const int3 i1 = CellIndex3(F3{cx, cy, cz}, res);
  const double q111 = value(i1.x, i1.y, i1.z);
  const double q112 = value(i1.x, i1.y, i1.z + 1);
  const double q121 = value(i1.x, i1.y + 1, i1.z);
  const double q122 = value(i1.x, i1.y + 1, i1.z + 1);
  const double q211 = value(i1.x + 1, i1.y, i1.z);
  const double q212 = value(i1.x + 1, i1.y, i1.z + 1);
  const double q221 = value(i1.x + 1, i1.y + 1, i1.z);
  const double q222 = value(i1.x + 1, i1.y + 1, i1.z + 1);
  const double nx = (x - x1) / (x2 - x1), ny = (y - y1) / (y2 - y1), nz = (z - z1) / (z2 - z1);
  const double nxx = nx * nx, nxxx = nx * nxx, nyy = ny * ny, nyyy = ny * nyy, nzz = nz * nz,
               nzzz = nz * nzz;
  const auto blend = [](double a, double b, double t2, double t3) {
    return (a - b) * t3 * 2. + (b - a) * t2 * 3. + a;
  };
  const auto dblend = [](double a, double b, double t, double t2) {
    return (a - b) * t2 * 6. + (b - a) * t * 6.;
  };
  const double q11 = blend(q111, q112, nzz, nzzz), q12 = blend(q121, q122, nzz, nzzz);
  const double q21 = blend(q211, q212, nzz, nzzz), q22 = blend(q221, q222, nzz, nzzz);
  const double q1 = blend(q11, q12, nyy, nyyy), q2 = blend(q21, q22, nyy, nyyy);
  const double d11 = dblend(q111, q112, nz, nzz), d12 = dblend(q121, q122, nz, nzz);
  const double d21 = dblend(q211, q212, nz, nzz), d22 = dblend(q221, q222, nz, nzz);
  const double q1_z = blend(d11, d12, nyy, nyyy), q2_z = blend(d21, d22, nyy, nyyy);
  const double q1_y = dblend(q11, q12, ny, nyy), q2_y = dblend(q21, q22, ny, nyy);
  gradient[0] = dblend(q1, q2, nx, nxx) / (x2 - x1);
  gradient[1] = blend(q1_y, q2_y, nxx, nxxx) / (y2 - y1);
  gradient[2] = blend(q1_z, q2_z, nxx, nxxx) / (z2 - z1);
  return blend(q1, q2, nxx, nxxx);
}

__device__ __forceinline__ void Cross3(const double a[3], const double b[3], double out[3]) {
  out[0] = a[1] * b[2] - a[2] * b[1];
  out[1] = a[2] * b[0] - a[0] * b[2];
  out[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void QuatProduct(const double z[4], const double w[4], double zw[4]) {
  zw[0] = z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3];
  zw[1] = z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2];
  zw[2] = z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1];
  zw[3] = z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0];
}

__device__ __forceinline__ double WaveSumF64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

struct Eval3D {
  double cost;
  double g[kMaxLocal];
  double H[kMaxLocal][kMaxLocal];
};

// Local (tangent-space) row of one residual: J_local = J_ambient * PlusJacobian.
__device__ __forceinline__ void Accumulate(double r, const double local[kMaxLocal], int K,
                                           double acc[kNumSums]) {
  acc[0] += r * r;
  int h = 1 + kMaxLocal;
  for (int a = 0; a < kMaxLocal; ++a) {
    if (a < K) acc[1 + a] += local[a] * r;
    for (int b = a; b < kMaxLocal; ++b, ++h)
      if (b < K) acc[h] += local[a] * local[b];
  }
}

// Residual blocks at x = (t, q): 1/2 |r|^2, J^T r and J^T J of the local Jacobian (valid in
// every thread).
__device__ void Evaluate3D(const Ceres3DProblem& P, const double x[7], int K,
                           double (*scratch)[kNumSums], Eval3D* e) {
  const double w = x[3];
  const double u[3] = {x[4], x[5], x[6]};
  // Plus-Jacobian of the rotation block at delta = 0.
  double plus[4][3];
  if (P.yaw_only) {
    plus[0][0] = -x[6]; plus[1][0] = -x[5]; plus[2][0] = x[4]; plus[3][0] = x[3];
  } else {
    plus[0][0] = -x[4]; plus[0][1] = -x[5]; plus[0][2] = -x[6];
    plus[1][0] = x[3];  plus[1][1] = x[6];  plus[1][2] = -x[5];
    plus[2][0] = -x[6]; plus[2][1] = x[3];  plus[2][2] = x[4];
    plus[3][0] = x[5];  plus[3][1] = -x[4]; plus[3][2] = x[3];
  }
  const int kr = K - 3;
  // One residual of point i of pair `pr`: occupied space (sign -1: scaling (1 - p)) or intensity
  // (sign +1: scaling (interpolated - intensity)); accumulated into `acc`.
  const auto point_row = [&](const Ceres3DPair& pr, int i, bool intensity_term, double* acc) {
    const double v[3] = {static_cast<double>(pr.xyz[3 * i]), static_cast<double>(pr.xyz[3 * i + 1]),
                         static_cast<double>(pr.xyz[3 * i + 2])};
    double uv[3], uuv[3];
    Cross3(u, v, uv);
    for (int a = 0; a < 3; ++a) uv[a] += uv[a];
    Cross3(u, uv, uuv);
    double world[3];
    for (int a = 0; a < 3; ++a) world[a] = ((v[a] + w * uv[a]) + uuv[a]) + x[a];
    double grad[3];
    double r, sc;
    if (intensity_term) {
      const double interpolated = Interpolate<true>(pr, world[0], world[1], world[2], grad);
      r = pr.iscaling * (interpolated - static_cast<double>(pr.intensities[i]));
      sc = pr.iscaling;
    } else {
      const double probability = Interpolate<false>(pr, world[0], world[1], world[2], grad);
      r = pr.scaling * (1. - probability);
      sc = -pr.scaling;
    }
    double amb[4];            // d r / d (w, ux, uy, uz)
    amb[0] = sc * (grad[0] * uv[0] + grad[1] * uv[1] + grad[2] * uv[2]);
    for (int c = 0; c < 3; ++c) {
      double e3[3] = {0., 0., 0.};
      e3[c] = 1.;
      double ev[3], euv[3], uev[3];
      Cross3(e3, v, ev);
      for (int a = 0; a < 3; ++a) ev[a] += ev[a];
      Cross3(e3, uv, euv);
      Cross3(u, ev, uev);
      double d = 0.;
      for (int a = 0; a < 3; ++a) d += grad[a] * ((w * ev[a] + euv[a]) + uev[a]);
      amb[1 + c] = sc * d;
    }
    double local[kMaxLocal] = {0., 0., 0., 0., 0., 0.};
    for (int a = 0; a < 3; ++a) local[a] = sc * grad[a];
    for (int c = 0; c < kr; ++c) {
      double s = 0.;
      for (int m = 0; m < 4; ++m) s += amb[m] * plus[m][c];
      local[3 + c] = s;
    }
    Accumulate(r, local, K, acc);
  };
  const int wave = threadIdx.x >> 6;
  // Block-wide sums of a thread-local accumulator, in a FIXED order (valid in every thread).
  const auto reduce = [&](const double* acc, double* total) {
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) {
      const double t = WaveSumF64(acc[k]);
      if ((threadIdx.x & 63) == 0) scratch[wave][k] = t;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kNumSums; ++k)
      total[k] = ((scratch[0][k] + scratch[1][k]) + scratch[2][k]) + scratch[3][k];
    __syncthreads();
  };
  double acc[kNumSums];
#pragma unroll
  for (int k = 0; k < kNumSums; ++k) acc[k] = 0.;
  for (int pi = 0; pi < P.num_pairs; ++pi) {
    const Ceres3DPair& pr = P.pair[pi];
    for (int i = threadIdx.x; i < pr.n; i += kCeres3DThreads) point_row(pr, i, false, acc);
  }
  double total[kNumSums];
  reduce(acc, total);
  // Intensity blocks: ceres::HuberLoss acts on the BLOCK's squared norm s (residual_block.cc,
  // corrector.cc with rho'' <= 0): cost 1/2 rho(s), J^T r and J^T J times rho'(s).
  for (int pi = 0; pi < P.num_pairs; ++pi) {
    const Ceres3DPair& pr = P.pair[pi];
    if (!pr.has_intensity) continue;                          // (uniform)
#pragma unroll
    for (int k = 0; k < kNumSums; ++k) acc[k] = 0.;
    for (int i = threadIdx.x; i < pr.n; i += kCeres3DThreads)
      if (!(pr.intensities[i] > pr.intensity_threshold)) point_row(pr, i, true, acc);
    double block[kNumSums];
    reduce(acc, block);
    const double s = block[0], b_ = pr.huber_a * pr.huber_a;
    double rho0 = s, rho1 = 1.;
    if (s > b_) {
      const double root = sqrt(s);
      rho0 = 2. * pr.huber_a * root - b_;
      rho1 = fmax(2.2250738585072014e-308, pr.huber_a / root);
    }
    total[0] += rho0;
#pragma unroll
    for (int k = 1; k < kNumSums; ++k) total[k] += rho1 * block[k];
  }
  // TranslationDeltaCostFunctor3D and RotationDeltaCostFunctor3D: six more rows, the same in
  // every thread.
  for (int a = 0; a < 3; ++a) {
    double local[kMaxLocal] = {0., 0., 0., 0., 0., 0.};
    local[a] = P.translation_weight;
    Accumulate(P.translation_weight * (x[a] - P.target[a]), local, K, total);
  }
  const double inv[4] = {P.init[3], -P.init[4], -P.init[5], -P.init[6]};
  double delta[4];
  QuatProduct(inv, x + 3, delta);
  const double rows[3][4] = {{inv[1], inv[0], -inv[3], inv[2]},
                             {inv[2], inv[3], inv[0], -inv[1]},
                             {inv[3], -inv[2], inv[1], inv[0]}};
  for (int a = 0; a < 3; ++a) {
    double local[kMaxLocal] = {0., 0., 0., 0., 0., 0.};
    for (int c = 0; c < kr; ++c) {
      double s = 0.;
      for (int m = 0; m < 4; ++m) s += P.rotation_weight * rows[a][m] * plus[m][c];
      local[3 + c] = s;
    }
    Accumulate(P.rotation_weight * delta[1 + a], local, K, total);
  }
  e->cost = 0.5 * total[0];
  int h = 1 + kMaxLocal;
  for (int a = 0; a < kMaxLocal; ++a) {
    e->g[a] = total[1 + a];
    for (int b = a; b < kMaxLocal; ++b, ++h) e->H[a][b] = e->H[b][a] = total[h];
  }
}

__device__ bool SolveSpd(int k, const double A[kMaxLocal][kMaxLocal], const double* b, double* x) {
  double L[kMaxLocal][kMaxLocal];
  for (int i = 0; i < kMaxLocal; ++i)
    for (int j = 0; j < kMaxLocal; ++j) L[i][j] = 0.;
  for (int i = 0; i < k; ++i) {
    for (int j = 0; j <= i; ++j) {
      double sum = A[i][j];
      for (int m = 0; m < j; ++m) sum -= L[i][m] * L[j][m];
      if (i == j) {
        if (!(sum > 0.)) return false;
        L[i][i] = sqrt(sum);
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
    if (!isfinite(x[i])) return false;
  return true;
}

// QuaternionParameterization::Plus / YawOnlyQuaternionPlus.
__device__ void PlusRotation(bool yaw_only, const double q[4], const double* delta, double out[4]) {
  double q_delta[4];
  if (yaw_only) {
    const double c = fmin(fmax(delta[0], -0.5), 0.5);
    q_delta[0] = sqrt(1. - c * c); q_delta[1] = 0.; q_delta[2] = 0.; q_delta[3] = c;
  } else {
    const double norm = sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
    if (!(norm > 0.)) {
      for (int k = 0; k < 4; ++k) out[k] = q[k];
      return;
    }
    const double s = sin(norm) / norm;
    q_delta[0] = cos(norm); q_delta[1] = s * delta[0]; q_delta[2] = s * delta[1];
    q_delta[3] = s * delta[2];
  }
  QuatProduct(q_delta, q, out);
}

// The trust-region loop of oracle/oracle_ceres_3d.cc, statement for statement.
__global__ void __launch_bounds__(kCeres3DThreads)
Ceres3DKernel(const Ceres3DProblem* __restrict__ problems) {
  const Ceres3DProblem& P = problems[blockIdx.x];
  __shared__ double scratch[4][kNumSums];
  const double kFunctionTolerance = 1e-6, kGradientTolerance = 1e-10, kParameterTolerance = 1e-8;
  const double kMinRelativeDecrease = 1e-3, kMinLmDiagonal = 1e-6, kMaxLmDiagonal = 1e32;
  const double kMaxRadius = 1e16, kMinRadius = 1e-32;
  const int kMaxConsecutiveInvalidSteps = 5;
  const int max_consecutive_nonmonotonic_steps = P.use_nonmonotonic_steps ? 5 : 0;
  const bool yaw_only = P.yaw_only != 0;
  const int K = 3 + (yaw_only ? 1 : 3);

  double x[7];
  for (int a = 0; a < 7; ++a) x[a] = P.init[a];
  Eval3D at_x;
  Evaluate3D(P, x, K, scratch, &at_x);
  const double initial_cost = at_x.cost;
  double scale[kMaxLocal];
  for (int a = 0; a < K; ++a) scale[a] = 1. / (1. + sqrt(at_x.H[a][a]));
  const auto norm7 = [](const double* v) {
    double s = 0.;
    for (int a = 0; a < 7; ++a) s += v[a] * v[a];
    return sqrt(s);
  };
  // TrustRegionMinimizer::ComputeGradientMaxNorm: |x - Plus(x, -g)|_inf over the ambient
  // coordinates (`x` is always the point the evaluation was taken at).
  const auto gradient_max_norm = [K, yaw_only, &x](const Eval3D& e) {
    double m = 0.;
    for (int a = 0; a < 3; ++a) m = fmax(m, fabs(e.g[a]));
    double negative[3] = {0., 0., 0.}, projected[4];
    for (int a = 3; a < K; ++a) negative[a - 3] = -e.g[a];
    PlusRotation(yaw_only, x + 3, negative, projected);
    for (int a = 0; a < 4; ++a) m = fmax(m, fabs(x[3 + a] - projected[a]));
    return m;
  };
  double x_cost = at_x.cost, x_norm = norm7(x);
  double radius = 1e4, decrease_factor = 2.;
  bool reuse_diagonal = false;
  double diagonal[kMaxLocal] = {0., 0., 0., 0., 0., 0.};
  double minimum_cost = x_cost, current_cost = x_cost, reference_cost = x_cost,
         candidate_cost_eval = x_cost;
  double accumulated_reference_model_cost_change = 0., accumulated_candidate_model_cost_change = 0.;
  int num_consecutive_nonmonotonic_steps = 0, num_consecutive_invalid_steps = 0;
  double best_x[7];
  for (int a = 0; a < 7; ++a) best_x[a] = x[a];
  double best_cost = x_cost;
  int successful = 0, unsuccessful = 0;
  int termination = 1;   // NO_CONVERGENCE
  bool done = gradient_max_norm(at_x) <= kGradientTolerance;
  if (done) termination = 0;
  bool last_step_successful = false;
  for (int iteration = 1; !done; ++iteration) {
    if (iteration - 1 >= P.max_num_iterations) { termination = 1; break; }
    if (last_step_successful && gradient_max_norm(at_x) <= kGradientTolerance) {
      termination = 0;
      break;
    }
    if (radius < kMinRadius) { termination = 0; break; }
    last_step_successful = false;

    double Hs[kMaxLocal][kMaxLocal], gs[kMaxLocal];
    for (int a = 0; a < K; ++a) {
      gs[a] = at_x.g[a] * scale[a];
      for (int b = 0; b < K; ++b) Hs[a][b] = at_x.H[a][b] * scale[a] * scale[b];
    }
    if (!reuse_diagonal)
      for (int a = 0; a < K; ++a) diagonal[a] = fmin(fmax(Hs[a][a], kMinLmDiagonal), kMaxLmDiagonal);
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
      if (++num_consecutive_invalid_steps >= kMaxConsecutiveInvalidSteps) { termination = 2; break; }
      radius *= 0.5;
      reuse_diagonal = false;
      ++unsuccessful;
      continue;
    }
    num_consecutive_invalid_steps = 0;
    double delta[kMaxLocal], candidate[7];
    for (int a = 0; a < K; ++a) delta[a] = step[a] * scale[a];
    for (int a = 0; a < 3; ++a) candidate[a] = x[a] + delta[a];
    PlusRotation(yaw_only, x + 3, delta + 3, candidate + 3);
    Eval3D at_candidate;
    Evaluate3D(P, candidate, K, scratch, &at_candidate);
    const double candidate_cost = at_candidate.cost;

    double step_sq = 0.;
    for (int a = 0; a < 7; ++a) step_sq += (x[a] - candidate[a]) * (x[a] - candidate[a]);
    if (sqrt(step_sq) <= kParameterTolerance * (x_norm + kParameterTolerance)) {
      termination = 0;
      break;
    }
    const double cost_change = x_cost - candidate_cost;
    if (fabs(cost_change) <= kFunctionTolerance * x_cost) { termination = 0; break; }
    const double relative_decrease_now = (current_cost - candidate_cost) / model_cost_change;
    const double historical_relative_decrease =
        (reference_cost - candidate_cost) /
        (accumulated_reference_model_cost_change + model_cost_change);
    const double relative_decrease = fmax(relative_decrease_now, historical_relative_decrease);
    if (relative_decrease > kMinRelativeDecrease) {
      for (int a = 0; a < 7; ++a) x[a] = candidate[a];
      x_norm = norm7(x);
      at_x = at_candidate;
      x_cost = candidate_cost;
      last_step_successful = true;
      ++successful;
      if (x_cost < best_cost) {
        best_cost = x_cost;
        for (int a = 0; a < 7; ++a) best_x[a] = x[a];
      }
      // std::pow(t, 3) of the restatement is t * t * t for these arguments up to rounding;
      // pow is evaluated here as well so that both sides take the same value.
      radius = radius / fmax(1. / 3., 1. - pow(2. * relative_decrease - 1., 3.));
      radius = fmin(kMaxRadius, radius);
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
      ++unsuccessful;
    }
  }
  if (threadIdx.x == 0) {
    for (int a = 0; a < 7; ++a) P.out[a] = best_x[a];
    P.out[7] = initial_cost; P.out[8] = best_cost;
    P.out[9] = successful; P.out[10] = unsuccessful; P.out[11] = termination;
  }
}

// Checks of CeresScanMatcher3D's constructor / Match that do not depend on the pairs
// (ceres_scan_matcher_3d.cc:110,138,144).
void CheckOptions(const cmx_ceres3d_options& o) {
  CMX_REQUIRE(o.num_pairs >= 1 && o.num_pairs <= kMaxPairs, "num_pairs %d outside [1,%d]",
              o.num_pairs, kMaxPairs);
  CMX_REQUIRE(o.translation_weight > 0. && o.rotation_weight > 0.,
              "translation_weight and rotation_weight must be > 0");
  // common/internal/ceres_solver_options.cc:30 CHECK_GT (the 2D entry points check the same)
  CMX_REQUIRE(o.max_num_iterations > 0, "max_num_iterations must be > 0");
  for (int k = 0; k < o.num_pairs; ++k)
    CMX_REQUIRE(o.occupied_space_weight[k] > 0., "occupied_space_weight must be > 0");
}

// IntensityHybridGrid as a dense f32 brick of GetIntensity values (hybrid_grid.h:560-566: sum /
// count in f32, 0 where the cell holds no data; the division is done on the host).
struct IntensityCell { int32_t x, y, z; float value; };
__global__ void ScatterIntensityKernel(const IntensityCell* __restrict__ cells, long long n, Brick b,
                                       float* __restrict__ out) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const IntensityCell c = cells[i];
  const int ix = c.x - b.lo_x, iy = c.y - b.lo_y, iz = c.z - b.lo_z;
  out[(static_cast<size_t>(iz) * b.ny + iy) * b.nx + ix] = c.value;
}

void BuildIntensityBrick(Workspace& ws, const cmx_intensity_voxel* voxels, int64_t n,
                         DeviceBrick* out) {
  int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  std::vector<IntensityCell> cells;
  cells.reserve(static_cast<size_t>(n));
  for (int64_t i = 0; i < n; ++i) {
    const cmx_intensity_voxel& v = voxels[i];
    CMX_REQUIRE(v.count >= 0, "intensity voxel %lld has a negative count",
                static_cast<long long>(i));
    if (v.count == 0) continue;
    const int at[3] = {v.x, v.y, v.z};
    for (int k = 0; k < 3; ++k) {
      CMX_REQUIRE(at[k] > -(1 << 20) && at[k] < (1 << 20), "voxel index out of range");
      lo[k] = cells.empty() ? at[k] : std::min(lo[k], at[k]);
      hi[k] = cells.empty() ? at[k] : std::max(hi[k], at[k]);
    }
    cells.push_back({v.x, v.y, v.z, v.sum / static_cast<float>(v.count)});
  }
  Brick b{};
  b.lo_x = lo[0]; b.lo_y = lo[1]; b.lo_z = lo[2];
  b.nx = hi[0] - lo[0] + 1; b.ny = hi[1] - lo[1] + 1; b.nz = hi[2] - lo[2] + 1;
  const size_t num_cells = static_cast<size_t>(b.nx) * b.ny * b.nz;
  CMX_REQUIRE(num_cells * sizeof(float) < (size_t(8) << 30),
              "dense intensity grid of %d x %d x %d cells is too large", b.nx, b.ny, b.nz);
  out->bytes = num_cells * sizeof(float);
  CMX_HIP(hipMalloc(&out->mem, out->bytes + 16));
  b.cells = out->mem;
  out->desc = b;
  CMX_HIP(hipMemsetAsync(out->mem, 0, out->bytes, ws.stream));
  if (!cells.empty()) {
    IntensityCell* d_cells = ws.dev[15].ReserveAs<IntensityCell>(cells.size());
    CMX_HIP(hipMemcpyAsync(d_cells, cells.data(), cells.size() * sizeof(IntensityCell),
                           hipMemcpyHostToDevice, ws.stream));
    ScatterIntensityKernel<<<DivUp(static_cast<long long>(cells.size()), 256), 256, 0, ws.stream>>>(
        d_cells, static_cast<long long>(cells.size()), b, static_cast<float*>(out->mem));
    CMX_HIP(hipGetLastError());
  }
  CMX_HIP(hipStreamSynchronize(ws.stream));   // `cells` is host memory of this frame
}

void SetOptions(const cmx_ceres3d_options& o, Ceres3DProblem* P) {
  P->num_pairs = o.num_pairs;
  P->yaw_only = o.only_optimize_yaw ? 1 : 0;
  P->use_nonmonotonic_steps = o.use_nonmonotonic_steps ? 1 : 0;
  P->max_num_iterations = o.max_num_iterations;
  P->translation_weight = o.translation_weight;
  P->rotation_weight = o.rotation_weight;
}

// Uploads the clouds (already laid out in h_xyz) and `num` problem records, runs one workgroup
// per problem and brings the 12 result numbers of each back.  Problems' `out` pointers are set
// here.
void SolveProblems(Workspace& ws, Ceres3DProblem* problems, int num, size_t cloud_floats,
                   const float* h_xyz, float* d_xyz, double* results /*[num][12]*/) {
  const size_t prob_bytes = sizeof(Ceres3DProblem) * static_cast<size_t>(num);
  const size_t out_bytes = 12 * sizeof(double) * static_cast<size_t>(num);
  char* d_misc = static_cast<char*>(ws.dev[1].Reserve(prob_bytes + out_bytes));
  char* h_misc = static_cast<char*>(ws.pinned[1].Reserve(prob_bytes + out_bytes));
  double* d_out = reinterpret_cast<double*>(d_misc + prob_bytes);
  for (int p = 0; p < num; ++p) problems[p].out = d_out + 12 * static_cast<size_t>(p);
  std::memcpy(h_misc, problems, prob_bytes);
  SmallCopyAsync(d_xyz, h_xyz, sizeof(float) * cloud_floats, /*to_device=*/true, ws.stream);
  SmallCopyAsync(d_misc, h_misc, prob_bytes, /*to_device=*/true, ws.stream);
  RecordEvent(ws.ev_begin, ws.stream);
  Ceres3DKernel<<<num, kCeres3DThreads, 0, ws.stream>>>(
      reinterpret_cast<const Ceres3DProblem*>(d_misc));
  CMX_HIP(hipGetLastError());
  RecordEvent(ws.ev_end, ws.stream);
  double* h_out = reinterpret_cast<double*>(h_misc + prob_bytes);
  SmallCopyAsync(h_out, d_out, out_bytes, /*to_device=*/false, ws.stream);
  CMX_HIP(hipStreamSynchronize(ws.stream));
  std::memcpy(results, h_out, out_bytes);
}

void WriteResult(const double* out, cmx_pose3d* pose_estimate, cmx_ceres_summary* summary) {
  for (int a = 0; a < 3; ++a) pose_estimate->t[a] = out[a];
  for (int a = 0; a < 4; ++a) pose_estimate->q[a] = out[3 + a];
  if (summary) {
    summary->initial_cost = out[7];
    summary->final_cost = out[8];
    summary->num_successful_steps = static_cast<int32_t>(out[9]);
    summary->num_unsuccessful_steps = static_cast<int32_t>(out[10]);
    summary->termination = static_cast<int32_t>(out[11]);
    summary->reserved = 0;
  }
}

}  // namespace
}  // namespace cmx

extern "C" cmx_status cmx_ceres3d_match(const cmx_ceres3d_options* options,
                                        const double* target_translation_xyz,
                                        const cmx_pose3d* initial_pose_estimate,
                                        const cmx_ceres3d_pair* pairs, int32_t device,
                                        cmx_pose3d* pose_estimate, cmx_ceres_summary* summary) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(options && target_translation_xyz && initial_pose_estimate && pairs, "null argument");
    CMX_REQUIRE(pose_estimate != nullptr, "pose_estimate must not be null");
    CheckOptions(*options);
    WorkspaceLease ws(device);
    std::vector<std::unique_ptr<DeviceBrick>> bricks;
    Ceres3DProblem P{};
    SetOptions(*options, &P);
    size_t cloud_floats = 0;
    for (int k = 0; k < P.num_pairs; ++k) {
      const cmx_ceres3d_pair& in = pairs[k];
      CMX_REQUIRE(in.point_cloud_xyz && in.num_points >= 1 && in.num_points <= (1 << 24),
                  "bad point cloud %d", k);
      CMX_REQUIRE(in.resolution > 0.f, "resolution must be > 0");
      CMX_REQUIRE(in.num_voxels == 0 || in.voxels != nullptr, "voxels is null");
      cloud_floats += 3 * static_cast<size_t>(in.num_points);
      if (in.intensities != nullptr) {
        // CHECKs of CreateIntensityCostFunctionOptions' consumers (ceres_scan_matcher_3d.cc:
        // 118-137, intensity_cost_function_3d.h:41-47) and of ceres::HuberLoss (a > 0).
        CMX_REQUIRE(in.num_intensity_voxels >= 0 &&
                        (in.num_intensity_voxels == 0 || in.intensity_voxels != nullptr),
                    "intensity_voxels is null");
        CMX_REQUIRE(in.intensity_weight > 0., "intensity weight must be > 0");
        CMX_REQUIRE(in.intensity_huber_scale > 0., "intensity huber_scale must be > 0");
        CMX_REQUIRE(in.intensity_threshold > 0.f, "intensity_threshold must be > 0");
        cloud_floats += static_cast<size_t>(in.num_points);
      }
    }
    float* d_xyz = ws->dev[0].ReserveAs<float>(cloud_floats);
    float* h_xyz = ws->pinned[0].ReserveAs<float>(cloud_floats);
    size_t off = 0;
    for (int k = 0; k < P.num_pairs; ++k) {
      const cmx_ceres3d_pair& in = pairs[k];
      bricks.emplace_back(new DeviceBrick);
      BuildBrickFromVoxels(*ws, in.voxels, in.num_voxels, 2, bricks.back().get());
      std::memcpy(h_xyz + off, in.point_cloud_xyz, 3 * sizeof(float) * in.num_points);
      P.pair[k].grid = bricks.back()->desc;
      P.pair[k].resolution = in.resolution;
      P.pair[k].n = in.num_points;
      P.pair[k].xyz = d_xyz + off;
      P.pair[k].scaling =
          options->occupied_space_weight[k] / std::sqrt(static_cast<double>(in.num_points));
      off += 3 * static_cast<size_t>(in.num_points);
      if (in.intensities != nullptr) {
        bricks.emplace_back(new DeviceBrick);
        BuildIntensityBrick(*ws, in.intensity_voxels, in.num_intensity_voxels, bricks.back().get());
        std::memcpy(h_xyz + off, in.intensities, sizeof(float) * in.num_points);
        P.pair[k].has_intensity = 1;
        P.pair[k].intensity_threshold = in.intensity_threshold;
        P.pair[k].igrid = bricks.back()->desc;
        P.pair[k].intensities = d_xyz + off;
        P.pair[k].iscaling = in.intensity_weight / std::sqrt(static_cast<double>(in.num_points));
        P.pair[k].huber_a = in.intensity_huber_scale;
        off += static_cast<size_t>(in.num_points);
      }
    }
    for (int a = 0; a < 3; ++a) {
      P.target[a] = target_translation_xyz[a];
      P.init[a] = initial_pose_estimate->t[a];
    }
    for (int a = 0; a < 4; ++a) P.init[3 + a] = initial_pose_estimate->q[a];
    double out[12];
    SolveProblems(*ws, &P, 1, cloud_floats, h_xyz, d_xyz, out);
    WriteResult(out, pose_estimate, summary);
  });
}

// LocalTrajectoryBuilder3D::ScanMatch's refinement (local_trajectory_builder_3d.cc:96-123) against
// the ACTIVE submap's HybridGrids where cmx_grid3d keeps them in HBM: no upload, no allocation.
extern "C" cmx_status cmx_ceres3d_match_grids(const cmx_ceres3d_options* options,
                                              const double* target_translation_xyz,
                                              const cmx_pose3d* initial_pose_estimate,
                                              const cmx_grid3d* const* grids,
                                              const float* const* point_clouds_xyz,
                                              const int32_t* num_points,
                                              cmx_pose3d* pose_estimate,
                                              cmx_ceres_summary* summary) {
  return cmx_ceres3d_match_grids_intensity(options, target_translation_xyz, initial_pose_estimate,
                                           grids, point_clouds_xyz, num_points, nullptr,
                                           pose_estimate, summary);
}

extern "C" cmx_status cmx_ceres3d_match_grids_intensity(
    const cmx_ceres3d_options* options, const double* target_translation_xyz,
    const cmx_pose3d* initial_pose_estimate, const cmx_grid3d* const* grids,
    const float* const* point_clouds_xyz, const int32_t* num_points,
    const cmx_ceres3d_intensity_term* terms, cmx_pose3d* pose_estimate,
    cmx_ceres_summary* summary) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(options && target_translation_xyz && initial_pose_estimate && grids &&
                    point_clouds_xyz && num_points,
                "null argument");
    CMX_REQUIRE(pose_estimate != nullptr, "pose_estimate must not be null");
    CheckOptions(*options);
    Ceres3DProblem P{};
    SetOptions(*options, &P);
    size_t cloud_floats = 0;
    int device = -1;
    bool any_empty = false;
    for (int k = 0; k < P.num_pairs; ++k) {
      CMX_REQUIRE(grids[k] != nullptr, "grid %d is null", k);
      CMX_REQUIRE(point_clouds_xyz[k] && num_points[k] >= 1 && num_points[k] <= (1 << 24),
                  "bad point cloud %d", k);
      int d = 0;
      if (!Grid3DBrick(grids[k], &P.pair[k].grid, &P.pair[k].resolution, &d)) any_empty = true;
      CMX_REQUIRE(device < 0 || d == device, "the grids live on different devices");
      device = d;
      cloud_floats += 3 * static_cast<size_t>(num_points[k]);
      if (terms && terms[k].grid) {
        // CHECKs of CreateIntensityCostFunctionOptions' consumers (ceres_scan_matcher_3d.cc:
        // 118-137, intensity_cost_function_3d.h:41-47) and of ceres::HuberLoss (a > 0).
        CMX_REQUIRE(terms[k].intensities != nullptr, "intensities of pair %d are null", k);
        CMX_REQUIRE(terms[k].weight > 0., "intensity weight must be > 0");
        CMX_REQUIRE(terms[k].huber_scale > 0., "intensity huber_scale must be > 0");
        CMX_REQUIRE(terms[k].intensity_threshold > 0.f, "intensity_threshold must be > 0");
        cloud_floats += static_cast<size_t>(num_points[k]);
      }
    }
    WorkspaceLease ws(device);
    float* zero_intensity = nullptr;            // an intensity grid nothing was inserted into
    for (int k = 0; terms && k < P.num_pairs; ++k) {
      if (!terms[k].grid) continue;
      float res = 0.f;
      int d = 0;
      if (!IntensityGrid3DBrick(terms[k].grid, ws->stream, &P.pair[k].igrid, &res, &d)) {
        if (!zero_intensity) {
          zero_intensity = ws->dev[3].ReserveAs<float>(4);
          CMX_HIP(hipMemsetAsync(zero_intensity, 0, 16, ws->stream));
        }
        P.pair[k].igrid = Brick{};
        P.pair[k].igrid.cells = zero_intensity;
        P.pair[k].igrid.nx = P.pair[k].igrid.ny = P.pair[k].igrid.nz = 1;
      }
      CMX_REQUIRE(d == device, "the grids live on different devices");
      CMX_REQUIRE(res == P.pair[k].resolution,
                  "the intensity grid of pair %d must have its hybrid grid's resolution", k);
    }
    if (any_empty) {
      // A HybridGrid nothing was inserted into: every lookup is value 0 (kMinProbability).
      uint16_t* zero = ws->dev[2].ReserveAs<uint16_t>(8);
      CMX_HIP(hipMemsetAsync(zero, 0, 16, ws->stream));
      for (int k = 0; k < P.num_pairs; ++k) {
        if (P.pair[k].grid.cells != nullptr) continue;
        P.pair[k].grid = Brick{};
        P.pair[k].grid.cells = zero;
        P.pair[k].grid.nx = P.pair[k].grid.ny = P.pair[k].grid.nz = 1;
      }
    }
    float* d_xyz = ws->dev[0].ReserveAs<float>(cloud_floats);
    float* h_xyz = ws->pinned[0].ReserveAs<float>(cloud_floats);
    size_t off = 0;
    for (int k = 0; k < P.num_pairs; ++k) {
      std::memcpy(h_xyz + off, point_clouds_xyz[k], 3 * sizeof(float) * num_points[k]);
      P.pair[k].n = num_points[k];
      P.pair[k].xyz = d_xyz + off;
      P.pair[k].scaling =
          options->occupied_space_weight[k] / std::sqrt(static_cast<double>(num_points[k]));
      off += 3 * static_cast<size_t>(num_points[k]);
      if (terms && terms[k].grid) {
        std::memcpy(h_xyz + off, terms[k].intensities, sizeof(float) * num_points[k]);
        P.pair[k].has_intensity = 1;
        P.pair[k].intensity_threshold = terms[k].intensity_threshold;
        P.pair[k].intensities = d_xyz + off;
        P.pair[k].iscaling = terms[k].weight / std::sqrt(static_cast<double>(num_points[k]));
        P.pair[k].huber_a = terms[k].huber_scale;
        off += static_cast<size_t>(num_points[k]);
      }
    }
    for (int a = 0; a < 3; ++a) {
      P.target[a] = target_translation_xyz[a];
      P.init[a] = initial_pose_estimate->t[a];
    }
    for (int a = 0; a < 4; ++a) P.init[3 + a] = initial_pose_estimate->q[a];
    double out[12];
    SolveProblems(*ws, &P, 1, cloud_floats, h_xyz, d_xyz, out);
    WriteResult(out, pose_estimate, summary);
  });
}

// ConstraintBuilder3D::ComputeConstraint's refinement (constraints/constraint_builder_3d.cc:
// 263-276) for a node's batch: one workgroup per found pair, grids already in HBM.
extern "C" cmx_status cmx_fast3d_refine_batch(const cmx_ceres3d_options* options,
                                              const cmx_fast3d* const* matchers,
                                              int32_t num_pairs, const int32_t* found,
                                              const cmx_pose3d* pose_estimates_in,
                                              const cmx_node_data3d* data,
                                              cmx_pose3d* pose_estimates_out,
                                              cmx_ceres_summary* summaries) {
  using namespace cmx;
  return Guard([&] {
    CMX_REQUIRE(options && data && pose_estimates_in && pose_estimates_out, "null argument");
    CMX_REQUIRE(num_pairs >= 0 && (num_pairs == 0 || matchers), "bad matcher list");
    CheckOptions(*options);
    // {high-resolution cloud, high-resolution grid}, {low-resolution cloud, low-resolution grid}
    CMX_REQUIRE(options->num_pairs == 2, "the constraint refinement uses two (cloud, grid) pairs");
    CMX_REQUIRE(data->high_resolution_point_cloud && data->num_high_resolution_points >= 1 &&
                    data->low_resolution_point_cloud && data->num_low_resolution_points >= 1 &&
                    data->num_high_resolution_points <= (1 << 24) &&
                    data->num_low_resolution_points <= (1 << 24),
                "bad point clouds");
    for (int p = 0; p < num_pairs; ++p) {
      CMX_REQUIRE(matchers[p] != nullptr, "matcher %d is null", p);
      pose_estimates_out[p] = pose_estimates_in[p];        // not found: passed through
      if (summaries) summaries[p] = cmx_ceres_summary{};
    }
    const int n_hi = data->num_high_resolution_points, n_lo = data->num_low_resolution_points;
    const size_t cloud_floats = 3 * (static_cast<size_t>(n_hi) + n_lo);
    // Pairs are grouped by the device their grids live on (a node's batch may span the GPUs
    // of a cmx_comm); each group is one launch.
    std::vector<int> order;
    for (int p = 0; p < num_pairs; ++p)
      if (!found || found[p]) order.push_back(p);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      return Fast3DDevice(matchers[a]) < Fast3DDevice(matchers[b]);
    });
    for (size_t begin = 0; begin < order.size();) {
      const int device = Fast3DDevice(matchers[order[begin]]);
      size_t end = begin;
      while (end < order.size() && Fast3DDevice(matchers[order[end]]) == device) ++end;
      const int num = static_cast<int>(end - begin);
      WorkspaceLease ws(device);
      float* d_xyz = ws->dev[0].ReserveAs<float>(cloud_floats);
      float* h_xyz = ws->pinned[0].ReserveAs<float>(cloud_floats);
      std::memcpy(h_xyz, data->high_resolution_point_cloud, 3 * sizeof(float) * n_hi);
      std::memcpy(h_xyz + 3 * static_cast<size_t>(n_hi), data->low_resolution_point_cloud,
                  3 * sizeof(float) * n_lo);
      std::vector<Ceres3DProblem> problems(num);
      for (int k = 0; k < num; ++k) {
        const int p = order[begin + k];
        Ceres3DProblem& P = problems[k];
        SetOptions(*options, &P);
        Fast3DGrids(matchers[p], &P.pair[0].grid, &P.pair[0].resolution, &P.pair[1].grid,
                    &P.pair[1].resolution);
        P.pair[0].n = n_hi;
        P.pair[0].xyz = d_xyz;
        P.pair[0].scaling = options->occupied_space_weight[0] / std::sqrt(static_cast<double>(n_hi));
        P.pair[1].n = n_lo;
        P.pair[1].xyz = d_xyz + 3 * static_cast<size_t>(n_hi);
        P.pair[1].scaling = options->occupied_space_weight[1] / std::sqrt(static_cast<double>(n_lo));
        // Match(match_result->pose_estimate.translation(), match_result->pose_estimate, ...)
        for (int a = 0; a < 3; ++a) P.target[a] = P.init[a] = pose_estimates_in[p].t[a];
        for (int a = 0; a < 4; ++a) P.init[3 + a] = pose_estimates_in[p].q[a];
      }
      std::vector<double> out(12 * static_cast<size_t>(num));
      SolveProblems(*ws, problems.data(), num, cloud_floats, h_xyz, d_xyz, out.data());
      for (int k = 0; k < num; ++k) {
        const int p = order[begin + k];
        WriteResult(out.data() + 12 * static_cast<size_t>(k), pose_estimates_out + p,
                    summaries ? summaries + p : nullptr);
      }
      begin = end;
    }
  });
}
