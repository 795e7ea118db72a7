// CeresScanMatcher2D::Match on gfx950: the refinement step that follows
// every correlative match in both callers
//   mapping/internal/2d/local_trajectory_builder_2d.cc:104-107 (after the real-time matcher)
//   mapping/internal/constraints/constraint_builder_2d.cc:245-249 (after the fast matcher)
// so that a constraint no longer leaves HBM between the correlative search and the refined
// pose.
//
// Reference: SM2/ceres_scan_matcher_2d.cc:63-107, SM2/occupied_space_cost_function_2d.cc:39-108
// (bicubic interpolation of the correspondence costs at the transformed points),
// SM2/translation_delta_cost_functor_2d.h:42-47, SM2/rotation_delta_cost_functor_2d.h:42-45.
// The least-squares solver is Ceres (third party, absent from the reference tree): what runs
// here is its published trust-region Levenberg-Marquardt loop with Solver::Options defaults,
// restated the same way as oracle/oracle_ceres_2d.cc (which documents every rule taken from
// Ceres and what pins it).  Three unknowns, so the damped system is a 3 x 3 solve; all the
// work is in the residual blocks.
//
// One workgroup per problem.  Per evaluation every thread takes a few points: the 16
// correspondence costs around the point (uint16 -> f32 table expression -> f64, gathered
// from the grid in HBM / L2), two Catmull-Rom passes, the residual and its three partials,
// accumulated as cost, J^T r and J^T J in f64 (ten numbers per thread), reduced across the
// wavefront with DPP-free shuffles and across the four wavefronts through LDS in a FIXED
// order -- results do not depend on scheduling.  Thread 0 then runs the trust-region logic.
// Residual + Jacobian are evaluated together at every candidate (the Jacobian of a rejected
// candidate is wasted; evaluating it separately would be a second pass over the points).
#include <climits>
#include <cmath>
#include <map>
#include <vector>

#include "scan_matching_2d.h"

struct cmx_grid2d;   // grid_2d.hip
namespace cmx {
const uint16_t* Grid2DDeviceCells(const cmx_grid2d* grid, cmx_grid2d_limits* limits, int* device);
}

namespace cmx {
namespace {

constexpr int kCeresThreads = 256;
constexpr int kPadding = INT_MAX / 4;   // occupied_space_cost_function_2d.cc:78

struct Ceres2DProblem {
  const uint16_t* cells;       // device grid
  int nx, ny;
  double res, max_x, max_y;
  float min_cc, max_cc;        // the grid's correspondence cost range (value table)
  const float* xyz;            // device cloud
  int n;
  double init[3];              // initial pose estimate (x, y, theta)
  double target_x, target_y;   // target translation; the target angle is init[2]
  double occupied_scaling;     // occupied_space_weight / sqrt(n)
  double translation_weight, rotation_weight;
  int use_nonmonotonic_steps, max_num_iterations;
  int skip;                    // 1: no search result to refine (found == 0): pass through
  double* out;                 // [8]: pose x, y, theta, initial cost, final cost, successful,
                               //      unsuccessful, termination
};

// Grid2D::GetCorrespondenceCost through the per-grid table of
// mapping/value_conversion_tables.cc:29-51, evaluated arithmetically (same f32 expression).
__device__ __forceinline__ double CellCost(const Ceres2DProblem& P, int ix, int iy) {
  const bool inside = static_cast<unsigned>(ix) < static_cast<unsigned>(P.nx) &&
                      static_cast<unsigned>(iy) < static_cast<unsigned>(P.ny);
  const unsigned v = AsGlobal(P.cells)[inside ? P.nx * iy + ix : 0] & 0x7fffu;
  float cost = P.max_cc;
  if (inside && v != 0) {
    const float scale = (P.max_cc - P.min_cc) / 32766.f;
    cost = static_cast<float>(v) * scale + (P.min_cc - scale);
  }
  return static_cast<double>(cost);
}
// GridArrayAdapter::GetValue: kMaxCorrespondenceCost outside (the constant, not the grid's).
__device__ __forceinline__ double AdapterValue(const Ceres2DProblem& P, int row, int column) {
  const int ix = column - kPadding, iy = row - kPadding;
  const bool inside = static_cast<unsigned>(ix) < static_cast<unsigned>(P.nx) &&
                      static_cast<unsigned>(iy) < static_cast<unsigned>(P.ny);
  const double inner = CellCost(P, ix, iy);
  return inside ? inner : static_cast<double>(1.f - 0.1f);
}

// ceres::CubicHermiteSpline<1>.
__device__ __forceinline__ void Spline(double p0, double p1, double p2, double p3, double x,
                                       double* f, double* dfdx) {
  const double a = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3);
  const double b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3);
  const double c = 0.5 * (-p0 + p2);
  const double d = p1;
  *f = d + x * (c + x * (b + x * a));
  *dfdx = c + x * (2.0 * b + 3.0 * a * x);
}

struct Sums {     // cost (sum of squares), g = J^T r, H = J^T J (upper triangle)
  double v[10];
};

__device__ __forceinline__ double WaveSumF64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Residual blocks at `x`: leaves 1/2 |r|^2, J^T r and J^T J in `e` (valid in every thread).
__device__ void Evaluate(const Ceres2DProblem& P, const double x[3], double (*scratch)[10],
                         double* cost, double g[3], double H[3][3]) {
  const double c = cos(x[2]), s = sin(x[2]);
  Sums acc;
#pragma unroll
  for (int k = 0; k < 10; ++k) acc.v[k] = 0.;
  for (int i = threadIdx.x; i < P.n; i += kCeresThreads) {
    const double px = static_cast<double>(P.xyz[3 * i]), py = static_cast<double>(P.xyz[3 * i + 1]);
    const double wx = c * px + -s * py + x[0] * 1.;
    const double wy = s * px + c * py + x[1] * 1.;
    const double r = (P.max_x - wx) / P.res - 0.5 + static_cast<double>(kPadding);
    const double cc = (P.max_y - wy) / P.res - 0.5 + static_cast<double>(kPadding);
    const int row = static_cast<int>(floor(r)), col = static_cast<int>(floor(cc));
    double fr[4], dfr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rr = row - 1 + k;
      Spline(AdapterValue(P, rr, col - 1), AdapterValue(P, rr, col), AdapterValue(P, rr, col + 1),
             AdapterValue(P, rr, col + 2), cc - col, &fr[k], &dfr[k]);
    }
    double f, dfdr, dfdc, unused;
    Spline(fr[0], fr[1], fr[2], fr[3], r - row, &f, &dfdr);
    Spline(dfr[0], dfr[1], dfr[2], dfr[3], r - row, &dfdc, &unused);
    const double res_i = P.occupied_scaling * f;
    const double dwx_dt = -s * px - c * py, dwy_dt = c * px - s * py;
    const double j0 = P.occupied_scaling * (dfdr * (-1. / P.res));
    const double j1 = P.occupied_scaling * (dfdc * (-1. / P.res));
    const double j2 = P.occupied_scaling * (dfdr * (-dwx_dt / P.res) + dfdc * (-dwy_dt / P.res));
    acc.v[0] += res_i * res_i;
    acc.v[1] += j0 * res_i; acc.v[2] += j1 * res_i; acc.v[3] += j2 * res_i;
    acc.v[4] += j0 * j0; acc.v[5] += j0 * j1; acc.v[6] += j0 * j2;
    acc.v[7] += j1 * j1; acc.v[8] += j1 * j2; acc.v[9] += j2 * j2;
  }
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const double t = WaveSumF64(acc.v[k]);
    if ((threadIdx.x & 63) == 0) scratch[wave][k] = t;
  }
  __syncthreads();
  double total[10];
#pragma unroll
  for (int k = 0; k < 10; ++k)
    total[k] = ((scratch[0][k] + scratch[1][k]) + scratch[2][k]) + scratch[3][k];
  __syncthreads();
  // TranslationDeltaCostFunctor2D / RotationDeltaCostFunctor2D.
  const double rt0 = P.translation_weight * (x[0] - P.target_x);
  const double rt1 = P.translation_weight * (x[1] - P.target_y);
  const double rr = P.rotation_weight * (x[2] - P.init[2]);
  total[0] += rt0 * rt0; total[0] += rt1 * rt1; total[0] += rr * rr;
  total[1] += P.translation_weight * rt0;
  total[2] += P.translation_weight * rt1;
  total[3] += P.rotation_weight * rr;
  total[4] += P.translation_weight * P.translation_weight;
  total[7] += P.translation_weight * P.translation_weight;
  total[9] += P.rotation_weight * P.rotation_weight;
  *cost = 0.5 * total[0];
  g[0] = total[1]; g[1] = total[2]; g[2] = total[3];
  H[0][0] = total[4]; H[0][1] = H[1][0] = total[5]; H[0][2] = H[2][0] = total[6];
  H[1][1] = total[7]; H[1][2] = H[2][1] = total[8]; H[2][2] = total[9];
}

// Cholesky solve of the symmetric positive definite 3 x 3 system; false if not SPD.
__device__ bool SolveSpd3(const double A[3][3], const double b[3], double x[3]) {
  double L[3][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j <= i; ++j) {
      double sum = A[i][j];
      for (int k = 0; k < j; ++k) sum -= L[i][k] * L[j][k];
      if (i == j) {
        if (!(sum > 0.)) return false;
        L[i][i] = sqrt(sum);
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
  return isfinite(x[0]) && isfinite(x[1]) && isfinite(x[2]);
}

// The trust-region loop of oracle/oracle_ceres_2d.cc (Ceres: trust_region_minimizer.cc,
// levenberg_marquardt_strategy.cc, trust_region_step_evaluator.cc), statement for statement.
// Every thread carries the (identical) minimizer state: no broadcast is needed, the block
// only meets inside Evaluate.
__global__ void __launch_bounds__(kCeresThreads)
Ceres2DKernel(const Ceres2DProblem* __restrict__ problems) {
  const Ceres2DProblem& P = problems[blockIdx.x];
  __shared__ double scratch[4][10];
  if (P.skip) {
    if (threadIdx.x == 0) {
      P.out[0] = P.init[0]; P.out[1] = P.init[1]; P.out[2] = P.init[2];
      P.out[3] = 0.; P.out[4] = 0.; P.out[5] = 0.; P.out[6] = 0.; P.out[7] = 1.;
    }
    return;
  }
  const double kFunctionTolerance = 1e-6, kGradientTolerance = 1e-10, kParameterTolerance = 1e-8;
  const double kMinRelativeDecrease = 1e-3, kMinLmDiagonal = 1e-6, kMaxLmDiagonal = 1e32;
  const double kMaxRadius = 1e16, kMinRadius = 1e-32;
  const int kMaxConsecutiveInvalidSteps = 5;
  const int max_consecutive_nonmonotonic_steps = P.use_nonmonotonic_steps ? 5 : 0;

  double x[3] = {P.init[0], P.init[1], P.init[2]};
  double x_cost, g[3], H[3][3];
  Evaluate(P, x, scratch, &x_cost, g, H);
  const double initial_cost = x_cost;
  double scale[3];
  for (int a = 0; a < 3; ++a) scale[a] = 1. / (1. + sqrt(H[a][a]));
  double x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  double radius = 1e4, decrease_factor = 2.;
  bool reuse_diagonal = false;
  double diagonal[3] = {0., 0., 0.};
  double minimum_cost = x_cost, current_cost = x_cost, reference_cost = x_cost,
         candidate_cost_eval = x_cost;
  double accumulated_reference_model_cost_change = 0., accumulated_candidate_model_cost_change = 0.;
  int num_consecutive_nonmonotonic_steps = 0, num_consecutive_invalid_steps = 0;
  double best_x[3] = {x[0], x[1], x[2]};
  double best_cost = x_cost;
  int successful = 0, unsuccessful = 0;
  int termination = 1;   // NO_CONVERGENCE
  const auto gradient_max_norm = [&]() {
    return fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
  };
  bool done = gradient_max_norm() <= kGradientTolerance;
  if (done) termination = 0;
  bool last_step_successful = false;
  for (int iteration = 1; !done; ++iteration) {
    if (iteration - 1 >= P.max_num_iterations) { termination = 1; break; }
    if (last_step_successful && gradient_max_norm() <= kGradientTolerance) { termination = 0; break; }
    if (radius < kMinRadius) { termination = 0; break; }
    last_step_successful = false;

    double Hs[3][3], gs[3];
    for (int a = 0; a < 3; ++a) {
      gs[a] = g[a] * scale[a];
      for (int b = 0; b < 3; ++b) Hs[a][b] = H[a][b] * scale[a] * scale[b];
    }
    if (!reuse_diagonal) {
      for (int a = 0; a < 3; ++a) diagonal[a] = fmin(fmax(Hs[a][a], kMinLmDiagonal), kMaxLmDiagonal);
    }
    double A[3][3], step[3];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) A[a][b] = Hs[a][b] + (a == b ? diagonal[a] / radius : 0.);
    const bool solved = SolveSpd3(A, gs, step);
    for (int a = 0; a < 3; ++a) step[a] = -step[a];
    reuse_diagonal = true;
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
      if (++num_consecutive_invalid_steps >= kMaxConsecutiveInvalidSteps) { termination = 2; break; }
      radius *= 0.5;
      reuse_diagonal = false;
      ++unsuccessful;
      continue;
    }
    num_consecutive_invalid_steps = 0;
    double delta[3], candidate[3];
    for (int a = 0; a < 3; ++a) {
      delta[a] = step[a] * scale[a];
      candidate[a] = x[a] + delta[a];
    }
    double candidate_cost, cg[3], cH[3][3];
    Evaluate(P, candidate, scratch, &candidate_cost, cg, cH);

    const double step_norm = sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
    if (step_norm <= kParameterTolerance * (x_norm + kParameterTolerance)) { termination = 0; break; }
    const double cost_change = x_cost - candidate_cost;
    if (fabs(cost_change) <= kFunctionTolerance * x_cost) { termination = 0; break; }
    const double relative_decrease_now = (current_cost - candidate_cost) / model_cost_change;
    const double historical_relative_decrease =
        (reference_cost - candidate_cost) /
        (accumulated_reference_model_cost_change + model_cost_change);
    const double relative_decrease = fmax(relative_decrease_now, historical_relative_decrease);
    if (relative_decrease > kMinRelativeDecrease) {
      for (int a = 0; a < 3; ++a) {
        x[a] = candidate[a];
        g[a] = cg[a];
        for (int b = 0; b < 3; ++b) H[a][b] = cH[a][b];
      }
      x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
      x_cost = candidate_cost;
      last_step_successful = true;
      ++successful;
      if (x_cost < best_cost) {
        best_cost = x_cost;
        for (int a = 0; a < 3; ++a) best_x[a] = x[a];
      }
      const double t = 2. * relative_decrease - 1.;
      radius = radius / fmax(1. / 3., 1. - t * t * t);
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
    P.out[0] = best_x[0]; P.out[1] = best_x[1]; P.out[2] = best_x[2];
    P.out[3] = initial_cost; P.out[4] = best_cost;
    P.out[5] = successful; P.out[6] = unsuccessful; P.out[7] = termination;
  }
}

struct RefineItem {
  const uint16_t* device_cells;     // grid already in HBM, or null with host_cells
  const uint16_t* host_cells;
  cmx_grid2d_limits limits;
  double target[2];
  cmx_pose2d initial;
  int skip;
};

void CheckOptions(const cmx_ceres2d_options* o) {
  // CHECK_GT at ceres_scan_matcher_2d.cc:73,91,96; CreateCeresSolverOptionsProto's CHECK_GT.
  CMX_REQUIRE(o != nullptr, "null options");
  CMX_REQUIRE(o->occupied_space_weight > 0. && o->translation_weight > 0. &&
                  o->rotation_weight > 0.,
              "occupied_space / translation / rotation weights must be > 0");
  CMX_REQUIRE(o->max_num_iterations > 0, "max_num_iterations must be > 0");
}

// One launch for `num` problems sharing one point cloud; device_xyz may be null (host_xyz is
// uploaded then).
void RefineBatch(const cmx_ceres2d_options* options, const RefineItem* items, int num,
                 const float* host_xyz, int n, int device, cmx_pose2d* poses,
                 cmx_ceres_summary* summaries) {
  CheckOptions(options);
  CMX_REQUIRE(items && num >= 1 && poses, "null argument");
  CMX_REQUIRE(host_xyz != nullptr && n >= 1 && n <= (1 << 24), "bad point cloud");
  WorkspaceLease ws(device);
  // Staging: problems | cloud | host grids.
  const auto align = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  size_t bytes = align(sizeof(Ceres2DProblem) * num);
  const size_t off_xyz = bytes;
  bytes += align(12 * static_cast<size_t>(n));
  std::vector<size_t> off_cells(num, 0);
  for (int p = 0; p < num; ++p) {
    const cmx_grid2d_limits& lim = items[p].limits;
    CMX_REQUIRE(lim.resolution > 0. && lim.num_x_cells >= 1 && lim.num_y_cells >= 1,
                "bad map limits");
    CMX_REQUIRE(items[p].device_cells || items[p].host_cells, "null grid");
    if (!items[p].device_cells) {
      off_cells[p] = bytes;
      bytes += align(2 * static_cast<size_t>(lim.num_x_cells) * lim.num_y_cells);
    }
  }
  char* h_in = ws->pinned[0].ReserveAs<char>(bytes);
  char* d_in = ws->dev[0].ReserveAs<char>(bytes);
  double* d_out = ws->dev[1].ReserveAs<double>(8 * static_cast<size_t>(num));
  double* h_out = ws->pinned[1].ReserveAs<double>(8 * static_cast<size_t>(num));
  std::memcpy(h_in + off_xyz, host_xyz, 12 * static_cast<size_t>(n));
  Ceres2DProblem* h_prob = reinterpret_cast<Ceres2DProblem*>(h_in);
  for (int p = 0; p < num; ++p) {
    const RefineItem& it = items[p];
    Ceres2DProblem P{};
    if (it.device_cells) {
      P.cells = it.device_cells;
    } else {
      std::memcpy(h_in + off_cells[p], it.host_cells,
                  2 * static_cast<size_t>(it.limits.num_x_cells) * it.limits.num_y_cells);
      P.cells = reinterpret_cast<const uint16_t*>(d_in + off_cells[p]);
    }
    P.nx = it.limits.num_x_cells; P.ny = it.limits.num_y_cells;
    P.res = it.limits.resolution; P.max_x = it.limits.max_x; P.max_y = it.limits.max_y;
    P.min_cc = it.limits.min_correspondence_cost; P.max_cc = it.limits.max_correspondence_cost;
    P.xyz = reinterpret_cast<const float*>(d_in + off_xyz);
    P.n = n;
    P.init[0] = it.initial.x; P.init[1] = it.initial.y; P.init[2] = it.initial.theta;
    P.target_x = it.target[0]; P.target_y = it.target[1];
    P.occupied_scaling = options->occupied_space_weight / std::sqrt(static_cast<double>(n));
    P.translation_weight = options->translation_weight;
    P.rotation_weight = options->rotation_weight;
    P.use_nonmonotonic_steps = options->use_nonmonotonic_steps ? 1 : 0;
    P.max_num_iterations = options->max_num_iterations;
    P.skip = it.skip;
    P.out = d_out + 8 * static_cast<size_t>(p);
    h_prob[p] = P;
  }
  SmallCopyAsync(d_in, h_in, bytes, /*to_device=*/true, ws->stream);
  Ceres2DKernel<<<num, kCeresThreads, 0, ws->stream>>>(reinterpret_cast<const Ceres2DProblem*>(d_in));
  CMX_HIP(hipGetLastError());
  SmallCopyAsync(h_out, d_out, 64 * static_cast<size_t>(num), /*to_device=*/false, ws->stream);
  CMX_HIP(hipStreamSynchronize(ws->stream));
  for (int p = 0; p < num; ++p) {
    const double* o = h_out + 8 * static_cast<size_t>(p);
    poses[p].x = o[0]; poses[p].y = o[1]; poses[p].theta = o[2];
    if (summaries) {
      summaries[p].initial_cost = o[3];
      summaries[p].final_cost = o[4];
      summaries[p].num_successful_steps = static_cast<int32_t>(o[5]);
      summaries[p].num_unsuccessful_steps = static_cast<int32_t>(o[6]);
      summaries[p].termination = static_cast<int32_t>(o[7]);
      summaries[p].reserved = 0;
    }
  }
}

}  // namespace
}  // namespace cmx

using cmx::Guard;

extern "C" {

cmx_status cmx_ceres2d_match(const cmx_ceres2d_options* options, const cmx_grid2d_limits* limits,
                             const uint16_t* cells, const double* target_translation_xy,
                             const cmx_pose2d* initial_pose_estimate, const float* point_cloud_xyz,
                             int32_t num_points, int32_t device, cmx_pose2d* pose_estimate,
                             cmx_ceres_summary* summary) {
  return Guard([&] {
    CMX_REQUIRE(limits && cells && target_translation_xy && initial_pose_estimate && pose_estimate,
                "null argument");
    cmx::RefineItem item{};
    item.host_cells = cells;
    item.limits = *limits;
    item.target[0] = target_translation_xy[0];
    item.target[1] = target_translation_xy[1];
    item.initial = *initial_pose_estimate;
    cmx::RefineBatch(options, &item, 1, point_cloud_xyz, num_points, device, pose_estimate, summary);
  });
}

cmx_status cmx_ceres2d_match_grid(const cmx_ceres2d_options* options, const cmx_grid2d* grid,
                                  const double* target_translation_xy,
                                  const cmx_pose2d* initial_pose_estimate,
                                  const float* point_cloud_xyz, int32_t num_points,
                                  cmx_pose2d* pose_estimate, cmx_ceres_summary* summary) {
  return Guard([&] {
    CMX_REQUIRE(grid && target_translation_xy && initial_pose_estimate && pose_estimate,
                "null argument");
    cmx::RefineItem item{};
    int device = 0;
    item.device_cells = cmx::Grid2DDeviceCells(grid, &item.limits, &device);
    item.target[0] = target_translation_xy[0];
    item.target[1] = target_translation_xy[1];
    item.initial = *initial_pose_estimate;
    cmx::RefineBatch(options, &item, 1, point_cloud_xyz, num_points, device, pose_estimate, summary);
  });
}

cmx_status cmx_fast2d_refine_batch(const cmx_ceres2d_options* options,
                                   const cmx_fast2d* const* matchers, int32_t num_matchers,
                                   const int32_t* found, const cmx_pose2d* pose_estimates_in,
                                   const float* point_cloud_xyz, int32_t num_points,
                                   cmx_pose2d* pose_estimates_out, cmx_ceres_summary* summaries) {
  return Guard([&] {
    CMX_REQUIRE(matchers && num_matchers >= 1 && pose_estimates_in && pose_estimates_out,
                "null argument");
    // Entries are grouped by the device their grid lives on (a node's batch may span the GPUs of
    // a cmx_comm); every group is one launch.
    std::map<int, std::vector<int>> by_device;
    for (int p = 0; p < num_matchers; ++p) {
      CMX_REQUIRE(matchers[p] && matchers[p]->impl, "null matcher handle");
      by_device[matchers[p]->impl->device()].push_back(p);
    }
    for (const auto& group : by_device) {
      const std::vector<int>& idx = group.second;
      const int m = static_cast<int>(idx.size());
      std::vector<cmx::RefineItem> items(m);
      for (int k = 0; k < m; ++k) {
        const int p = idx[k];
        const cmx::Fast2DMatcher& matcher = *matchers[p]->impl;
        cmx::RefineItem& it = items[k];
        it.device_cells = matcher.grid_cells();
        it.limits = matcher.limits();
        // constraint_builder_2d.cc:245-249: Match(pose_estimate.translation(), pose_estimate, ...)
        it.initial = pose_estimates_in[p];
        it.target[0] = pose_estimates_in[p].x;
        it.target[1] = pose_estimates_in[p].y;
        it.skip = found && !found[p] ? 1 : 0;
      }
      std::vector<cmx_pose2d> poses(m);
      std::vector<cmx_ceres_summary> sums(m);
      cmx::RefineBatch(options, items.data(), m, point_cloud_xyz, num_points, group.first,
                       poses.data(), summaries ? sums.data() : nullptr);
      for (int k = 0; k < m; ++k) {
        pose_estimates_out[idx[k]] = poses[k];
        if (summaries) summaries[idx[k]] = sums[k];
      }
    }
  });
}

}  // extern "C"
