// Stand-in for ceres/ceres.h -- a small dense non-linear least-squares solver with Ceres' public
// interface, so that the reference's cost functions AND its CeresScanMatcher2D / 3D::Match compile
// unmodified (oracle/Makefile `ref_ceres`).  TEST INFRASTRUCTURE ONLY; ours, written from Ceres'
// published algorithm -- ceres-solver itself is a third-party dependency absent from
// /root/reference (pinned at 58c5edae2f7c4d2533fe8a975c1f5f0b892dfd3e,
// bazel/repositories.bzl:134-144), so PARITY WITH CERES' ITERATES REMAINS UNPINNED; what building
// the reference on this pins is every line of the reference AROUND the solver (the residual
// functors through Jets, the problem set-up, parameterizations, weights, the returned pose).
//
//   Jet, AutoDiffCostFunction            dual-number automatic differentiation (jet.h)
//   LocalParameterization                QuaternionParameterization, AutoDiffLocalParameterization
//   LossFunction                         HuberLoss, applied per residual block through Ceres'
//                                        Corrector (rho'' <= 0: scale by sqrt(rho'))
//   Problem                              parameter blocks in order of first appearance
//   Solve                                TrustRegionMinimizer + LevenbergMarquardtStrategy +
//                                        DENSE_QR (Householder on [J; D]) with Solver::Options
//                                        defaults: radius 1e4 / max 1e16 / min 1e-32,
//                                        min_relative_decrease 1e-3, LM diagonal in [1e-6, 1e32],
//                                        function / gradient / parameter tolerance 1e-6 / 1e-10 /
//                                        1e-8, Jacobi scaling 1 / (1 + |column|) from iteration 0,
//                                        non-monotonic steps (window 5) through
//                                        TrustRegionStepEvaluator, the lowest-cost iterate returned.
#ifndef ORACLE_REF_SHIMS_CERES_H_
#define ORACLE_REF_SHIMS_CERES_H_
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "ceres/jet.h"
#include "glog/logging.h"   // the real ceres.h pulls glog (and <climits>) in
#include "ceres/rotation.h"

namespace ceres {

enum { DYNAMIC = -1 };
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE };
enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };

// ---------------------------------------------------------------- cost functions ---
class CostFunction {
 public:
  virtual ~CostFunction() {}
  // jacobians[i] (may be null) is num_residuals x parameter_block_sizes()[i], row-major.
  virtual bool Evaluate(double const* const* parameters, double* residuals,
                        double** jacobians) const = 0;
  const std::vector<int32_t>& parameter_block_sizes() const { return parameter_block_sizes_; }
  int num_residuals() const { return num_residuals_; }

 protected:
  std::vector<int32_t>* mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
  void set_num_residuals(int n) { num_residuals_ = n; }

 private:
  std::vector<int32_t> parameter_block_sizes_;
  int num_residuals_ = 0;
};

namespace internal {
template <int... Ns> struct Sum;
template <> struct Sum<> { enum { value = 0 }; };
template <int N, int... Ns> struct Sum<N, Ns...> { enum { value = N + Sum<Ns...>::value }; };

// Calls functor(p[0], ..., p[k-1], out) for one or two parameter blocks.
template <typename Functor, typename T>
bool Call(const Functor& f, T const* const* p, int num_blocks, T* out) {
  (void)num_blocks;
  return false;
}
template <int kBlocks> struct Caller;
template <> struct Caller<1> {
  template <typename Functor, typename T>
  static bool Run(const Functor& f, T const* const* p, T* out) { return f(p[0], out); }
};
template <> struct Caller<2> {
  template <typename Functor, typename T>
  static bool Run(const Functor& f, T const* const* p, T* out) { return f(p[0], p[1], out); }
};
}  // namespace internal

template <typename CostFunctor, int kNumResiduals, int... Ns>
class AutoDiffCostFunction : public CostFunction {
 public:
  explicit AutoDiffCostFunction(CostFunctor* functor) : functor_(functor) {
    static_assert(kNumResiduals != DYNAMIC, "use the two-argument constructor");
    set_num_residuals(kNumResiduals);
    *mutable_parameter_block_sizes() = std::vector<int32_t>{Ns...};
  }
  AutoDiffCostFunction(CostFunctor* functor, int num_residuals) : functor_(functor) {
    set_num_residuals(num_residuals);
    *mutable_parameter_block_sizes() = std::vector<int32_t>{Ns...};
  }
  bool Evaluate(double const* const* parameters, double* residuals,
                double** jacobians) const override {
    constexpr int kBlocks = sizeof...(Ns);
    constexpr int kParams = internal::Sum<Ns...>::value;
    if (jacobians == nullptr)
      return internal::Caller<kBlocks>::Run(*functor_, parameters, residuals);
    typedef Jet<double, kParams> JetT;
    const int sizes[kBlocks] = {Ns...};
    std::vector<JetT> x(kParams);
    const JetT* blocks[kBlocks];
    int offset = 0;
    for (int b = 0; b < kBlocks; ++b) {
      blocks[b] = x.data() + offset;
      for (int j = 0; j < sizes[b]; ++j) x[offset + j] = JetT(parameters[b][j], offset + j);
      offset += sizes[b];
    }
    std::vector<JetT> out(num_residuals());
    if (!internal::Caller<kBlocks>::Run(*functor_, blocks, out.data())) return false;
    offset = 0;
    for (int r = 0; r < num_residuals(); ++r) residuals[r] = out[r].a;
    for (int b = 0; b < kBlocks; ++b) {
      if (jacobians[b] != nullptr)
        for (int r = 0; r < num_residuals(); ++r)
          for (int j = 0; j < sizes[b]; ++j) jacobians[b][r * sizes[b] + j] = out[r].v[offset + j];
      offset += sizes[b];
    }
    return true;
  }

 private:
  std::unique_ptr<CostFunctor> functor_;
};

// ------------------------------------------------------------- parameterizations ---
class LocalParameterization {
 public:
  virtual ~LocalParameterization() {}
  virtual bool Plus(const double* x, const double* delta, double* x_plus_delta) const = 0;
  // GlobalSize() x LocalSize(), row-major.
  virtual bool ComputeJacobian(const double* x, double* jacobian) const = 0;
  virtual int GlobalSize() const = 0;
  virtual int LocalSize() const = 0;
};

// local_parameterization.cc: x_plus_delta = [cos|d|, sin|d| d/|d|] * x; the Jacobian at delta = 0.
class QuaternionParameterization : public LocalParameterization {
 public:
  bool Plus(const double* x, const double* delta, double* x_plus_delta) const override {
    const double norm_delta =
        std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
    if (norm_delta > 0.0) {
      const double sin_delta_by_delta = std::sin(norm_delta) / norm_delta;
      double q_delta[4];
      q_delta[0] = std::cos(norm_delta);
      q_delta[1] = sin_delta_by_delta * delta[0];
      q_delta[2] = sin_delta_by_delta * delta[1];
      q_delta[3] = sin_delta_by_delta * delta[2];
      QuaternionProduct(q_delta, x, x_plus_delta);
    } else {
      for (int i = 0; i < 4; ++i) x_plus_delta[i] = x[i];
    }
    return true;
  }
  bool ComputeJacobian(const double* x, double* jacobian) const override {
    jacobian[0] = -x[1]; jacobian[1] = -x[2]; jacobian[2] = -x[3];
    jacobian[3] = x[0];  jacobian[4] = x[3];  jacobian[5] = -x[2];
    jacobian[6] = -x[3]; jacobian[7] = x[0];  jacobian[8] = x[1];
    jacobian[9] = x[2];  jacobian[10] = -x[1]; jacobian[11] = x[0];
    return true;
  }
  int GlobalSize() const override { return 4; }
  int LocalSize() const override { return 3; }
};

template <typename Functor, int kGlobalSize, int kLocalSize>
class AutoDiffLocalParameterization : public LocalParameterization {
 public:
  AutoDiffLocalParameterization() : functor_(new Functor()) {}
  explicit AutoDiffLocalParameterization(Functor* functor) : functor_(functor) {}
  bool Plus(const double* x, const double* delta, double* x_plus_delta) const override {
    return (*functor_)(x, delta, x_plus_delta);
  }
  bool ComputeJacobian(const double* x, double* jacobian) const override {
    typedef Jet<double, kLocalSize> JetT;   // derivative w.r.t. delta at delta = 0, x constant
    JetT xj[kGlobalSize], dj[kLocalSize], out[kGlobalSize];
    for (int i = 0; i < kGlobalSize; ++i) xj[i] = JetT(x[i]);
    for (int i = 0; i < kLocalSize; ++i) dj[i] = JetT(0.0, i);
    if (!(*functor_)(xj, dj, out)) return false;
    for (int i = 0; i < kGlobalSize; ++i)
      for (int j = 0; j < kLocalSize; ++j) jacobian[i * kLocalSize + j] = out[i].v[j];
    return true;
  }
  int GlobalSize() const override { return kGlobalSize; }
  int LocalSize() const override { return kLocalSize; }

 private:
  std::unique_ptr<Functor> functor_;
};

// ------------------------------------------------------------------------ losses ---
class LossFunction {
 public:
  virtual ~LossFunction() {}
  virtual void Evaluate(double sq_norm, double out[3]) const = 0;   // rho, rho', rho''
};
class HuberLoss : public LossFunction {   // loss_function.cc
 public:
  explicit HuberLoss(double a) : a_(a), b_(a * a) {}
  void Evaluate(double s, double rho[3]) const override {
    if (s > b_) {
      const double r = std::sqrt(s);
      rho[0] = 2.0 * a_ * r - b_;
      rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r);
      rho[2] = -rho[1] / (2.0 * s);
    } else {
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
  }

 private:
  const double a_, b_;
};

// ----------------------------------------------------------------------- problem ---
class Problem {
 public:
  struct Options {};
  Problem() {}
  explicit Problem(const Options&) {}
  Problem(const Problem&) = delete;
  Problem& operator=(const Problem&) = delete;

  void AddParameterBlock(double* values, int size) { Block(values, size); }
  void AddParameterBlock(double* values, int size, LocalParameterization* parameterization) {
    ParameterBlock& b = Block(values, size);
    if (parameterization != nullptr) b.parameterization.reset(parameterization);
  }
  void SetParameterization(double* values, LocalParameterization* parameterization) {
    blocks_[index_.at(values)].parameterization.reset(parameterization);
  }
  void* AddResidualBlock(CostFunction* cost, LossFunction* loss, double* x0) {
    return AddResidualBlock(cost, loss, std::vector<double*>{x0});
  }
  void* AddResidualBlock(CostFunction* cost, LossFunction* loss, double* x0, double* x1) {
    return AddResidualBlock(cost, loss, std::vector<double*>{x0, x1});
  }
  void* AddResidualBlock(CostFunction* cost, LossFunction* loss,
                         const std::vector<double*>& parameter_blocks) {
    ResidualBlock r;
    r.cost.reset(cost);
    r.loss.reset(loss);
    for (size_t i = 0; i < parameter_blocks.size(); ++i) {
      Block(parameter_blocks[i], cost->parameter_block_sizes()[i]);
      r.blocks.push_back(index_.at(parameter_blocks[i]));
    }
    residual_blocks_.push_back(std::move(r));
    return &residual_blocks_.back();
  }
  int NumResiduals() const {
    int n = 0;
    for (const ResidualBlock& r : residual_blocks_) n += r.cost->num_residuals();
    return n;
  }

  // -- evaluation (used by Solve and by the test wrapper) --
  int NumParameters() const { int n = 0; for (const auto& b : blocks_) n += b.size; return n; }
  int NumEffectiveParameters() const {
    int n = 0;
    for (const auto& b : blocks_) n += b.parameterization ? b.parameterization->LocalSize() : b.size;
    return n;
  }
  void GetState(double* x) const {
    int o = 0;
    for (const auto& b : blocks_) { for (int i = 0; i < b.size; ++i) x[o + i] = b.values[i]; o += b.size; }
  }
  void SetState(const double* x) {
    int o = 0;
    for (auto& b : blocks_) { for (int i = 0; i < b.size; ++i) b.values[i] = x[o + i]; o += b.size; }
  }
  // x_plus_delta = Plus(x, delta), block by block (delta in the tangent space).
  void Plus(const double* x, const double* delta, double* x_plus_delta) const {
    int o = 0, l = 0;
    for (const auto& b : blocks_) {
      if (b.parameterization) {
        b.parameterization->Plus(x + o, delta + l, x_plus_delta + o);
        l += b.parameterization->LocalSize();
      } else {
        for (int i = 0; i < b.size; ++i) x_plus_delta[o + i] = x[o + i] + delta[l + i];
        l += b.size;
      }
      o += b.size;
    }
  }
  // cost = sum over residual blocks of 1/2 rho(|r|^2); residuals / jacobian (num_residuals x
  // NumEffectiveParameters, row-major, tangent space) corrected for the loss (Corrector) when
  // requested.  `x` = ambient state.
  bool Evaluate(const double* x, double* cost, std::vector<double>* residuals,
                std::vector<double>* jacobian) const {
    const int m = NumResiduals(), n = NumEffectiveParameters();
    if (residuals) residuals->assign(m, 0.0);
    if (jacobian) jacobian->assign(static_cast<size_t>(m) * n, 0.0);
    std::vector<int> ambient_offset(blocks_.size()), local_offset(blocks_.size());
    {
      int o = 0, l = 0;
      for (size_t i = 0; i < blocks_.size(); ++i) {
        ambient_offset[i] = o; local_offset[i] = l;
        o += blocks_[i].size;
        l += blocks_[i].parameterization ? blocks_[i].parameterization->LocalSize() : blocks_[i].size;
      }
    }
    *cost = 0.0;
    int row = 0;
    for (const ResidualBlock& rb : residual_blocks_) {
      const int nr = rb.cost->num_residuals();
      std::vector<const double*> params;
      std::vector<std::vector<double>> jac(rb.blocks.size());
      std::vector<double*> jac_ptr(rb.blocks.size(), nullptr);
      for (size_t i = 0; i < rb.blocks.size(); ++i) {
        params.push_back(x + ambient_offset[rb.blocks[i]]);
        if (jacobian) {
          jac[i].assign(static_cast<size_t>(nr) * blocks_[rb.blocks[i]].size, 0.0);
          jac_ptr[i] = jac[i].data();
        }
      }
      std::vector<double> r(nr);
      if (!rb.cost->Evaluate(params.data(), r.data(), jacobian ? jac_ptr.data() : nullptr))
        return false;
      double sq_norm = 0.0;
      for (int k = 0; k < nr; ++k) sq_norm += r[k] * r[k];
      double residual_scaling = 1.0, sqrt_rho1 = 1.0, alpha_sq_norm = 0.0;
      if (rb.loss) {
        double rho[3];
        rb.loss->Evaluate(sq_norm, rho);
        *cost += 0.5 * rho[0];
        // corrector.cc
        sqrt_rho1 = std::sqrt(rho[1]);
        if (sq_norm == 0.0 || rho[2] <= 0.0) {
          residual_scaling = sqrt_rho1;
          alpha_sq_norm = 0.0;
        } else {
          const double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
          const double alpha = 1.0 - std::sqrt(D);
          residual_scaling = sqrt_rho1 / (1 - alpha);
          alpha_sq_norm = alpha / sq_norm;
        }
      } else {
        *cost += 0.5 * sq_norm;
      }
      if (jacobian) {
        for (size_t i = 0; i < rb.blocks.size(); ++i) {
          const ParameterBlock& pb = blocks_[rb.blocks[i]];
          const int gs = pb.size;
          if (rb.loss) {   // Corrector::CorrectJacobian, on the ambient block, before the residuals
            if (alpha_sq_norm == 0.0) {
              for (double& v : jac[i]) v *= sqrt_rho1;
            } else {
              for (int c = 0; c < gs; ++c) {
                double r_transpose_j = 0.0;
                for (int k = 0; k < nr; ++k) r_transpose_j += jac[i][k * gs + c] * r[k];
                for (int k = 0; k < nr; ++k)
                  jac[i][k * gs + c] =
                      sqrt_rho1 * (jac[i][k * gs + c] - alpha_sq_norm * r[k] * r_transpose_j);
              }
            }
          }
          const int lo = local_offset[rb.blocks[i]];
          if (pb.parameterization) {
            const int ls = pb.parameterization->LocalSize();
            std::vector<double> pj(static_cast<size_t>(gs) * ls);
            pb.parameterization->ComputeJacobian(params[i], pj.data());
            for (int k = 0; k < nr; ++k)
              for (int c = 0; c < ls; ++c) {
                double s = 0.0;
                for (int g = 0; g < gs; ++g) s += jac[i][k * gs + g] * pj[g * ls + c];
                (*jacobian)[static_cast<size_t>(row + k) * n + lo + c] += s;
              }
          } else {
            for (int k = 0; k < nr; ++k)
              for (int c = 0; c < gs; ++c)
                (*jacobian)[static_cast<size_t>(row + k) * n + lo + c] += jac[i][k * gs + c];
          }
        }
      }
      if (residuals)
        for (int k = 0; k < nr; ++k) (*residuals)[row + k] = r[k] * residual_scaling;
      row += nr;
    }
    return true;
  }

 private:
  struct ParameterBlock {
    double* values = nullptr;
    int size = 0;
    std::unique_ptr<LocalParameterization> parameterization;
  };
  struct ResidualBlock {
    std::unique_ptr<CostFunction> cost;
    std::unique_ptr<LossFunction> loss;
    std::vector<int> blocks;
  };
  ParameterBlock& Block(double* values, int size) {
    auto it = index_.find(values);
    if (it == index_.end()) {
      index_[values] = static_cast<int>(blocks_.size());
      blocks_.emplace_back();
      blocks_.back().values = values;
      blocks_.back().size = size;
      return blocks_.back();
    }
    return blocks_[it->second];
  }
  std::vector<ParameterBlock> blocks_;
  std::map<double*, int> index_;
  std::vector<ResidualBlock> residual_blocks_;
};

// ------------------------------------------------------------------------ solver ---
struct IterationSummary {
  int iteration = 0;
  bool step_is_valid = false, step_is_successful = false;
  double cost = 0., cost_change = 0., gradient_max_norm = 0., step_norm = 0.,
         relative_decrease = 0., trust_region_radius = 0.;
};

class Solver {
 public:
  struct Options {
    LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY;
    bool use_nonmonotonic_steps = false;
    int max_consecutive_nonmonotonic_steps = 5;
    int max_num_iterations = 50;
    int num_threads = 1;
    double initial_trust_region_radius = 1e4, max_trust_region_radius = 1e16,
           min_trust_region_radius = 1e-32, min_relative_decrease = 1e-3,
           min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
    int max_num_consecutive_invalid_steps = 5;
    double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    bool jacobi_scaling = true;
  };
  struct Summary {
    double initial_cost = -1., final_cost = -1.;
    int num_successful_steps = -1, num_unsuccessful_steps = -1;
    TerminationType termination_type = FAILURE;
    std::string message;
    std::vector<IterationSummary> iterations;
    std::string BriefReport() const { return message; }
    std::string FullReport() const { return message; }
    bool IsSolutionUsable() const {
      return termination_type == CONVERGENCE || termination_type == NO_CONVERGENCE;
    }
  };
};

namespace internal {
// min |A y - b|^2 for the (rows x cols) row-major A by Householder QR, no pivoting (what
// DenseQRSolver does with Eigen::HouseholderQR on [J; D], [r; 0]).  False on a zero pivot.
inline bool HouseholderLeastSquares(std::vector<double> A, std::vector<double> b, int rows,
                                    int cols, double* y) {
  for (int k = 0; k < cols; ++k) {
    double tail = 0.0;
    for (int i = k + 1; i < rows; ++i) tail += A[i * cols + k] * A[i * cols + k];
    const double c0 = A[k * cols + k];
    if (tail == 0.0) {
      if (c0 == 0.0) return false;
      continue;   // already upper triangular in this column
    }
    double beta = std::sqrt(c0 * c0 + tail);
    if (c0 >= 0.0) beta = -beta;
    // v = (1, essential), essential = tail / (c0 - beta); tau = (beta - c0) / beta
    const double denom = c0 - beta, tau = (beta - c0) / beta;
    for (int i = k + 1; i < rows; ++i) A[i * cols + k] /= denom;
    A[k * cols + k] = beta;
    for (int j = k + 1; j <= cols; ++j) {   // column `cols` = the right-hand side
      double dot = j < cols ? A[k * cols + j] : b[k];
      for (int i = k + 1; i < rows; ++i) dot += A[i * cols + k] * (j < cols ? A[i * cols + j] : b[i]);
      dot *= tau;
      if (j < cols) A[k * cols + j] -= dot; else b[k] -= dot;
      for (int i = k + 1; i < rows; ++i) {
        if (j < cols) A[i * cols + j] -= dot * A[i * cols + k];
        else b[i] -= dot * A[i * cols + k];
      }
    }
  }
  for (int k = cols - 1; k >= 0; --k) {
    double s = b[k];
    for (int j = k + 1; j < cols; ++j) s -= A[k * cols + j] * y[j];
    if (A[k * cols + k] == 0.0) return false;
    y[k] = s / A[k * cols + k];
  }
  for (int k = 0; k < cols; ++k) if (!std::isfinite(y[k])) return false;
  return true;
}
}  // namespace internal

// trust_region_minimizer.cc / levenberg_marquardt_strategy.cc / trust_region_step_evaluator.cc.
inline void Solve(const Solver::Options& options, Problem* problem, Solver::Summary* summary) {
  Solver::Summary local;
  Solver::Summary& sum = summary ? *summary : local;
  sum = Solver::Summary();
  const int n_ambient = problem->NumParameters(), n = problem->NumEffectiveParameters(),
            m = problem->NumResiduals();
  std::vector<double> x(n_ambient), candidate_x(n_ambient), best_x(n_ambient);
  problem->GetState(x.data());
  best_x = x;
  std::vector<double> r, J, r_candidate;
  double x_cost = 0.;
  auto norm = [](const std::vector<double>& v) {
    double s = 0.; for (double e : v) s += e * e; return std::sqrt(s);
  };
  std::vector<double> scale(n, 1.0), gradient(n), step(n), delta(n);
  double gradient_max_norm = 0.;
  bool scale_known = false;
  auto evaluate_gradient_and_jacobian = [&]() -> bool {
    if (!problem->Evaluate(x.data(), &x_cost, &r, &J)) return false;
    for (int c = 0; c < n; ++c) {      // gradient = J^T r of the UNSCALED Jacobian
      double g = 0.;
      for (int k = 0; k < m; ++k) g += J[static_cast<size_t>(k) * n + c] * r[k];
      gradient[c] = g;
    }
    if (options.jacobi_scaling) {
      if (!scale_known) {
        for (int c = 0; c < n; ++c) {
          double s = 0.;
          for (int k = 0; k < m; ++k) s += J[static_cast<size_t>(k) * n + c] * J[static_cast<size_t>(k) * n + c];
          scale[c] = 1.0 / (1.0 + std::sqrt(s));
        }
        scale_known = true;
      }
      for (int k = 0; k < m; ++k)
        for (int c = 0; c < n; ++c) J[static_cast<size_t>(k) * n + c] *= scale[c];
    }
    // |x - Plus(x, -gradient)|_inf in the ambient space.
    std::vector<double> neg(n), projected(n_ambient);
    for (int c = 0; c < n; ++c) neg[c] = -gradient[c];
    problem->Plus(x.data(), neg.data(), projected.data());
    gradient_max_norm = 0.;
    for (int i = 0; i < n_ambient; ++i)
      gradient_max_norm = std::max(gradient_max_norm, std::fabs(x[i] - projected[i]));
    return true;
  };

  // ---- iteration zero ----
  double x_norm = norm(x);
  if (!evaluate_gradient_and_jacobian()) {
    sum.termination_type = FAILURE; sum.message = "initial evaluation failed";
    return;
  }
  sum.initial_cost = x_cost;
  double minimum_cost_of_minimizer = std::numeric_limits<double>::max();
  sum.num_successful_steps = 0;
  sum.num_unsuccessful_steps = 0;
  IterationSummary it;
  it.iteration = 0; it.step_is_valid = true; it.step_is_successful = true; it.cost = x_cost;
  it.gradient_max_norm = gradient_max_norm;

  // LevenbergMarquardtStrategy
  double radius = options.initial_trust_region_radius, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  std::vector<double> diagonal(n, 0.0);
  // TrustRegionStepEvaluator
  const int max_nonmonotonic =
      options.use_nonmonotonic_steps ? options.max_consecutive_nonmonotonic_steps : 0;
  double ev_minimum_cost = x_cost, ev_current_cost = x_cost, ev_reference_cost = x_cost,
         ev_candidate_cost = x_cost;
  double acc_reference = 0., acc_candidate = 0.;
  int num_consecutive_nonmonotonic_steps = 0, num_consecutive_invalid_steps = 0;

  for (;;) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (it.step_is_successful) {
      ++sum.num_successful_steps;
      if (x_cost < minimum_cost_of_minimizer) {
        minimum_cost_of_minimizer = x_cost;
        best_x = x;
      }
    } else {
      ++sum.num_unsuccessful_steps;
    }
    it.trust_region_radius = radius;
    sum.iterations.push_back(it);
    if (it.iteration >= options.max_num_iterations) {
      sum.termination_type = NO_CONVERGENCE; sum.message = "Maximum number of iterations reached.";
      break;
    }
    if (it.step_is_successful && it.gradient_max_norm <= options.gradient_tolerance) {
      sum.termination_type = CONVERGENCE; sum.message = "Gradient tolerance reached.";
      break;
    }
    if (radius < options.min_trust_region_radius) {
      sum.termination_type = CONVERGENCE; sum.message = "Minimum trust region radius reached.";
      break;
    }
    const double previous_gradient_max_norm = it.gradient_max_norm;
    const int iteration = it.iteration + 1;
    it = IterationSummary();
    it.iteration = iteration;

    // ---- LevenbergMarquardtStrategy::ComputeStep ----
    if (!reuse_diagonal) {
      for (int c = 0; c < n; ++c) {
        double s = 0.;
        for (int k = 0; k < m; ++k) s += J[static_cast<size_t>(k) * n + c] * J[static_cast<size_t>(k) * n + c];
        diagonal[c] = std::min(std::max(s, options.min_lm_diagonal), options.max_lm_diagonal);
      }
    }
    std::vector<double> A(static_cast<size_t>(m + n) * n, 0.0), b(m + n, 0.0);
    std::copy(J.begin(), J.end(), A.begin());
    for (int c = 0; c < n; ++c) A[static_cast<size_t>(m + c) * n + c] = std::sqrt(diagonal[c] / radius);
    std::copy(r.begin(), r.end(), b.begin());
    const bool solved = internal::HouseholderLeastSquares(A, b, m + n, n, step.data());
    for (int c = 0; c < n; ++c) step[c] = -step[c];
    reuse_diagonal = true;
    double model_cost_change = 0.;
    if (solved) {   // -(J step)^T (r + J step / 2)
      for (int k = 0; k < m; ++k) {
        double js = 0.;
        for (int c = 0; c < n; ++c) js += J[static_cast<size_t>(k) * n + c] * step[c];
        model_cost_change -= js * (r[k] + js / 2.0);
      }
    }
    it.step_is_valid = solved && model_cost_change > 0.0;
    if (!it.step_is_valid) {
      // HandleInvalidStep
      if (++num_consecutive_invalid_steps >= options.max_num_consecutive_invalid_steps) {
        sum.termination_type = FAILURE; sum.message = "Too many consecutive invalid steps.";
        break;
      }
      radius *= 0.5;            // LevenbergMarquardtStrategy::StepIsInvalid
      reuse_diagonal = false;
      it.cost = x_cost; it.gradient_max_norm = previous_gradient_max_norm;
      continue;
    }
    num_consecutive_invalid_steps = 0;
    for (int c = 0; c < n; ++c) delta[c] = step[c] * scale[c];
    problem->Plus(x.data(), delta.data(), candidate_x.data());
    double candidate_cost = 0.;
    if (!problem->Evaluate(candidate_x.data(), &candidate_cost, nullptr, nullptr))
      candidate_cost = std::numeric_limits<double>::max();

    // ParameterToleranceReached / FunctionToleranceReached
    double sn = 0.;
    for (int i = 0; i < n_ambient; ++i) sn += (x[i] - candidate_x[i]) * (x[i] - candidate_x[i]);
    it.step_norm = std::sqrt(sn);
    if (it.step_norm <= options.parameter_tolerance * (x_norm + options.parameter_tolerance)) {
      sum.termination_type = CONVERGENCE; sum.message = "Parameter tolerance reached.";
      break;
    }
    it.cost_change = x_cost - candidate_cost;
    if (std::fabs(it.cost_change) <= options.function_tolerance * x_cost) {
      sum.termination_type = CONVERGENCE; sum.message = "Function tolerance reached.";
      break;
    }
    // TrustRegionStepEvaluator::StepQuality
    if (candidate_cost >= std::numeric_limits<double>::max()) {
      it.relative_decrease = std::numeric_limits<double>::lowest();
    } else {
      const double relative_decrease = (ev_current_cost - candidate_cost) / model_cost_change;
      const double historical_relative_decrease =
          (ev_reference_cost - candidate_cost) / (acc_reference + model_cost_change);
      it.relative_decrease = std::max(relative_decrease, historical_relative_decrease);
    }
    if (it.relative_decrease > options.min_relative_decrease) {
      // HandleSuccessfulStep
      x = candidate_x;
      x_norm = norm(x);
      if (!evaluate_gradient_and_jacobian()) {
        sum.termination_type = FAILURE; sum.message = "evaluation failed";
        break;
      }
      it.step_is_successful = true;
      it.cost = x_cost; it.gradient_max_norm = gradient_max_norm;
      // LevenbergMarquardtStrategy::StepAccepted
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * it.relative_decrease - 1.0, 3));
      radius = std::min(options.max_trust_region_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = false;
      // TrustRegionStepEvaluator::StepAccepted
      ev_current_cost = candidate_cost;
      acc_candidate += model_cost_change;
      acc_reference += model_cost_change;
      if (ev_current_cost < ev_minimum_cost) {
        ev_minimum_cost = ev_current_cost;
        num_consecutive_nonmonotonic_steps = 0;
        ev_candidate_cost = ev_current_cost;
        acc_candidate = 0.0;
      } else {
        ++num_consecutive_nonmonotonic_steps;
        if (ev_current_cost > ev_candidate_cost) {
          ev_candidate_cost = ev_current_cost;
          acc_candidate = 0.0;
        }
      }
      if (num_consecutive_nonmonotonic_steps == max_nonmonotonic) {
        ev_reference_cost = ev_candidate_cost;
        acc_reference = acc_candidate;
      }
    } else {
      it.step_is_successful = false;
      it.cost = candidate_cost; it.gradient_max_norm = previous_gradient_max_norm;
      radius = radius / decrease_factor;   // LevenbergMarquardtStrategy::StepRejected
      decrease_factor *= 2.0;
      reuse_diagonal = true;
    }
  }
  // the parameters handed back are those of the lowest cost seen
  problem->SetState(best_x.data());
  sum.final_cost = sum.initial_cost;
  for (const IterationSummary& s : sum.iterations)
    if (s.step_is_successful) sum.final_cost = std::min(sum.final_cost, s.cost);
}

}  // namespace ceres
#endif  // ORACLE_REF_SHIMS_CERES_H_
