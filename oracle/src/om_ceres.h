// ORACLE (test infrastructure, NOT product code).
//
// Restatement of the part of Ceres Solver 1.13.0 that CeresScanMatcher3D
// drives (call sites: .../scan_matching/ceres_scan_matcher_3d.cc:63-123,
// common/ceres_solver_options.cc:35-42, optimization/ceres_pose.cc:30-44).
// Ceres is a THIRD-PARTY dependency that is not in /root/reference and not
// installed here (pinned: scripts/install_ceres.sh VERSION="1.13.0",
// bazel/repositories.bzl:128).  What follows restates its published algorithm
// for this configuration only:
//   * one 3-dof block (translation, no parameterisation) + one 4-dof block with
//     QuaternionParameterization (or the yaw-only autodiff parameterisation of
//     mapping/internal/3d/rotation_parameterization.h:27-39),
//   * residual blocks evaluated by forward-mode autodiff (Jet<7>),
//   * TRUST_REGION / LEVENBERG_MARQUARDT / DENSE_QR with every other option
//     at its 1.13 default (jacobi_scaling, initial radius 1e4, tolerances
//     1e-6 / 1e-10 / 1e-8, min_relative_decrease 1e-3, lm diagonal clamp
//     [1e-6, 1e32], max_consecutive_nonmonotonic_steps 5, max 5 invalid steps),
//     following internal/ceres/trust_region_minimizer.cc,
//     levenberg_marquardt_strategy.cc, trust_region_step_evaluator.cc,
//     dense_qr_solver.cc and solver.cc(SetSummaryFinalCost) of that release.
// Parity of this restatement is pinned only through the reference's own tests
// (ceres_scan_matcher_3d_test.cc:34-116, rotation_delta_cost_functor_3d_test.cc),
// see tests/test_oracle_kat.py.
#ifndef ORACLE_OM_CERES_H_
#define ORACLE_OM_CERES_H_

#include <algorithm>
#include <cmath>
#include <functional>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "om_jet.h"

namespace oracle {
namespace ceres_like {

constexpr int kAmbient = 7;  // t[3] + q[4]

// ---- local parameterisations of the rotation block -------------------------
struct RotationParameterization {
  virtual ~RotationParameterization() {}
  virtual int LocalSize() const = 0;
  virtual void Plus(const double* x, const double* delta, double* out) const = 0;
  // 4 x LocalSize, row-major
  virtual void ComputeJacobian(const double* x, double* jac) const = 0;
};

inline void QuaternionProductD(const double z[4], const double w[4], double zw[4]) {
  // ceres/rotation.h QuaternionProduct
  zw[0] = z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3];
  zw[1] = z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2];
  zw[2] = z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1];
  zw[3] = z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0];
}

// ceres/local_parameterization.cc QuaternionParameterization
struct QuaternionParameterization : RotationParameterization {
  int LocalSize() const override { return 3; }
  void Plus(const double* x, const double* delta, double* out) const override {
    const double norm_delta = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] +
                                        delta[2] * delta[2]);
    if (norm_delta > 0.0) {
      const double sin_delta_by_delta = std::sin(norm_delta) / norm_delta;
      double q_delta[4];
      q_delta[0] = std::cos(norm_delta);
      q_delta[1] = sin_delta_by_delta * delta[0];
      q_delta[2] = sin_delta_by_delta * delta[1];
      q_delta[3] = sin_delta_by_delta * delta[2];
      QuaternionProductD(q_delta, x, out);
    } else {
      for (int i = 0; i < 4; ++i) out[i] = x[i];
    }
  }
  void ComputeJacobian(const double* x, double* j) const override {
    j[0] = -x[1]; j[1] = -x[2]; j[2] = -x[3];
    j[3] = x[0];  j[4] = x[3];  j[5] = -x[2];
    j[6] = -x[3]; j[7] = x[0];  j[8] = x[1];
    j[9] = x[2];  j[10] = -x[1]; j[11] = x[0];
  }
};

// rotation_parameterization.h:27-39 wrapped in
// ceres::AutoDiffLocalParameterization<YawOnlyQuaternionPlus, 4, 1>:
// Plus evaluates the functor in double; the Jacobian is d(Plus)/d(delta) at
// delta = 0, i.e. (0,0,0,1) (x) x.
struct YawOnlyQuaternionParameterization : RotationParameterization {
  int LocalSize() const override { return 1; }
  void Plus(const double* x, const double* delta, double* out) const override {
    double d = delta[0];
    if (d > 0.5) d = 0.5;
    if (d < -0.5) d = -0.5;
    const double q_delta[4] = {std::sqrt(1. - d * d), 0., 0., d};
    QuaternionProductD(q_delta, x, out);
  }
  void ComputeJacobian(const double* x, double* j) const override {
    // d/dd [sqrt(1-d^2),0,0,d] at 0 = (0,0,0,1); product with x, via Jets the
    // zero terms drop out exactly.
    j[0] = -x[3];
    j[1] = -x[2];
    j[2] = x[1];
    j[3] = x[0];
  }
};

// ---- residual blocks ---------------------------------------------------------
// Evaluate(t, q, residuals, jac_t (n x 3) or null, jac_q (n x 4) or null)
struct ResidualBlock {
  int num_residuals;
  bool uses_translation;
  bool uses_rotation;
  std::function<bool(const double* t, const double* q, double* residuals,
                     double* jac_t, double* jac_q)>
      evaluate;
};

// AutoDiffCostFunction<F, DYNAMIC, 3, 4>: both blocks, Jet<7>, parameter k of
// block 0 seeded at 0..2, block 1 at 3..6.
template <typename Functor>
ResidualBlock MakeAutoDiffBlock34(std::shared_ptr<Functor> f, int num_residuals) {
  ResidualBlock b;
  b.num_residuals = num_residuals;
  b.uses_translation = true;
  b.uses_rotation = true;
  b.evaluate = [f, num_residuals](const double* t, const double* q, double* r,
                                  double* jt, double* jq) {
    if (jt == nullptr && jq == nullptr) return (*f)(t, q, r);
    using J = Jet<7>;
    J jt_in[3], jq_in[4];
    for (int i = 0; i < 3; ++i) jt_in[i] = J(t[i], i);
    for (int i = 0; i < 4; ++i) jq_in[i] = J(q[i], 3 + i);
    std::vector<J> out(num_residuals);
    if (!(*f)(jt_in, jq_in, out.data())) return false;
    for (int i = 0; i < num_residuals; ++i) {
      r[i] = out[i].a;
      if (jt != nullptr)
        for (int k = 0; k < 3; ++k) jt[i * 3 + k] = out[i].v[k];
      if (jq != nullptr)
        for (int k = 0; k < 4; ++k) jq[i * 4 + k] = out[i].v[3 + k];
    }
    return true;
  };
  return b;
}
// AutoDiffCostFunction<F, 3, 3> on the translation block.
template <typename Functor>
ResidualBlock MakeAutoDiffBlock3(std::shared_ptr<Functor> f) {
  ResidualBlock b;
  b.num_residuals = 3;
  b.uses_translation = true;
  b.uses_rotation = false;
  b.evaluate = [f](const double* t, const double*, double* r, double* jt, double*) {
    if (jt == nullptr) return (*f)(t, r);
    using J = Jet<3>;
    J in[3], out[3];
    for (int i = 0; i < 3; ++i) in[i] = J(t[i], i);
    if (!(*f)(in, out)) return false;
    for (int i = 0; i < 3; ++i) {
      r[i] = out[i].a;
      for (int k = 0; k < 3; ++k) jt[i * 3 + k] = out[i].v[k];
    }
    return true;
  };
  return b;
}
// AutoDiffCostFunction<F, 3, 4> on the rotation block.
template <typename Functor>
ResidualBlock MakeAutoDiffBlock4(std::shared_ptr<Functor> f) {
  ResidualBlock b;
  b.num_residuals = 3;
  b.uses_translation = false;
  b.uses_rotation = true;
  b.evaluate = [f](const double*, const double* q, double* r, double*, double* jq) {
    if (jq == nullptr) return (*f)(q, r);
    using J = Jet<4>;
    J in[4], out[3];
    for (int i = 0; i < 4; ++i) in[i] = J(q[i], i);
    if (!(*f)(in, out)) return false;
    for (int i = 0; i < 3; ++i) {
      r[i] = out[i].a;
      for (int k = 0; k < 4; ++k) jq[i * 4 + k] = out[i].v[k];
    }
    return true;
  };
  return b;
}

// ---- problem + evaluator ------------------------------------------------------
struct Problem {
  double t[3];
  double q[4];
  std::unique_ptr<RotationParameterization> rotation_parameterization;
  std::vector<ResidualBlock> blocks;

  int NumResiduals() const {
    int n = 0;
    for (const ResidualBlock& b : blocks) n += b.num_residuals;
    return n;
  }
  int NumEffectiveParameters() const {
    return 3 + rotation_parameterization->LocalSize();
  }
};

struct Summary {
  double initial_cost = 0;
  double final_cost = 0;
  int num_successful_steps = 0;
  int num_unsuccessful_steps = 0;
  int num_iterations = 0;            // iterations.size() incl. iteration 0
  int num_residual_evaluations = 0;  // cost evaluations (with or w/o Jacobian)
  int num_jacobian_evaluations = 0;
  int termination_type = 0;          // 0 CONVERGENCE, 1 NO_CONVERGENCE, 2 FAILURE
  std::string message;
  std::vector<double> iteration_costs;
};

struct Options {
  bool use_nonmonotonic_steps = false;
  int max_num_iterations = 50;
  // 1.13 defaults the path never overrides:
  int max_consecutive_nonmonotonic_steps = 5;
  double initial_trust_region_radius = 1e4;
  double max_trust_region_radius = 1e16;
  double min_trust_region_radius = 1e-32;
  double min_relative_decrease = 1e-3;
  double min_lm_diagonal = 1e-6;
  double max_lm_diagonal = 1e32;
  int max_num_consecutive_invalid_steps = 5;
  double function_tolerance = 1e-6;
  double gradient_tolerance = 1e-10;
  double parameter_tolerance = 1e-8;
  bool jacobi_scaling = true;
};

// Dense column-major-free helper: row-major m x n matrix.
struct DenseMatrix {
  int rows = 0, cols = 0;
  std::vector<double> a;
  void Resize(int r, int c) {
    rows = r;
    cols = c;
    a.assign(static_cast<size_t>(r) * c, 0.0);
  }
  double& operator()(int r, int c) { return a[static_cast<size_t>(r) * cols + c]; }
  double operator()(int r, int c) const { return a[static_cast<size_t>(r) * cols + c]; }
};

class Evaluator {
 public:
  explicit Evaluator(Problem* p) : p_(p) {}
  int NumParameters() const { return kAmbient; }
  int NumEffectiveParameters() const { return p_->NumEffectiveParameters(); }
  int NumResiduals() const { return p_->NumResiduals(); }

  void Plus(const double* x, const double* delta, double* out) const {
    for (int i = 0; i < 3; ++i) out[i] = x[i] + delta[i];
    p_->rotation_parameterization->Plus(x + 3, delta + 3, out + 3);
  }

  // cost = 1/2 sum r^2 accumulated block by block; gradient = J^T r;
  // jacobian (NumResiduals x NumEffectiveParameters) in the tangent space.
  bool Evaluate(const double* x, double* cost, double* residuals, double* gradient,
                DenseMatrix* jacobian, Summary* summary) const {
    const int nloc = p_->rotation_parameterization->LocalSize();
    const int ncols = 3 + nloc;
    const bool want_j = (jacobian != nullptr) || (gradient != nullptr);
    ++summary->num_residual_evaluations;
    if (want_j) ++summary->num_jacobian_evaluations;
    std::vector<double> local_res;
    double param_jac[12];
    if (want_j) p_->rotation_parameterization->ComputeJacobian(x + 3, param_jac);
    if (jacobian != nullptr) jacobian->Resize(NumResiduals(), ncols);
    if (gradient != nullptr)
      for (int i = 0; i < ncols; ++i) gradient[i] = 0.0;
    *cost = 0.0;
    int row = 0;
    for (const ResidualBlock& b : p_->blocks) {
      const int n = b.num_residuals;
      std::vector<double> scratch;
      double* r = residuals != nullptr ? residuals + row : nullptr;
      if (r == nullptr) {
        local_res.resize(n);
        r = local_res.data();
      }
      std::vector<double> jt, jq;
      if (want_j) {
        if (b.uses_translation) jt.resize(static_cast<size_t>(n) * 3);
        if (b.uses_rotation) jq.resize(static_cast<size_t>(n) * 4);
      }
      if (!b.evaluate(x, x + 3, r, jt.empty() ? nullptr : jt.data(),
                      jq.empty() ? nullptr : jq.data())) {
        return false;
      }
      double sq = 0.0;
      for (int i = 0; i < n; ++i) sq += r[i] * r[i];
      *cost += 0.5 * sq;
      if (want_j) {
        for (int i = 0; i < n; ++i) {
          double jrow[6] = {0, 0, 0, 0, 0, 0};
          if (b.uses_translation)
            for (int k = 0; k < 3; ++k) jrow[k] = jt[i * 3 + k];
          if (b.uses_rotation) {
            // local = global(1x4) * param_jac(4 x nloc)
            for (int c = 0; c < nloc; ++c) {
              double s = 0.0;
              for (int k = 0; k < 4; ++k) s += jq[i * 4 + k] * param_jac[k * nloc + c];
              jrow[3 + c] = s;
            }
          }
          if (jacobian != nullptr)
            for (int c = 0; c < ncols; ++c) (*jacobian)(row + i, c) = jrow[c];
          if (gradient != nullptr)
            for (int c = 0; c < ncols; ++c) gradient[c] += jrow[c] * r[i];
        }
      }
      row += n;
    }
    return true;
  }

 private:
  Problem* p_;
};

// Least squares solve min ||A y - b|| by Householder QR (what
// Eigen::HouseholderQR::solve does for the dense_qr_solver.cc path).
inline bool HouseholderQrSolve(DenseMatrix A, std::vector<double> b, double* y) {
  const int m = A.rows, n = A.cols;
  for (int k = 0; k < n; ++k) {
    double tail_sq = 0.0;
    for (int i = k + 1; i < m; ++i) tail_sq += A(i, k) * A(i, k);
    const double c0 = A(k, k);
    double beta, tau;
    if (tail_sq <= std::numeric_limits<double>::min()) {
      tau = 0.0;
      beta = c0;
      for (int i = k + 1; i < m; ++i) A(i, k) = 0.0;
    } else {
      beta = std::sqrt(c0 * c0 + tail_sq);
      if (c0 >= 0.0) beta = -beta;
      for (int i = k + 1; i < m; ++i) A(i, k) /= (c0 - beta);
      tau = (beta - c0) / beta;
    }
    A(k, k) = beta;
    if (tau != 0.0) {
      // apply H = I - tau v v^T (v0 = 1) to the remaining columns and to b
      for (int j = k + 1; j < n; ++j) {
        double s = A(k, j);
        for (int i = k + 1; i < m; ++i) s += A(i, k) * A(i, j);
        s *= tau;
        A(k, j) -= s;
        for (int i = k + 1; i < m; ++i) A(i, j) -= s * A(i, k);
      }
      double s = b[k];
      for (int i = k + 1; i < m; ++i) s += A(i, k) * b[i];
      s *= tau;
      b[k] -= s;
      for (int i = k + 1; i < m; ++i) b[i] -= s * A(i, k);
    }
  }
  for (int k = n - 1; k >= 0; --k) {
    double s = b[k];
    for (int j = k + 1; j < n; ++j) s -= A(k, j) * y[j];
    if (A(k, k) == 0.0) return false;
    y[k] = s / A(k, k);
  }
  for (int k = 0; k < n; ++k)
    if (!std::isfinite(y[k])) return false;
  return true;
}

// levenberg_marquardt_strategy.cc
class LevenbergMarquardtStrategy {
 public:
  explicit LevenbergMarquardtStrategy(const Options& o)
      : radius_(o.initial_trust_region_radius),
        max_radius_(o.max_trust_region_radius),
        min_diagonal_(o.min_lm_diagonal),
        max_diagonal_(o.max_lm_diagonal),
        decrease_factor_(2.0),
        reuse_diagonal_(false) {}

  // Solves for the step in the (column-scaled) tangent space; returns false on
  // linear solver failure.
  bool ComputeStep(const DenseMatrix& jacobian, const double* residuals, double* step) {
    const int n = jacobian.cols, m = jacobian.rows;
    if (!reuse_diagonal_) {
      diagonal_.assign(n, 0.0);
      for (int r = 0; r < m; ++r)
        for (int c = 0; c < n; ++c) diagonal_[c] += jacobian(r, c) * jacobian(r, c);
      for (int c = 0; c < n; ++c)
        diagonal_[c] = std::min(std::max(diagonal_[c], min_diagonal_), max_diagonal_);
    }
    std::vector<double> lm_diagonal(n);
    for (int c = 0; c < n; ++c) lm_diagonal[c] = std::sqrt(diagonal_[c] / radius_);
    // dense_qr_solver.cc: lhs = [J; diag(D)], rhs = [r; 0]; solve J y = r.
    DenseMatrix lhs;
    lhs.Resize(m + n, n);
    std::copy(jacobian.a.begin(), jacobian.a.end(), lhs.a.begin());
    for (int c = 0; c < n; ++c) lhs(m + c, c) = lm_diagonal[c];
    std::vector<double> rhs(m + n, 0.0);
    for (int r = 0; r < m; ++r) rhs[r] = residuals[r];
    const bool ok = HouseholderQrSolve(std::move(lhs), std::move(rhs), step);
    if (ok)
      for (int c = 0; c < n; ++c) step[c] *= -1.0;
    reuse_diagonal_ = true;
    return ok;
  }
  void StepAccepted(double step_quality) {
    radius_ = radius_ / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * step_quality - 1.0, 3));
    radius_ = std::min(max_radius_, radius_);
    decrease_factor_ = 2.0;
    reuse_diagonal_ = false;
  }
  void StepRejected(double /*step_quality*/) {
    radius_ = radius_ / decrease_factor_;
    decrease_factor_ *= 2.0;
    reuse_diagonal_ = true;
  }
  void StepIsInvalid() {
    radius_ *= 0.5;
    reuse_diagonal_ = true;
  }
  double Radius() const { return radius_; }

 private:
  double radius_, max_radius_, min_diagonal_, max_diagonal_, decrease_factor_;
  bool reuse_diagonal_;
  std::vector<double> diagonal_;
};

// trust_region_step_evaluator.cc
class TrustRegionStepEvaluator {
 public:
  TrustRegionStepEvaluator(double initial_cost, int max_consecutive_nonmonotonic_steps)
      : max_consecutive_nonmonotonic_steps_(max_consecutive_nonmonotonic_steps),
        minimum_cost_(initial_cost),
        current_cost_(initial_cost),
        reference_cost_(initial_cost),
        candidate_cost_(initial_cost),
        accumulated_reference_model_cost_change_(0.0),
        accumulated_candidate_model_cost_change_(0.0),
        num_consecutive_nonmonotonic_steps_(0) {}

  double StepQuality(double cost, double model_cost_change) const {
    const double relative_decrease = (current_cost_ - cost) / model_cost_change;
    const double historical_relative_decrease =
        (reference_cost_ - cost) /
        (accumulated_reference_model_cost_change_ + model_cost_change);
    return std::max(relative_decrease, historical_relative_decrease);
  }
  void StepAccepted(double cost, double model_cost_change) {
    current_cost_ = cost;
    accumulated_candidate_model_cost_change_ += model_cost_change;
    accumulated_reference_model_cost_change_ += model_cost_change;
    if (current_cost_ < minimum_cost_) {
      minimum_cost_ = current_cost_;
      num_consecutive_nonmonotonic_steps_ = 0;
      candidate_cost_ = current_cost_;
      accumulated_candidate_model_cost_change_ = 0.0;
    } else {
      ++num_consecutive_nonmonotonic_steps_;
      if (current_cost_ > candidate_cost_) {
        candidate_cost_ = current_cost_;
        accumulated_candidate_model_cost_change_ = 0.0;
      }
    }
    if (num_consecutive_nonmonotonic_steps_ == max_consecutive_nonmonotonic_steps_) {
      reference_cost_ = candidate_cost_;
      accumulated_reference_model_cost_change_ =
          accumulated_candidate_model_cost_change_;
    }
  }

 private:
  const int max_consecutive_nonmonotonic_steps_;
  double minimum_cost_, current_cost_, reference_cost_, candidate_cost_;
  double accumulated_reference_model_cost_change_;
  double accumulated_candidate_model_cost_change_;
  int num_consecutive_nonmonotonic_steps_;
};

// trust_region_minimizer.cc (1.13 structure: Init / IterationZero / loop).
// On return problem->t/q hold the best (minimum cost) iterate.
inline void Solve(const Options& options, Problem* problem, Summary* summary) {
  *summary = Summary();
  Evaluator evaluator(problem);
  const int num_params = kAmbient;
  const int num_eff = evaluator.NumEffectiveParameters();
  const int num_res = evaluator.NumResiduals();

  std::vector<double> x(num_params), candidate_x(num_params), best_x(num_params);
  for (int i = 0; i < 3; ++i) x[i] = problem->t[i];
  for (int i = 0; i < 4; ++i) x[3 + i] = problem->q[i];
  best_x = x;
  auto norm = [](const std::vector<double>& v) {
    double s = 0;
    for (double e : v) s += e * e;
    return std::sqrt(s);
  };
  double x_norm = norm(x);
  double x_cost = 0, candidate_cost = 0, minimum_cost = 0;
  std::vector<double> residuals(num_res), gradient(num_eff), jacobian_scaling(num_eff, 1.0);
  std::vector<double> trust_region_step(num_eff), delta(num_eff), model_residuals(num_res);
  std::vector<double> negative_gradient(num_eff), projected(num_params);
  DenseMatrix jacobian;
  LevenbergMarquardtStrategy strategy(options);
  int num_consecutive_invalid_steps = 0;
  double gradient_max_norm = 0;
  int iteration = 0;

  auto finish = [&](int type, const std::string& msg) {
    summary->termination_type = type;
    summary->message = msg;
    for (int i = 0; i < 3; ++i) problem->t[i] = best_x[i];
    for (int i = 0; i < 4; ++i) problem->q[i] = best_x[3 + i];
    // solver.cc SetSummaryFinalCost: min over iteration costs
    summary->final_cost = summary->initial_cost;
    for (double c : summary->iteration_costs)
      summary->final_cost = std::min(summary->final_cost, c);
    summary->num_iterations = static_cast<int>(summary->iteration_costs.size());
  };

  // EvaluateGradientAndJacobian
  auto evaluate_gradient_and_jacobian = [&]() -> bool {
    if (!evaluator.Evaluate(x.data(), &x_cost, residuals.data(), gradient.data(),
                            &jacobian, summary)) {
      return false;
    }
    if (options.jacobi_scaling) {
      if (iteration == 0) {
        for (int c = 0; c < num_eff; ++c) {
          double s = 0;
          for (int r = 0; r < num_res; ++r) s += jacobian(r, c) * jacobian(r, c);
          jacobian_scaling[c] = 1.0 / (1.0 + std::sqrt(s));
        }
      }
      for (int r = 0; r < num_res; ++r)
        for (int c = 0; c < num_eff; ++c) jacobian(r, c) *= jacobian_scaling[c];
    }
    // |Plus(x, -g) - x|_inf
    for (int c = 0; c < num_eff; ++c) negative_gradient[c] = -gradient[c];
    evaluator.Plus(x.data(), negative_gradient.data(), projected.data());
    gradient_max_norm = 0;
    for (int i = 0; i < num_params; ++i)
      gradient_max_norm = std::max(gradient_max_norm, std::fabs(x[i] - projected[i]));
    return true;
  };

  // ---- IterationZero
  if (!evaluate_gradient_and_jacobian()) {
    finish(2, "Initial residual and Jacobian evaluation failed.");
    return;
  }
  summary->initial_cost = x_cost;
  minimum_cost = x_cost;
  summary->iteration_costs.push_back(x_cost);
  if (gradient_max_norm <= options.gradient_tolerance) {
    finish(0, "Gradient tolerance reached.");
    return;
  }
  TrustRegionStepEvaluator step_evaluator(
      x_cost, options.use_nonmonotonic_steps ? options.max_consecutive_nonmonotonic_steps : 0);

  bool last_step_successful = false;  // iteration 0 is recorded as not successful
  // FinalizeIterationAndCheckIfMinimizerCanContinue for iteration 0 happens at
  // the top of the loop below.
  for (;;) {
    // -- FinalizeIterationAndCheckIfMinimizerCanContinue (of the previous iteration)
    if (last_step_successful) {
      ++summary->num_successful_steps;
      if (x_cost < minimum_cost) {
        minimum_cost = x_cost;
        best_x = x;
      }
    } else if (iteration > 0) {
      ++summary->num_unsuccessful_steps;
    }
    if (iteration >= options.max_num_iterations) {
      finish(1, "Maximum number of iterations reached.");
      return;
    }
    if (last_step_successful && gradient_max_norm <= options.gradient_tolerance) {
      finish(0, "Gradient tolerance reached.");
      return;
    }
    if (strategy.Radius() <= options.min_trust_region_radius) {
      finish(0, "Minimum trust region radius reached.");
      return;
    }

    ++iteration;
    last_step_successful = false;

    // -- ComputeTrustRegionStep
    bool step_is_valid = false;
    double model_cost_change = 0;
    const bool solved =
        strategy.ComputeStep(jacobian, residuals.data(), trust_region_step.data());
    if (solved) {
      for (int r = 0; r < num_res; ++r) {
        double s = 0;
        for (int c = 0; c < num_eff; ++c) s += jacobian(r, c) * trust_region_step[c];
        model_residuals[r] = s;
      }
      double dot = 0;
      for (int r = 0; r < num_res; ++r)
        dot += model_residuals[r] * (residuals[r] + model_residuals[r] / 2.0);
      model_cost_change = -dot;
      step_is_valid = model_cost_change > 0.0;
    }
    if (!step_is_valid) {
      // HandleInvalidStep
      if (++num_consecutive_invalid_steps >= options.max_num_consecutive_invalid_steps) {
        finish(2, "Number of consecutive invalid steps more than max.");
        return;
      }
      strategy.StepIsInvalid();
      summary->iteration_costs.push_back(x_cost);
      continue;
    }
    num_consecutive_invalid_steps = 0;
    for (int c = 0; c < num_eff; ++c) delta[c] = trust_region_step[c] * jacobian_scaling[c];

    // -- ComputeCandidatePointAndEvaluateCost
    evaluator.Plus(x.data(), delta.data(), candidate_x.data());
    if (!evaluator.Evaluate(candidate_x.data(), &candidate_cost, nullptr, nullptr, nullptr,
                            summary)) {
      candidate_cost = std::numeric_limits<double>::max();
    }

    // -- ParameterToleranceReached
    double step_norm = 0;
    for (int i = 0; i < num_params; ++i)
      step_norm += (x[i] - candidate_x[i]) * (x[i] - candidate_x[i]);
    step_norm = std::sqrt(step_norm);
    const double step_size_tolerance =
        options.parameter_tolerance * (x_norm + options.parameter_tolerance);
    if (step_norm <= step_size_tolerance) {
      finish(0, "Parameter tolerance reached.");
      return;
    }
    // -- FunctionToleranceReached
    const double cost_change = x_cost - candidate_cost;
    if (std::fabs(cost_change) <= options.function_tolerance * x_cost) {
      finish(0, "Function tolerance reached.");
      return;
    }

    // -- IsStepSuccessful
    const double relative_decrease =
        step_evaluator.StepQuality(candidate_cost, model_cost_change);
    if (relative_decrease > options.min_relative_decrease) {
      // HandleSuccessfulStep
      x = candidate_x;
      x_norm = norm(x);
      if (!evaluate_gradient_and_jacobian()) {
        finish(2, "Residual and Jacobian evaluation failed.");
        return;
      }
      last_step_successful = true;
      strategy.StepAccepted(relative_decrease);
      step_evaluator.StepAccepted(candidate_cost, model_cost_change);
      summary->iteration_costs.push_back(x_cost);
    } else {
      // HandleUnsuccessfulStep
      strategy.StepRejected(relative_decrease);
      summary->iteration_costs.push_back(candidate_cost);
    }
  }
}

}  // namespace ceres_like
}  // namespace oracle

#endif  // ORACLE_OM_CERES_H_
