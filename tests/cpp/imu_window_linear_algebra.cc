// The pieces of imu_window.cc's arithmetic that were restructured for speed must give the values of the plain forms they
// replaced: (1) the chain solver (round 6: block elimination of the block-tridiagonal normal equations from the oldest
// key on, the older part reused from scan to scan) against ONE dense Cholesky solve of the same factors; (2) propagate_covariance -- sums over the entries of A, B, C that can
// be non-zero -- against the full 9 x 9 x 9 loops (== on doubles: a skipped term was an exact zero).
#include "../../d-liom_amd/csrc/imu_window.cc"

#include <cstdio>
#include <cstring>

static int fail(const char* what) {
  std::printf("FAIL: %s\n", what);
  return 1;
}

// The window's normal equations as one dense matrix, every factor at the linearisation points w.x, and a dense Cholesky
// solve: the plain form of what the chain solver (block elimination from the oldest state on, reused between scans) does.
static bool dense_solve(dliom_imu_window& w, std::vector<double>* delta) {
  const int N = static_cast<int>(w.x.size()), n = N * kD;
  std::vector<double> H(static_cast<size_t>(n) * n, 0.0), g(n, 0.0);
  add_state0_prior(w, w.x[0], H.data(), g.data(), n);
  for (int i = 0; i + 1 < N; ++i) {
    std::vector<double> Linv;
    if (!whitening(w.between[i], &Linv)) return false;
    double r[15], J[15 * 30];
    imu_factor_jacobian(w, w.between[i], Linv, w.x[i], w.x[i + 1], r, J);
    accumulate_imu_factor(i, r, J, H, g, n);
  }
  for (const auto& f : w.pose_priors) {
    double r[6], J[6 * kD];
    pose_prior_jacobian(f, w.x[f.index], r, J);
    accumulate_pose_prior(f.index, r, J, H, g, n);
  }
  for (const auto& f : w.gravity) {
    double r[2], J[2 * kD];
    gravity_residual(f, w.x[f.index], r, J);
    add_factor_with_jacobian(f.index, 2, r, J, H, g, n);
  }
  for (int i = 0; i < n; ++i) H[static_cast<size_t>(i) * n + i] += 1e-12;
  if (!cholesky(H, n)) return false;
  delta->assign(n, 0.0);
  for (int i = 0; i < n; ++i) (*delta)[i] = -g[i];
  chol_solve(H, n, delta->data());
  return true;
}

static void propagate_plain(const double (&A)[81], const double (&Bm)[27], const double (&Cm)[27], double qa, double qg, double* cov) {
  double tmp[81], next[81];
  for (int i = 0; i < 9; ++i)
    for (int j = 0; j < 9; ++j) {
      double s = 0;
      for (int k = 0; k < 9; ++k) s += A[9 * i + k] * cov[9 * k + j];
      tmp[9 * i + j] = s;
    }
  for (int i = 0; i < 9; ++i)
    for (int j = 0; j < 9; ++j) {
      double s = 0;
      for (int k = 0; k < 9; ++k) s += tmp[9 * i + k] * A[9 * j + k];
      for (int k = 0; k < 3; ++k) s += qa * Bm[3 * i + k] * Bm[3 * j + k] + qg * Cm[3 * i + k] * Cm[3 * j + k];
      next[9 * i + j] = s;
    }
  std::memcpy(cov, next, sizeof next);
}

int main() {
  unsigned seed = 20260926u;
  auto rnd = [&]() {
    seed = seed * 1664525u + 1013904223u;
    return static_cast<double>(seed >> 8) / (1 << 24) - 0.5;
  };
  // (1) the chain solver against the dense solve: a graph of up to 40 keys that keeps its linearisation points (threshold
  // never reached), so that most of every scan's elimination is reused from the scan before; gravity factors reach
  // three keys back.  The increments are the dense solution's to rounding.
  {
    dliom_imu_window_options o;
    if (dliom_imu_window_default_options(&o) != DLIOM_OK) return fail("default options");
    o.window_size = 0;
    o.graph_reset_every = 41;
    o.relinearize_threshold = 1e6;
    dliom_imu_window* w = nullptr;
    if (dliom_imu_window_create(&o, &w) != DLIOM_OK) return fail("create");
    const double pose0[7] = {0, 0, 0, 1, 0, 0, 0}, v0[3] = {1.0, 0.2, 0}, b0[6] = {0, 0, 0, 0, 0, 0};
    if (dliom_imu_window_initialize(w, pose0, v0, b0) != DLIOM_OK) return fail("initialize");
    int64_t eliminated_before = 0;
    for (int k = 1; k <= 39; ++k) {
      for (int i = 0; i < 20; ++i) {
        const double acc[3] = {0.3 * rnd(), 0.3 * rnd(), 9.80511 + 0.1 * rnd()}, gyr[3] = {0.05 * rnd(), 0.05 * rnd(), 0.2 + 0.05 * rnd()};
        if (dliom_imu_window_add_imu(w, acc, gyr, 0.005) != DLIOM_OK) return fail("add_imu");
      }
      double pred[7], vel[3];
      if (dliom_imu_window_predict(w, pred, vel) != DLIOM_OK) return fail("predict");
      for (int c = 0; c < 3; ++c) pred[c] += 0.01 * rnd();  // the matcher's correction
      if (k % 5 == 0 && k >= 3) {
        const double down[3] = {0.01 * rnd(), 0.01 * rnd(), -1.0};
        if (dliom_imu_window_add_gravity(w, 3, down) != DLIOM_OK) return fail("add_gravity");
      }
      double out[7], ov[3], ob[6];
      if (dliom_imu_window_add_pose(w, pred, 0, out, ov, ob) != DLIOM_OK) return fail("add_pose");
      std::vector<double> want;
      if (!dense_solve(*w, &want)) return fail("dense solve");
      double worst = 0.0, scale = 0.0;
      for (size_t i = 0; i < want.size(); ++i) {
        worst = std::max(worst, std::fabs(want[i] - w->delta[i]));
        scale = std::max(scale, std::fabs(want[i]));
      }
      if (!(worst <= 1e-9 * std::max(scale, 1e-3))) {
        std::printf("key %d: chain and dense increments differ by %.3g (largest increment %.3g)\n", k, worst, scale);
        return fail("chain solver");
      }
      int64_t relin = 0, eliminated = 0;
      dliom_imu_window_solver_stats(w, &relin, &eliminated);
      // a scan re-eliminates the two newest keys (+ the keys back to a gravity factor's), not the graph
      if (relin != 0 || eliminated - eliminated_before > 5) return fail("chain solver: elimination not reused");
      eliminated_before = eliminated;
    }
    dliom_imu_window_destroy(w);
  }
  // (2) covariance propagation with A, B, C of the shape both preintegration forms produce
  for (int rep = 0; rep < 200; ++rep) {
    double A[81] = {0}, Bm[27] = {0}, Cm[27] = {0}, cov[81], c1[81], c2[81];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        A[9 * i + j] = (i == j ? 1.0 : 0.0) + 0.01 * rnd();
        A[9 * (3 + i) + j] = 1e-4 * rnd();
        A[9 * (6 + i) + j] = 1e-2 * rnd();
        Bm[3 * (3 + i) + j] = 1e-5 * rnd();
        Bm[3 * (6 + i) + j] = 5e-3 * rnd();
        Cm[3 * i + j] = 5e-3 * rnd();
      }
    for (int i = 0; i < 3; ++i) {
      A[9 * (3 + i) + 3 + i] = 1.0;
      A[9 * (3 + i) + 6 + i] = 0.005;
      A[9 * (6 + i) + 6 + i] = 1.0;
    }
    for (int i = 0; i < 9; ++i)
      for (int j = 0; j <= i; ++j) cov[9 * i + j] = cov[9 * j + i] = (i == j ? 1e-4 : 0.0) + 1e-5 * rnd();
    if (rep == 0) std::memset(cov, 0, sizeof cov);  // the first sample after a reset
    std::memcpy(c1, cov, sizeof cov);
    std::memcpy(c2, cov, sizeof cov);
    propagate_covariance(A, Bm, Cm, 0.08 * 0.08 / 0.005, 0.004 * 0.004 / 0.005, c1);
    propagate_plain(A, Bm, Cm, 0.08 * 0.08 / 0.005, 0.004 * 0.004 / 0.005, c2);
    for (int i = 0; i < 81; ++i)
      if (!(c1[i] == c2[i])) return fail("propagate_covariance: values differ");
  }
  std::printf("OK\n");
  return 0;
}
