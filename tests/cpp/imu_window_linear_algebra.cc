// The two pieces of imu_window.cc's arithmetic that round 5 restructured for speed must give the values of the plain
// forms they replaced: (1) cholesky_chain -- right-looking on the block-tridiagonal normal matrix -- against the
// dot-product (left-looking) form, bit for bit; (2) propagate_covariance -- sums over the entries of A, B, C that can
// be non-zero -- against the full 9 x 9 x 9 loops (== on doubles: a skipped term was an exact zero).
#include "../../d-liom_amd/csrc/imu_window.cc"

#include <cstdio>
#include <cstring>

static int fail(const char* what) {
  std::printf("FAIL: %s\n", what);
  return 1;
}

static bool cholesky_chain_plain(std::vector<double>& a, int n, int block) {
  for (int j = 0; j < n; ++j) {
    const int k0 = std::max(0, (j / block - 1) * block);
    double d = a[j * n + j];
    for (int k = k0; k < j; ++k) d -= a[j * n + k] * a[j * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    a[j * n + j] = d;
    const int i_end = std::min(n, (j / block + 2) * block);
    for (int i = j + 1; i < i_end; ++i) {
      const int ki = std::max(k0, (i / block - 1) * block);
      double s = a[i * n + j];
      for (int k = ki; k < j; ++k) s -= a[i * n + k] * a[j * n + k];
      a[i * n + j] = s / d;
    }
    for (int i = i_end; i < n; ++i) a[i * n + j] = 0.0;
  }
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
  // (1) banded Cholesky, windows of 1 .. 8 states, also a matrix that is not positive definite
  for (int states = 1; states <= 8; ++states)
    for (int rep = 0; rep < 20; ++rep) {
      const int n = states * kD;
      std::vector<double> A(static_cast<size_t>(n) * n, 0.0);
      for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j)
          if (i / kD - j / kD <= 1) A[i * n + j] = A[j * n + i] = (i == j ? (rep == 19 ? 1.0 : 30.0) : 0.0) + rnd();
      std::vector<double> a = A, b = A;
      const bool oa = cholesky_chain(a, n, kD), ob = cholesky_chain_plain(b, n, kD);
      if (oa != ob) return fail("cholesky_chain: success differs");
      if (!oa) continue;
      for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j)
          if (std::memcmp(&a[i * n + j], &b[i * n + j], 8) != 0) return fail("cholesky_chain: bits differ");
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
