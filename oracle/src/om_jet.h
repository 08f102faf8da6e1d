// ORACLE (test infrastructure, NOT product code).
//
// Forward-mode dual numbers with N infinitesimal parts, restating the
// arithmetic of ceres::Jet<double, N> (Ceres Solver 1.13.0,
// include/ceres/jet.h -- third-party, NOT in /root/reference; pinned by
// scripts/install_ceres.sh VERSION="1.13.0" and bazel/repositories.bzl:128).
// Only the operators the cost functors of the path use are provided.  The
// scalar-part formulas matter for bit-level agreement of costs:
//   Jet / scalar  ->  multiply by (1.0 / scalar)
//   Jet / Jet     ->  g_inv = 1/g.a ; a = f.a * g_inv ; v = (f.v - a*g.v) * g_inv
#ifndef ORACLE_OM_JET_H_
#define ORACLE_OM_JET_H_

#include <cmath>

namespace oracle {

template <int N>
struct Jet {
  double a;
  double v[N];

  Jet() : a(0.0) {
    for (int i = 0; i < N; ++i) v[i] = 0.0;
  }
  explicit Jet(double value) : a(value) {
    for (int i = 0; i < N; ++i) v[i] = 0.0;
  }
  Jet(double value, int k) : a(value) {
    for (int i = 0; i < N; ++i) v[i] = 0.0;
    v[k] = 1.0;
  }
};

template <int N>
Jet<N> operator-(const Jet<N>& f) {
  Jet<N> r;
  r.a = -f.a;
  for (int i = 0; i < N; ++i) r.v[i] = -f.v[i];
  return r;
}
template <int N>
Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> r;
  r.a = f.a + g.a;
  for (int i = 0; i < N; ++i) r.v[i] = f.v[i] + g.v[i];
  return r;
}
template <int N>
Jet<N> operator+(const Jet<N>& f, double s) {
  Jet<N> r = f;
  r.a = f.a + s;
  return r;
}
template <int N>
Jet<N> operator+(double s, const Jet<N>& f) {
  Jet<N> r = f;
  r.a = f.a + s;
  return r;
}
template <int N>
Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> r;
  r.a = f.a - g.a;
  for (int i = 0; i < N; ++i) r.v[i] = f.v[i] - g.v[i];
  return r;
}
template <int N>
Jet<N> operator-(const Jet<N>& f, double s) {
  Jet<N> r = f;
  r.a = f.a - s;
  return r;
}
template <int N>
Jet<N> operator-(double s, const Jet<N>& f) {
  Jet<N> r;
  r.a = s - f.a;
  for (int i = 0; i < N; ++i) r.v[i] = -f.v[i];
  return r;
}
template <int N>
Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> r;
  r.a = f.a * g.a;
  for (int i = 0; i < N; ++i) r.v[i] = f.a * g.v[i] + f.v[i] * g.a;
  return r;
}
template <int N>
Jet<N> operator*(const Jet<N>& f, double s) {
  Jet<N> r;
  r.a = f.a * s;
  for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * s;
  return r;
}
template <int N>
Jet<N> operator*(double s, const Jet<N>& f) {
  Jet<N> r;
  r.a = f.a * s;
  for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * s;
  return r;
}
template <int N>
Jet<N> operator/(const Jet<N>& f, double s) {
  const double s_inverse = 1.0 / s;
  Jet<N> r;
  r.a = f.a * s_inverse;
  for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * s_inverse;
  return r;
}
template <int N>
Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  const double g_a_inverse = 1.0 / g.a;
  const double f_a_by_g_a = f.a * g_a_inverse;
  Jet<N> r;
  r.a = f_a_by_g_a;
  for (int i = 0; i < N; ++i) r.v[i] = (f.v[i] - f_a_by_g_a * g.v[i]) * g_a_inverse;
  return r;
}

// "scalar part" accessor usable on both double and Jet.
inline double ScalarPart(double x) { return x; }
template <int N>
double ScalarPart(const Jet<N>& x) {
  return x.a;
}

}  // namespace oracle

#endif  // ORACLE_OM_JET_H_
