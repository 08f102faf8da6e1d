// LocalTrajectoryBuilder3D::WindowOptimize without GTSAM: the IMU-preintegration
// cost, the bias random walk, the matched-pose prior and the gravity factor, solved as a fixed-lag smoother.
//
// Replaces mapping/internal/3d/local_trajectory_builder_3d.cc:693-863 (WindowOptimize), :179-199 (AddImuData's
// integrateMeasurement / predict) and gravity_factor/gravity_factor.cc:10-33.  GTSAM 4.0.2 is not in the tree
// (README.MD:12): its factors are restated from their published definitions, PARITY UNPINNED (the reference
// has no test at this boundary):
//   preintegration   TWO forms behind dliom_imu_window_options::tangent_preintegration:
//                    1 (default) TangentPreintegration -- what the reference's binary holds: README.MD:13-15 builds GTSAM
//                      4.0.2 without -DGTSAM_TANGENT_PREINTEGRATION=OFF and 4.0.x defaults to ON, so
//                      PreintegratedImuMeasurements (local_trajectory_builder_3d.cc:76-104,188-199) integrates the vector
//                      [theta, p, v] in the tangent space of the first NavState: theta += Jr(theta)^-1 w dt,
//                      p += v dt + R(theta) a dt^2 / 2, v += R(theta) a dt, bias Jacobians H <- A H - [B | C], covariance
//                      A S A^T + B (Sa / dt) B^T + C (Sw / dt) C^T + (1e-4)^2 dt on the position block (:79-82), and the
//                      factor's error is the NavState local coordinates of the predicted state at state j:
//                      [Log(Rj^T R*), Rj^T (p* - pj), Rj^T (v* - vj)], R* = Ri Exp(theta(b)), p* = pi + vi dt + g dt^2/2 +
//                      Ri P(b), v* = vi + g dt + Ri V(b);
//                    0 the manifold form (Forster et al.; GTSAM with the flag OFF): Delta R, Delta p, Delta v, error
//                      r_R = Log(DR(bg)^T Ri^T Rj), r_p = Ri^T (pj - pi - vi dt - g dt^2 / 2) - Dp(b),
//                      r_v = Ri^T (vj - vi - g dt) - Dv(b).
//                    Both whitened with the preintegrated 9 x 9 covariance, n_gravity = (0,0,-g).  DESIGN 3.8 quantifies
//                    the difference between the two on tools/stream.py's run (it is small: they differ in second order
//                    of the rotation within one scan interval and in the frame the residual is expressed in).
//   ImuFactor        see above
//   BetweenFactor    bias_j - bias_i with sigma = sqrt(dt) (acc_bias_noise x3, gyr_bias_noise x3)        (:808-812)
//   PriorFactor      Pose3 local coordinates [Log(R0^T R), R0^T (p - p0)] with the sigmas IN THE ORDER THE REFERENCE
//                    FILLS THEM, (t, t, t, r, r, r) (:94-101): the rotation rows get ceres_pose_noise_t -- kept
//   gravity factor   error = basis(nZ)^T (RzRyRx(roll, pitch, 0) bRef) + 1e-5 per component (gravity_factor.cc:10-33), with
//                    Unit3::basis() as GTSAM 4.0.2 builds it (b1 = n x e_k / |.|, e_k the coordinate axis along which |n|
//                    is smallest, ties x before y before z; b2 = n x b1) and the factor's OWN Jacobian
//                    H = B_nZ^T B_q (-B_q^T R_rp [bRef]x) on the rotation block -- the derivative of rotating bRef by
//                    R_rp = Ry(pitch) Rx(roll) taken as if R_rp were the state's rotation (the chain through xyz() is not
//                    in the reference's factor either)
//   EstimateGravity  :1106-1154 + gravity_factor/gravity_estimator.cc (in the tree, followed line by line, quirks kept):
//                    a deque of (pose of the previous key, copy of the running preintegration, previous velocity), linear
//                    solve for g in the first frame + four tangent-plane refinements; the factor goes on the state
//                    frames_for_online_gravity_estimate keys back when the estimate passes the reference's two gates
// ISAM2 (two update() calls per scan, graph reset with the marginal covariances every num_range_data keys,
// :750-797,841-842) becomes: Gauss-Newton (two iterations per scan) over the last `window_size` states, older states
// marginalised into a Gaussian prior on the oldest kept one (Schur complement).  For a linear problem both give the
// same estimate; the difference is where the old factors are linearised.
// Jacobians are closed-form (IMU factor + bias random walk, pose prior, gravity factor); dliom_diag_imu_factor_jacobians
// compares the IMU factor's with central differences of its own residual.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <memory>
#include <vector>

#include "../../include/dliom.h"

namespace {

struct V3 {
  double x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }

struct M3 {
  double m[9];  // row major
};
inline M3 identity() { return {{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
inline M3 operator*(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
  return r;
}
inline V3 operator*(const M3& a, V3 v) {
  return {a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
          a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z};
}
inline M3 transpose(const M3& a) { return {{a.m[0], a.m[3], a.m[6], a.m[1], a.m[4], a.m[7], a.m[2], a.m[5], a.m[8]}}; }
inline M3 scaled(const M3& a, double s) {
  M3 r = a;
  for (double& v : r.m) v *= s;
  return r;
}
inline M3 add(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] + b.m[i];
  return r;
}
inline M3 skew(V3 v) { return {{0, -v.z, v.y, v.z, 0, -v.x, -v.y, v.x, 0}}; }

// SO(3) exponential / logarithm / right Jacobian
M3 Exp(V3 w) {
  const double t = norm(w);
  const M3 K = skew(w);
  const M3 K2 = K * K;
  double a, b;
  if (t < 1e-6) {
    a = 1.0 - t * t / 6.0;
    b = 0.5 - t * t / 24.0;
  } else {
    a = std::sin(t) / t;
    b = (1.0 - std::cos(t)) / (t * t);
  }
  return add(identity(), add(scaled(K, a), scaled(K2, b)));
}
V3 Log(const M3& R) {
  const double tr = R.m[0] + R.m[4] + R.m[8];
  const V3 v{R.m[7] - R.m[5], R.m[2] - R.m[6], R.m[3] - R.m[1]};
  const double c = std::min(1.0, std::max(-1.0, 0.5 * (tr - 1.0)));
  const double t = std::acos(c);
  if (t < 1e-6) return 0.5 * (1.0 + t * t / 6.0) * v;
  if (t > 3.14159265358979 - 1e-6) {  // near pi: from the symmetric part
    const double xx = std::sqrt(std::max(0.0, 0.5 * (R.m[0] + 1.0))), yy = std::sqrt(std::max(0.0, 0.5 * (R.m[4] + 1.0))),
                 zz = std::sqrt(std::max(0.0, 0.5 * (R.m[8] + 1.0)));
    V3 ax{xx, yy, zz};
    if (v.x < 0) ax.x = -ax.x;
    if (v.y < 0) ax.y = -ax.y;
    if (v.z < 0) ax.z = -ax.z;
    const double n = norm(ax);
    return n > 0 ? (t / n) * ax : V3{t, 0, 0};
  }
  return (t / (2.0 * std::sin(t))) * v;
}
M3 RightJacobian(V3 w) {
  const double t = norm(w);
  const M3 K = skew(w);
  const M3 K2 = K * K;
  double a, b;
  if (t < 1e-6) {
    a = 0.5 - t * t / 24.0;
    b = 1.0 / 6.0 - t * t / 120.0;
  } else {
    a = (1.0 - std::cos(t)) / (t * t);
    b = (t - std::sin(t)) / (t * t * t);
  }
  return add(identity(), add(scaled(K, -a), scaled(K2, b)));
}
M3 RightJacobianInverse(V3 w) {  // Jr^-1(w) = I + [w]x / 2 + (1 / t^2 - (1 + cos t) / (2 t sin t)) [w]x^2
  const double t = norm(w);
  const M3 K = skew(w);
  const M3 K2 = K * K;
  const double c = t < 1e-5 ? 1.0 / 12.0 + t * t / 720.0 : 1.0 / (t * t) - (1.0 + std::cos(t)) / (2.0 * t * std::sin(t));
  return add(identity(), add(scaled(K, 0.5), scaled(K2, c)));
}
M3 quat_to_matrix(const double q[4]) {  // (w, x, y, z)
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
  return {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z),
           2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
}
void matrix_to_quat(const M3& R, double q[4]) {
  const double tr = R.m[0] + R.m[4] + R.m[8];
  if (tr > 0) {
    const double s = std::sqrt(tr + 1.0) * 2;
    q[0] = 0.25 * s;
    q[1] = (R.m[7] - R.m[5]) / s;
    q[2] = (R.m[2] - R.m[6]) / s;
    q[3] = (R.m[3] - R.m[1]) / s;
  } else if (R.m[0] > R.m[4] && R.m[0] > R.m[8]) {
    const double s = std::sqrt(1.0 + R.m[0] - R.m[4] - R.m[8]) * 2;
    q[0] = (R.m[7] - R.m[5]) / s;
    q[1] = 0.25 * s;
    q[2] = (R.m[1] + R.m[3]) / s;
    q[3] = (R.m[2] + R.m[6]) / s;
  } else if (R.m[4] > R.m[8]) {
    const double s = std::sqrt(1.0 + R.m[4] - R.m[0] - R.m[8]) * 2;
    q[0] = (R.m[2] - R.m[6]) / s;
    q[1] = (R.m[1] + R.m[3]) / s;
    q[2] = 0.25 * s;
    q[3] = (R.m[5] + R.m[7]) / s;
  } else {
    const double s = std::sqrt(1.0 + R.m[8] - R.m[0] - R.m[4]) * 2;
    q[0] = (R.m[3] - R.m[1]) / s;
    q[1] = (R.m[2] + R.m[6]) / s;
    q[2] = (R.m[5] + R.m[7]) / s;
    q[3] = 0.25 * s;
  }
  if (q[0] < 0)
    for (int i = 0; i < 4; ++i) q[i] = -q[i];
}

// ---- state: rotation, position, velocity, accelerometer bias, gyroscope bias; tangent order the same (15)
struct State {
  M3 R;
  V3 p, v, ba, bg;
};
State retract(const State& s, const double* d) {
  State r = s;
  r.R = s.R * Exp({d[0], d[1], d[2]});
  r.p = s.p + V3{d[3], d[4], d[5]};
  r.v = s.v + V3{d[6], d[7], d[8]};
  r.ba = s.ba + V3{d[9], d[10], d[11]};
  r.bg = s.bg + V3{d[12], d[13], d[14]};
  return r;
}
void local(const State& origin, const State& s, double* d) {  // s = retract(origin, d)
  const V3 w = Log(transpose(origin.R) * s.R), dp = s.p - origin.p, dv = s.v - origin.v, da = s.ba - origin.ba, dg = s.bg - origin.bg;
  const double out[15] = {w.x, w.y, w.z, dp.x, dp.y, dp.z, dv.x, dv.y, dv.z, da.x, da.y, da.z, dg.x, dg.y, dg.z};
  std::memcpy(d, out, sizeof out);
}

// ---- preintegration --------------------------------------------------------------------------------
struct Preint {
  double dt = 0.0;
  M3 dR = identity();
  V3 dp{0, 0, 0}, dv{0, 0, 0};
  // manifold form: d(Delta R)/d(bg) as a right perturbation; tangent form: d(theta)/d(bg) (the same slot: dR_dbg)
  M3 dR_dbg{}, dp_dba{}, dp_dbg{}, dv_dba{}, dv_dbg{};
  double cov[81];  // rotation, position, velocity
  V3 ba_lin{0, 0, 0}, bg_lin{0, 0, 0};
  bool tangent = false;  // GTSAM's TangentPreintegration: `th` is the integrated quantity, dR = Exp(th) is kept beside it
  V3 th{0, 0, 0};
  M3 dth_dba{};          // tangent form only (zero in exact arithmetic: kept because GTSAM carries the block)
  Preint() { reset({0, 0, 0}, {0, 0, 0}); }
  void reset(V3 ba, V3 bg) {
    dt = 0.0;
    dR = identity();
    dp = dv = V3{0, 0, 0};
    th = V3{0, 0, 0};
    std::memset(&dR_dbg, 0, sizeof(M3));
    dp_dba = dp_dbg = dv_dba = dv_dbg = dth_dba = dR_dbg;
    std::memset(cov, 0, sizeof cov);
    ba_lin = ba;
    bg_lin = bg;
  }
};

// d(Jr(theta) c)/d(theta) for a fixed c (GTSAM so3::DexpFunctor::applyDexp's H1):
//   Jr(theta) c = c - alpha theta x c + beta theta x (theta x c),  alpha = (1 - cos t) / t^2,  beta = (t - sin t) / t^3
M3 DexpDerivative(V3 theta, V3 c) {
  const double t2 = dot(theta, theta), t = std::sqrt(t2);
  if (t2 <= 2.220446049250313e-16) return scaled(skew(c), 0.5);  // Jr ~ I - [theta]x / 2
  const double alpha = (1.0 - std::cos(t)) / t2, beta = (t - std::sin(t)) / (t2 * t);
  const double dalpha = (t * std::sin(t) - 2.0 * (1.0 - std::cos(t))) / (t2 * t);       // d alpha / dt
  const double dbeta = (t * (1.0 - std::cos(t)) - 3.0 * (t - std::sin(t))) / (t2 * t2);  // d beta / dt
  const V3 txc = cross(theta, c), ttc = cross(theta, txc);
  const double tc = dot(theta, c);
  M3 D = scaled(skew(c), alpha);
  auto outer_add = [&D](V3 a, V3 b, double s) {  // D += s a b^T
    const double av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) D.m[3 * i + j] += s * av[i] * bv[j];
  };
  outer_add(txc, theta, -dalpha / t);
  for (int i = 0; i < 3; ++i) D.m[4 * i] += beta * tc;
  outer_add(theta, c, beta);
  outer_add(c, theta, -2.0 * beta);
  outer_add(ttc, theta, dbeta / t);
  return D;
}

// gtsam::TangentPreintegration::update + PreintegratedImuMeasurements::integrateMeasurement (GTSAM 4.0.2,
// navigation/TangentPreintegration.cpp, ImuFactor.cpp), restated from their published definitions.
// cov <- A cov A^T + (Sa / h) B B^T + (Sw / h) C C^T for the preintegration's 9 x 9 covariance (rotation, position,
// velocity).  Both forms of A have the same shape -- rows 0-2: columns 0-2; rows 3-5: columns 0-2, the diagonal and
// the velocity of the same axis; rows 6-8: columns 0-2 and the diagonal -- B is zero in rows 0-2 and C in rows 3-8.
// The sums run over the entries that can be non-zero, in the order of the full loops (k ascending), so the values are
// the ones the full products gave (a skipped term was an exact zero); 1 100 multiply-adds instead of 2 400 per sample
// with the bias Jacobians below: integrate() 1.3 us -> 0.8 us, twenty calls per scan.
struct RowPattern {
  int n;
  int k[5];
};
inline RowPattern row_pattern(int r) {
  if (r < 3) return RowPattern{3, {0, 1, 2, 0, 0}};
  if (r < 6) return RowPattern{5, {0, 1, 2, r, r + 3}};
  return RowPattern{4, {0, 1, 2, r, 0}};
}
void propagate_covariance(const double (&A)[81], const double (&Bm)[27], const double (&Cm)[27], double qa, double qg, double* cov) {
  double tmp[81], next[81];
  for (int i = 0; i < 9; ++i) {
    const RowPattern pi = row_pattern(i);
    for (int j = 0; j < 9; ++j) {
      double s = 0;
      for (int q = 0; q < pi.n; ++q) s += A[9 * i + pi.k[q]] * cov[9 * pi.k[q] + j];
      tmp[9 * i + j] = s;
    }
  }
  for (int i = 0; i < 9; ++i)
    for (int j = 0; j < 9; ++j) {
      const RowPattern pj = row_pattern(j);
      double s = 0;
      for (int q = 0; q < pj.n; ++q) s += tmp[9 * i + pj.k[q]] * A[9 * j + pj.k[q]];
      if (i >= 3 && j >= 3) {
        for (int k = 0; k < 3; ++k) s += qa * Bm[3 * i + k] * Bm[3 * j + k];
      } else if (i < 3 && j < 3) {
        for (int k = 0; k < 3; ++k) s += qg * Cm[3 * i + k] * Cm[3 * j + k];
      }
      next[9 * i + j] = s;
    }
  std::memcpy(cov, next, sizeof next);
}

void integrate_tangent(Preint& P, V3 acc_meas, V3 gyr_meas, double h, double acc_sigma, double gyr_sigma, double int_sigma) {
  const V3 a = acc_meas - P.ba_lin, w = gyr_meas - P.bg_lin;
  const M3 Jr = RightJacobian(P.th), invJ = RightJacobianInverse(P.th);
  const V3 w_tangent = invJ * w;
  const M3 R = Exp(P.th);
  const V3 a_nav = R * a;
  const double h22 = 0.5 * h * h;
  // A = d(new)/d(old), B = d(new)/d(acc), C = d(new)/d(omega)
  const M3 w_tangent_H_theta = scaled(invJ * DexpDerivative(P.th, w_tangent), -1.0);
  const M3 a_nav_H_theta = (R * skew(-1.0 * a)) * Jr;
  double A[81] = {0}, Bm[27] = {0}, Cm[27] = {0};
  for (int i = 0; i < 9; ++i) A[10 * i] = 1.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      A[9 * i + j] += h * w_tangent_H_theta.m[3 * i + j];
      A[9 * (3 + i) + j] = h22 * a_nav_H_theta.m[3 * i + j];
      A[9 * (6 + i) + j] = h * a_nav_H_theta.m[3 * i + j];
      Bm[3 * (3 + i) + j] = h22 * R.m[3 * i + j];
      Bm[3 * (6 + i) + j] = h * R.m[3 * i + j];
      Cm[3 * i + j] = h * invJ.m[3 * i + j];
    }
  for (int i = 0; i < 3; ++i) A[9 * (3 + i) + 6 + i] = h;
  const double qa = acc_sigma * acc_sigma / h, qg = gyr_sigma * gyr_sigma / h;
  propagate_covariance(A, Bm, Cm, qa, qg, P.cov);
  for (int i = 0; i < 3; ++i) P.cov[9 * (3 + i) + 3 + i] += int_sigma * int_sigma * h;
  // preintegrated_H_biasAcc = A H_a - B, preintegrated_H_biasOmega = A H_g - C (9 x 3 each, kept as three 3 x 3 blocks)
  double Ha[27], Hg[27], Ha2[27], Hg2[27];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Ha[3 * i + j] = P.dth_dba.m[3 * i + j];
      Ha[3 * (3 + i) + j] = P.dp_dba.m[3 * i + j];
      Ha[3 * (6 + i) + j] = P.dv_dba.m[3 * i + j];
      Hg[3 * i + j] = P.dR_dbg.m[3 * i + j];
      Hg[3 * (3 + i) + j] = P.dp_dbg.m[3 * i + j];
      Hg[3 * (6 + i) + j] = P.dv_dbg.m[3 * i + j];
    }
  for (int i = 0; i < 9; ++i)
    for (int j = 0; j < 3; ++j) {
      double sa = -Bm[3 * i + j], sg = -Cm[3 * i + j];
      const RowPattern pi = row_pattern(i);
      for (int q = 0; q < pi.n; ++q) {
        sa += A[9 * i + pi.k[q]] * Ha[3 * pi.k[q] + j];
        sg += A[9 * i + pi.k[q]] * Hg[3 * pi.k[q] + j];
      }
      Ha2[3 * i + j] = sa;
      Hg2[3 * i + j] = sg;
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      P.dth_dba.m[3 * i + j] = Ha2[3 * i + j];
      P.dp_dba.m[3 * i + j] = Ha2[3 * (3 + i) + j];
      P.dv_dba.m[3 * i + j] = Ha2[3 * (6 + i) + j];
      P.dR_dbg.m[3 * i + j] = Hg2[3 * i + j];
      P.dp_dbg.m[3 * i + j] = Hg2[3 * (3 + i) + j];
      P.dv_dbg.m[3 * i + j] = Hg2[3 * (6 + i) + j];
    }
  // the mean
  P.dp = P.dp + h * P.dv + h22 * a_nav;
  P.dv = P.dv + h * a_nav;
  P.th = P.th + h * w_tangent;
  P.dR = Exp(P.th);  // deltaRij() for EstimateGravity and the prediction
  P.dt += h;
}

void integrate(Preint& P, V3 acc_meas, V3 gyr_meas, double h, double acc_sigma, double gyr_sigma, double int_sigma) {
  if (P.tangent) return integrate_tangent(P, acc_meas, gyr_meas, h, acc_sigma, gyr_sigma, int_sigma);
  const V3 a = acc_meas - P.ba_lin, w = gyr_meas - P.bg_lin;
  const V3 wh = h * w;
  const M3 dRk = Exp(wh), Jr = RightJacobian(wh);
  const M3 R = P.dR;
  const V3 Ra = R * a;
  const M3 Ra_x = R * skew(a);
  // covariance: x+ = A x + B na + C ng
  double A[81] = {0}, Bm[27] = {0}, Cm[27] = {0};
  const M3 dRkT = transpose(dRk);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      A[9 * i + j] = dRkT.m[3 * i + j];
      A[9 * (3 + i) + j] = -0.5 * h * h * Ra_x.m[3 * i + j];
      A[9 * (6 + i) + j] = -h * Ra_x.m[3 * i + j];
      Bm[3 * (3 + i) + j] = 0.5 * h * h * R.m[3 * i + j];
      Bm[3 * (6 + i) + j] = h * R.m[3 * i + j];
      Cm[3 * i + j] = h * Jr.m[3 * i + j];
    }
  for (int i = 0; i < 3; ++i) {
    A[9 * (3 + i) + 3 + i] = 1.0;
    A[9 * (3 + i) + 6 + i] = h;
    A[9 * (6 + i) + 6 + i] = 1.0;
  }
  const double qa = acc_sigma * acc_sigma / h, qg = gyr_sigma * gyr_sigma / h;
  propagate_covariance(A, Bm, Cm, qa, qg, P.cov);
  for (int i = 0; i < 3; ++i) P.cov[9 * (3 + i) + 3 + i] += int_sigma * int_sigma * h;
  // bias Jacobians (old Delta R)
  const M3 Ra_x_dRdbg = Ra_x * P.dR_dbg;
  P.dp_dba = add(P.dp_dba, add(scaled(P.dv_dba, h), scaled(R, -0.5 * h * h)));
  P.dp_dbg = add(P.dp_dbg, add(scaled(P.dv_dbg, h), scaled(Ra_x_dRdbg, -0.5 * h * h)));
  P.dv_dba = add(P.dv_dba, scaled(R, -h));
  P.dv_dbg = add(P.dv_dbg, scaled(Ra_x_dRdbg, -h));
  P.dR_dbg = add(dRkT * P.dR_dbg, scaled(Jr, -h));
  // deltas
  P.dp = P.dp + h * P.dv + (0.5 * h * h) * Ra;
  P.dv = P.dv + h * Ra;
  P.dR = R * dRk;
  P.dt += h;
}

void corrected(const Preint& P, V3 ba, V3 bg, M3* dR, V3* dp, V3* dv, V3* theta = nullptr) {
  const V3 da = ba - P.ba_lin, dg = bg - P.bg_lin;
  if (P.tangent) {  // biasCorrectedDelta: a linear correction of the tangent vector
    const V3 th = P.th + P.dth_dba * da + P.dR_dbg * dg;
    if (theta != nullptr) *theta = th;
    *dR = Exp(th);
    *dp = P.dp + P.dp_dba * da + P.dp_dbg * dg;
    *dv = P.dv + P.dv_dba * da + P.dv_dbg * dg;
    return;
  }
  *dR = P.dR * Exp(P.dR_dbg * dg);
  *dp = P.dp + P.dp_dba * da + P.dp_dbg * dg;
  *dv = P.dv + P.dv_dba * da + P.dv_dbg * dg;
}

// ---- small dense linear algebra (symmetric positive definite; the window's own solver is the chain solver below)
bool cholesky(std::vector<double>& a, int n) {  // in place, lower
  for (int j = 0; j < n; ++j) {
    double d = a[j * n + j];
    for (int k = 0; k < j; ++k) d -= a[j * n + k] * a[j * n + k];
    if (!(d > 0)) return false;
    d = std::sqrt(d);
    a[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = a[i * n + j];
      for (int k = 0; k < j; ++k) s -= a[i * n + k] * a[j * n + k];
      a[i * n + j] = s / d;
    }
  }
  return true;
}
void chol_solve(const std::vector<double>& l, int n, double* b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= l[i * n + k] * b[k];
    b[i] = s / l[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= l[k * n + i] * b[k];
    b[i] = s / l[i * n + i];
  }
}

// ---- GravityEstimator (gravity_factor/gravity_estimator.cc, in the reference tree: followed line by line) ---------
struct GFrame {   // Rigid3dWithPreintegrator: a pose and the preintegration that was running when it was stored
  M3 R;
  V3 p;
  double dt;      // deltaTij()
  V3 dP, dV;      // deltaPij(), deltaVij()
};
bool solve_sym(const double* A, const double* b, int n, double* x) {  // A.ldlt().solve(b), n <= 3
  double M[3][4];
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) M[i][j] = A[i * n + j];
    M[i][n] = b[i];
  }
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r)
      if (std::fabs(M[r][c]) > std::fabs(M[piv][c])) piv = r;
    if (!(std::fabs(M[piv][c]) > 0.0)) return false;
    for (int j = 0; j <= n; ++j) std::swap(M[c][j], M[piv][j]);
    for (int r = c + 1; r < n; ++r) {
      const double f = M[r][c] / M[c][c];
      for (int j = c; j <= n; ++j) M[r][j] -= f * M[c][j];
    }
  }
  for (int i = n - 1; i >= 0; --i) {
    double v = M[i][n];
    for (int j = i + 1; j < n; ++j) v -= M[i][j] * x[j];
    x[i] = v / M[i][i];
  }
  return true;
}
// ApproximateGravity (:20-97): [R_i^T dt^2/2; R_i^T dt] g = [dP_j + R_i^T R_j tlb - tlb - R_i^T (T_j - T_i) + dt V_i;
// dV_j + V_i - R_i^T R_j V_{i+1}], normal equations over the consecutive pairs, x 1000, solved; |g| within 0.5 of g_norm.
// frame_j's OWN preintegration is paired with the poses of frames i and j, as the reference does.
bool approximate_gravity(const std::vector<GFrame>& f, V3 tlb, const std::vector<V3>& Vs, double g_norm, V3* g) {
  const size_t window = f.size();
  if (window < 3) return false;
  double A[9] = {0}, b[3] = {0};
  for (size_t i = 0; i + 1 < window; ++i) {
    const GFrame &fi = f[i], &fj = f[i + 1];
    const double dt = fj.dt;
    const M3 RiT = transpose(fi.R);
    const M3 RiTRj = RiT * fj.R;
    const M3 A0 = scaled(RiT, dt * dt / 2), A1 = scaled(RiT, dt);
    const V3 b0 = fj.dP + RiTRj * tlb - tlb - RiT * (fj.p - fi.p) + dt * Vs[i];
    const V3 b1 = fj.dV + Vs[i] - RiTRj * Vs[i + 1];
    const double bb0[3] = {b0.x, b0.y, b0.z}, bb1[3] = {b1.x, b1.y, b1.z};
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) {
        double v = 0;
        for (int k = 0; k < 3; ++k) v += A0.m[3 * k + r] * A0.m[3 * k + c] + A1.m[3 * k + r] * A1.m[3 * k + c];
        A[3 * r + c] += v;
      }
      double v = 0;
      for (int k = 0; k < 3; ++k) v += A0.m[3 * k + r] * bb0[k] + A1.m[3 * k + r] * bb1[k];
      b[r] += v;
    }
  }
  for (double& v : A) v *= 1000.0;
  for (double& v : b) v *= 1000.0;
  double x[3];
  if (!solve_sym(A, b, 3, x)) return false;
  *g = {x[0], x[1], x[2]};
  return std::fabs(norm(*g) - g_norm) < 0.5;
}
// TangentBasis (:6-18)
void tangent_basis(V3 g0, V3* b, V3* c) {
  const V3 a = (1.0 / norm(g0)) * g0;
  V3 tmp{0, 0, 1};
  if (a.x == tmp.x && a.y == tmp.y && a.z == tmp.z) tmp = {1, 0, 0};
  V3 bb = tmp - dot(a, tmp) * a;
  *b = (1.0 / norm(bb)) * bb;
  *c = cross(a, *b);
}
// RefineGravity (:99-170): four corrections of g0 in its tangent plane at fixed norm.  A and b are declared outside the
// loop over the four rounds and never cleared in the reference (every round adds to 1000 x the previous sums): kept.
void refine_gravity(const std::vector<GFrame>& f, V3 tlb, const std::vector<V3>& Vs, double g_norm, V3* g_approx) {
  V3 g0 = (g_norm / norm(*g_approx)) * *g_approx;
  const size_t window = f.size();
  double A[4] = {0}, b[2] = {0};
  for (int k = 0; k < 4; ++k) {
    V3 lx, ly;
    tangent_basis(g0, &lx, &ly);
    for (size_t i = 0; i + 1 < window; ++i) {
      const GFrame &fi = f[i], &fj = f[i + 1];
      const double dt = fj.dt;
      const M3 RiT = transpose(fi.R);
      const M3 RiTRj = RiT * fj.R;
      const V3 a0[2] = {(dt * dt / 2) * (RiT * lx), (dt * dt / 2) * (RiT * ly)};  // columns of the 6 x 2 block
      const V3 a1[2] = {dt * (RiT * lx), dt * (RiT * ly)};
      const V3 b0 = fj.dP + RiTRj * tlb - tlb - (dt * dt / 2) * (RiT * g0) - RiT * (fj.p - fi.p) + dt * Vs[i];
      const V3 b1 = fj.dV - dt * (RiT * g0) + Vs[i] - RiTRj * Vs[i + 1];
      for (int r = 0; r < 2; ++r) {
        for (int c = 0; c < 2; ++c) A[2 * r + c] += dot(a0[r], a0[c]) + dot(a1[r], a1[c]);
        b[r] += dot(a0[r], b0) + dot(a1[r], b1);
      }
    }
    for (double& v : A) v *= 1000.0;
    for (double& v : b) v *= 1000.0;
    double dg[2] = {0, 0};
    if (!solve_sym(A, b, 2, dg)) return;
    const V3 gn = g0 + dg[0] * lx + dg[1] * ly;
    g0 = (g_norm / norm(gn)) * gn;
  }
  *g_approx = g0;
}
// Estimate (:172-188)
bool estimate_gravity_vector(const std::vector<GFrame>& f, V3 tlb, const std::vector<V3>& Vs, double g_norm, V3* g) {
  if (!approximate_gravity(f, tlb, Vs, g_norm, g)) return false;
  refine_gravity(f, tlb, Vs, g_norm, g);
  return std::fabs(norm(*g) - g_norm) < 0.2;
}

}  // namespace

struct dliom_imu_window {
  dliom_imu_window_options o;
  std::vector<State> x;         // the window, oldest first
  std::vector<Preint> between;  // between[i]: x[i] -> x[i + 1]
  struct PosePrior {
    int index;  // state in the window
    M3 R;
    V3 p;
    double sigma_rot, sigma_trans;
  };
  std::vector<PosePrior> pose_priors;
  struct Gravity {
    int index;
    V3 nZ, bRef;
    double sigma;
  };
  std::vector<Gravity> gravity;
  // EstimateGravity (:1106-1154): g_est_transforms_ / g_est_Vs_ / g_vec_est_G_
  std::deque<GFrame> g_frames;
  std::deque<V3> g_vs;
  V3 g_est_G{0, 0, 0};
  bool g_est_valid = false;     // the last EstimateGravity() call returned true
  int64_t gravity_factors = 0;  // factors added by add_pose so far
  // Gaussian prior on x[0]: 1/2 d^T H d + b^T d, d = local(lin, x[0])
  double H0[225], b0[15];
  State lin0;
  Preint current;  // since the newest state
  bool initialized = false;
  int64_t num_states = 0;
  int key = 0;  // key_ of the next state: 1 after initialisation, 1 again after a graph reset (:792)
  // ---- the chain solver (round 6).  x[i] is the LINEARISATION POINT of state i (ISAM2's theta_), delta its solved
  // increment (delta_); the estimate is retract(x[i], delta[i]) (calculateEstimate()).  The fixed-lag mode folds delta
  // into x after every solve (plain Gauss-Newton); the reference-rule mode (window_size == 0) moves a linearisation
  // point only when its increment exceeds options.relinearize_threshold, like ISAM2 (:676-679), so the elimination of
  // the older part of the chain is reused from scan to scan.
  struct ChainFactor {  // ImuFactor + bias BetweenFactor between states i and i + 1, linearised at (x[i], x[i + 1])
    double Linv[81];    // whitening of the preintegrated covariance: fixed when the factor is created
    double A[225], B[225], C[225];  // Ja^T Ja, Jb^T Ja (rows: state i + 1, columns: state i), Jb^T Jb
    double ga[15], gb[15];          // Ja^T r, Jb^T r
    bool stale = true;
  };
  struct ChainBlock {   // block Cholesky of the block-tridiagonal normal equations, state i
    double L[225];      // S_i = D_i - W_{i-1} W_{i-1}^T = L L^T
    double W[225];      // B_i L^-T: what state i + 1's rows hold in state i's columns of the factor
    double y[15];       // forward-substituted right-hand side
    double Lt[225];     // L^T (row major) and 1 / diag(L): the back substitution walks every key of the graph every scan,
    double rdiag[15];   // so its inner loops run along rows and multiply instead of dividing
  };
  std::vector<double> delta;       // 15 per state
  std::vector<ChainFactor> fac;    // fac[i]: between[i]
  std::vector<ChainBlock> blk;     // per state
  int clean_until = 0;             // states [0, clean_until) are eliminated with the factors as they are now
  bool full_graph() const { return o.window_size == 0; }
  bool graph_started = false;  // gtsam_initialized_: false between InitializeIMU and the first WindowOptimize (:712-745)
  int64_t relinearizations = 0, blocks_eliminated = 0;  // dliom_imu_window_solver_stats
};

namespace {

constexpr int kD = 15;

// whitened residual of the IMU factor + bias random walk between two states: 15 rows
// The bias-corrected preintegrated measurement at the factor's UNPERTURBED biases: 48 of the 61 residual evaluations of
// a numerical Jacobian (everything but the 6 + 6 bias columns) reuse it instead of recomputing Exp / the corrections.
struct CorrectedCache {
  bool valid = false;
  V3 ba{0, 0, 0}, bg{0, 0, 0};
  M3 dR{};
  V3 dp{0, 0, 0}, dv{0, 0, 0};
};
inline bool same(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

void imu_residual(const dliom_imu_window& w, const Preint& P, const std::vector<double>& Linv, const State& a,
                  const State& b, double* r, CorrectedCache* cache = nullptr) {
  M3 dR;
  V3 dp, dv;
  if (cache != nullptr && cache->valid && same(cache->ba, a.ba) && same(cache->bg, a.bg)) {
    dR = cache->dR;
    dp = cache->dp;
    dv = cache->dv;
  } else {
    corrected(P, a.ba, a.bg, &dR, &dp, &dv);
    if (cache != nullptr && !cache->valid) {  // the first call is the unperturbed one (add_factor evaluates r0 first)
      cache->valid = true;
      cache->ba = a.ba;
      cache->bg = a.bg;
      cache->dR = dR;
      cache->dp = dp;
      cache->dv = dv;
    }
  }
  const V3 g{0, 0, -w.o.gravity};
  const M3 RiT = transpose(a.R);
  V3 rR, rp, rv;
  if (P.tangent) {
    // PreintegrationBase::computeError (GTSAM 4.0.2): state_j.localCoordinates(predict(state_i, bias_i))
    const M3 RjT = transpose(b.R);
    rR = Log(RjT * (a.R * dR));
    rp = RjT * ((a.p + P.dt * a.v + (0.5 * P.dt * P.dt) * g + a.R * dp) - b.p);
    rv = RjT * ((a.v + P.dt * g + a.R * dv) - b.v);
  } else {
    rR = Log(transpose(dR) * (RiT * b.R));
    rp = RiT * (b.p - a.p - P.dt * a.v - (0.5 * P.dt * P.dt) * g) - dp;
    rv = RiT * (b.v - a.v - P.dt * g) - dv;
  }
  const double raw[9] = {rR.x, rR.y, rR.z, rp.x, rp.y, rp.z, rv.x, rv.y, rv.z};
  for (int i = 0; i < 9; ++i) {  // r = L^-1 raw (L lower Cholesky factor of the covariance)
    double s = 0;
    for (int k = 0; k <= i; ++k) s += Linv[9 * i + k] * raw[k];
    r[i] = s;
  }
  const double sq = std::sqrt(std::max(P.dt, 1e-12));
  const V3 da = b.ba - a.ba, dg = b.bg - a.bg;
  const double sa = sq * w.o.acc_bias_noise, sg = sq * w.o.gyr_bias_noise;
  r[9] = da.x / sa;
  r[10] = da.y / sa;
  r[11] = da.z / sa;
  r[12] = dg.x / sg;
  r[13] = dg.y / sg;
  r[14] = dg.z / sg;
}

void pose_prior_residual(const dliom_imu_window::PosePrior& f, const State& s, double* r) {
  const V3 w = Log(transpose(f.R) * s.R), t = transpose(f.R) * (s.p - f.p);
  r[0] = w.x / f.sigma_rot;
  r[1] = w.y / f.sigma_rot;
  r[2] = w.z / f.sigma_rot;
  r[3] = t.x / f.sigma_trans;
  r[4] = t.y / f.sigma_trans;
  r[5] = t.z / f.sigma_trans;
}

// gtsam::Unit3::basis() (GTSAM 4.0.2, geometry/Unit3.cpp): the coordinate axis with the smallest |component| of n
// (x before y before z on ties), b1 = normalize(n x axis), b2 = n x b1.
void unit3_basis(V3 n, V3* b1, V3* b2) {
  const double mx = std::fabs(n.x), my = std::fabs(n.y), mz = std::fabs(n.z);
  V3 axis{0, 0, 1};
  if (mx <= my && mx <= mz)
    axis = {1, 0, 0};
  else if (my <= mx && my <= mz)
    axis = {0, 1, 0};
  V3 B1 = cross(n, axis);
  *b1 = (1.0 / norm(B1)) * B1;
  *b2 = cross(n, *b1);
}

// Pose3GravityFactor::evaluateError (gravity_factor.cc:10-33, .h:190-199): r (2, whitened) and, if J != nullptr, the
// factor's 2 x 15 Jacobian (only the rotation block is non-zero).
void gravity_residual(const dliom_imu_window::Gravity& f, const State& s, double* r, double* J = nullptr) {
  // Rot3::xyz(): roll, pitch of R = Rz Ry Rx
  const double roll = std::atan2(s.R.m[7], s.R.m[8]);
  const double pitch = std::atan2(-s.R.m[6], std::sqrt(s.R.m[7] * s.R.m[7] + s.R.m[8] * s.R.m[8]));
  const double cr = std::cos(roll), sr = std::sin(roll), cp = std::cos(pitch), sp = std::sin(pitch);
  const M3 Rrp{{cp, sp * sr, sp * cr, 0, cr, -sr, -sp, cp * sr, cp * cr}};  // Rot3::RzRyRx(roll, pitch, 0) = Ry(pitch) Rx(roll)
  const V3 nRef = Rrp * f.bRef;
  V3 p1, p2;
  unit3_basis(f.nZ, &p1, &p2);
  const double k_cost_value = 1e-5;
  r[0] = (dot(p1, nRef) + k_cost_value) / f.sigma;  // nZ_.error(nRef) = B_nZ^T nRef, + 1e-5 per component
  r[1] = (dot(p2, nRef) + k_cost_value) / f.sigma;
  if (J != nullptr) {
    // D_nRef_R = -B_q^T R_rp [bRef]x (Rot3::rotate(Unit3)), D_e_nRef = B_nZ^T B_q (Unit3::error): H = D_e_nRef D_nRef_R
    V3 q1, q2;
    unit3_basis(nRef, &q1, &q2);
    const M3 RS = Rrp * skew(f.bRef);
    double D[2][3];
    const V3 rows[2] = {q1, q2};
    for (int i = 0; i < 2; ++i) {
      D[i][0] = -(rows[i].x * RS.m[0] + rows[i].y * RS.m[3] + rows[i].z * RS.m[6]);
      D[i][1] = -(rows[i].x * RS.m[1] + rows[i].y * RS.m[4] + rows[i].z * RS.m[7]);
      D[i][2] = -(rows[i].x * RS.m[2] + rows[i].y * RS.m[5] + rows[i].z * RS.m[8]);
    }
    const double E[2][2] = {{dot(p1, q1), dot(p1, q2)}, {dot(p2, q1), dot(p2, q2)}};
    for (int i = 0; i < 2 * kD; ++i) J[i] = 0.0;
    for (int i = 0; i < 2; ++i)
      for (int c = 0; c < 3; ++c) J[i * kD + c] = (E[i][0] * D[0][c] + E[i][1] * D[1][c]) / f.sigma;
  }
}

// Lower-triangular inverse of the Cholesky factor of the 9 x 9 preintegrated covariance.
bool whitening(const Preint& P, std::vector<double>* Linv) {
  std::vector<double> L(P.cov, P.cov + 81);
  for (int i = 0; i < 9; ++i) L[10 * i] += 1e-18;
  if (!cholesky(L, 9)) return false;
  Linv->assign(81, 0.0);
  for (int c = 0; c < 9; ++c) {  // solve L X = I column by column
    for (int i = c; i < 9; ++i) {
      double s = i == c ? 1.0 : 0.0;
      for (int k = c; k < i; ++k) s -= L[9 * i + k] * (*Linv)[9 * k + c];
      (*Linv)[9 * i + c] = s / L[9 * i + i];
    }
  }
  return true;
}

// The same accumulation for a factor on ONE state that supplies its own Jacobian (rows x 15).
void add_factor_with_jacobian(int ia, int rows, const double* r0, const double* J, std::vector<double>& H, std::vector<double>& g,
                              int n) {
  for (int c1 = 0; c1 < kD; ++c1) {
    const int g1 = ia * kD + c1;
    double s = 0;
    for (int i = 0; i < rows; ++i) s += J[i * kD + c1] * r0[i];
    g[g1] += s;
    for (int c2 = c1; c2 < kD; ++c2) {
      const int g2 = ia * kD + c2;
      double h = 0;
      for (int i = 0; i < rows; ++i) h += J[i * kD + c1] * J[i * kD + c2];
      H[static_cast<size_t>(g1) * n + g2] += h;
      if (c2 != c1) H[static_cast<size_t>(g2) * n + g1] += h;
    }
  }
}

// ---- analytic Jacobians (round 3) -----------------------------------------------------------------------------------
// The numerical Jacobians above cost 61 residual evaluations per IMU factor and 13 per pose prior, two Gauss-Newton
// iterations a scan: 0.14 ms of the 1 ms the reference's whole per-scan chain takes on the device.  The factors'
// derivatives are closed forms on this parametrisation (R <- R Exp(d), p <- p + d, ...):
//   rR = Log(dR~^T Ra^T Rb)          d/dtheta_a = -Jr^-1(rR) Rb^T Ra     d/dtheta_b = Jr^-1(rR)
//                                    d/dbg_a    = -Jr^-1(rR) Exp(rR)^T Jr(dR_dbg dbg) dR_dbg
//   rp = Ra^T (pb - pa - dt va - g dt^2 / 2) - dp~    d/dtheta_a = [Ra^T (...)]x, d/dpa = -Ra^T, d/dva = -dt Ra^T,
//                                                     d/dba = -dp_dba, d/dbg = -dp_dbg, d/dpb = Ra^T
//   rv = Ra^T (vb - va - g dt) - dv~                  d/dtheta_a = [Ra^T (...)]x, d/dva = -Ra^T, d/dba = -dv_dba,
//                                                     d/dbg = -dv_dbg, d/dvb = Ra^T
// (Forster et al., the derivation behind gtsam::ImuFactor); rows 0-8 are whitened by L^-1 like the residual.
// dliom_diag_imu_factor_jacobians returns both versions of a factor's Jacobian; the test compares them.
void imu_factor_jacobian(const dliom_imu_window& w, const Preint& P, const std::vector<double>& Linv, const State& a, const State& b,
                         double* r, double* J /* 15 x 30, row major */) {
  imu_residual(w, P, Linv, a, b, r, nullptr);
  M3 dR;
  V3 dp, dv, theta{0, 0, 0};
  corrected(P, a.ba, a.bg, &dR, &dp, &dv, &theta);
  const V3 g{0, 0, -w.o.gravity};
  const M3 RaT = transpose(a.R);
  double raw[9][30];
  for (auto& row : raw)
    for (double& v : row) v = 0.0;
  auto put = [&](int r0, int c0, const M3& m, double scale) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) raw[r0 + i][c0 + j] = scale * m.m[3 * i + j];
  };
  if (P.tangent) {
    // e_R = Log(Rb^T Ra Exp(theta)),  e_p = Rb^T (p* - pb),  e_v = Rb^T (v* - vb)   (imu_residual); on this
    // parametrisation (R <- R Exp(d), p <- p + d, v <- v + d, biases additive):
    //   e_R: d/dtheta_a = Jr^-1(e_R) Exp(theta)^T,  d/dtheta_b = -Jr^-1(e_R) E^T (E = Rb^T Ra Exp(theta)),
    //        d/db = Jr^-1(e_R) Jr(theta) dtheta/db
    //   e_p: d/dtheta_b = [e_p]x, d/dpb = -Rb^T, d/dpa = Rb^T, d/dva = dt Rb^T, d/dtheta_a = -Rb^T Ra [P]x, d/db = Rb^T Ra dP/db
    //   e_v: d/dtheta_b = [e_v]x, d/dvb = -Rb^T, d/dva = Rb^T, d/dtheta_a = -Rb^T Ra [V]x, d/db = Rb^T Ra dV/db
    const M3 RbT = transpose(b.R);
    const M3 RbTRa = RbT * a.R;
    const M3 E = RbTRa * dR;
    const V3 eR = Log(E);
    const M3 Jri = RightJacobianInverse(eR);
    const V3 ep = RbT * ((a.p + P.dt * a.v + (0.5 * P.dt * P.dt) * g + a.R * dp) - b.p);
    const V3 ev = RbT * ((a.v + P.dt * g + a.R * dv) - b.v);
    const M3 JrTheta = RightJacobian(theta);
    put(0, 0, Jri * transpose(dR), 1.0);
    put(0, 15, Jri * transpose(E), -1.0);
    put(0, 9, Jri * (JrTheta * P.dth_dba), 1.0);
    put(0, 12, Jri * (JrTheta * P.dR_dbg), 1.0);
    put(3, 15, skew(ep), 1.0);
    put(3, 18, RbT, -1.0);
    put(3, 3, RbT, 1.0);
    put(3, 6, RbT, P.dt);
    put(3, 0, RbTRa * skew(dp), -1.0);
    put(3, 9, RbTRa * P.dp_dba, 1.0);
    put(3, 12, RbTRa * P.dp_dbg, 1.0);
    put(6, 15, skew(ev), 1.0);
    put(6, 21, RbT, -1.0);
    put(6, 6, RbT, 1.0);
    put(6, 0, RbTRa * skew(dv), -1.0);
    put(6, 9, RbTRa * P.dv_dba, 1.0);
    put(6, 12, RbTRa * P.dv_dbg, 1.0);
  } else {
  const M3 E = transpose(dR) * (RaT * b.R);
  const V3 rR = Log(E);
  const M3 Jri = RightJacobianInverse(rR);
  const V3 xp = RaT * (b.p - a.p - P.dt * a.v - (0.5 * P.dt * P.dt) * g);
  const V3 xv = RaT * (b.v - a.v - P.dt * g);
  // rR
  put(0, 0, Jri * (transpose(b.R) * a.R), -1.0);
  put(0, 15, Jri, 1.0);
  {
    const V3 dg = a.bg - P.bg_lin;
    put(0, 12, Jri * (transpose(E) * (RightJacobian(P.dR_dbg * dg) * P.dR_dbg)), -1.0);
  }
  // rp
  put(3, 0, skew(xp), 1.0);
  put(3, 3, RaT, -1.0);
  put(3, 6, RaT, -P.dt);
  put(3, 9, P.dp_dba, -1.0);
  put(3, 12, P.dp_dbg, -1.0);
  put(3, 18, RaT, 1.0);
  // rv
  put(6, 0, skew(xv), 1.0);
  put(6, 6, RaT, -1.0);
  put(6, 9, P.dv_dba, -1.0);
  put(6, 12, P.dv_dbg, -1.0);
  put(6, 21, RaT, 1.0);
  }
  for (int i = 0; i < 9; ++i)
    for (int c = 0; c < 30; ++c) {
      double s = 0;
      for (int k = 0; k <= i; ++k) s += Linv[9 * i + k] * raw[k][c];
      J[30 * i + c] = s;
    }
  for (int i = 9; i < 15; ++i)
    for (int c = 0; c < 30; ++c) J[30 * i + c] = 0.0;
  const double sq = std::sqrt(std::max(P.dt, 1e-12));
  const double sa = sq * w.o.acc_bias_noise, sg = sq * w.o.gyr_bias_noise;
  for (int i = 0; i < 3; ++i) {
    J[30 * (9 + i) + 9 + i] = -1.0 / sa;
    J[30 * (9 + i) + 24 + i] = 1.0 / sa;
    J[30 * (12 + i) + 12 + i] = -1.0 / sg;
    J[30 * (12 + i) + 27 + i] = 1.0 / sg;
  }
}

void pose_prior_jacobian(const dliom_imu_window::PosePrior& f, const State& s, double* r, double* J /* 6 x 15 */) {
  pose_prior_residual(f, s, r);
  const M3 RmT = transpose(f.R);
  const M3 Jri = RightJacobianInverse(Log(RmT * s.R));
  for (int i = 0; i < 6 * kD; ++i) J[i] = 0.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      J[kD * i + j] = Jri.m[3 * i + j] / f.sigma_rot;
      J[kD * (3 + i) + 3 + j] = RmT.m[3 * i + j] / f.sigma_trans;
    }
}

// J^T J and J^T r of the two factor shapes that make up almost all of a window, without the structural zeros: the IMU
// factor's nine whitened rows do not touch the second state's biases and its six bias rows have two entries each; the
// pose prior is two 3 x 3 blocks.  Same products, same order within every sum that is not a structural zero.
void accumulate_imu_factor(int ia, const double* r0, const double* J /* 15 x 30 */, std::vector<double>& H, std::vector<double>& g,
                           int n) {
  const int base_a = ia * kD, base_b = (ia + 1) * kD;
  auto gi = [&](int c) { return c < kD ? base_a + c : base_b + (c - kD); };
  for (int c1 = 0; c1 < 24; ++c1) {  // columns 24 .. 29 (ba_b, bg_b) are zero in rows 0 .. 8
    const int g1 = gi(c1);
    double s = 0;
    for (int i = 0; i < 9; ++i) s += J[30 * i + c1] * r0[i];
    g[g1] += s;
    for (int c2 = c1; c2 < 24; ++c2) {
      const int g2 = gi(c2);
      double h = 0;
      for (int i = 0; i < 9; ++i) h += J[30 * i + c1] * J[30 * i + c2];
      H[static_cast<size_t>(g1) * n + g2] += h;
      if (c2 != c1) H[static_cast<size_t>(g2) * n + g1] += h;
    }
  }
  for (int i = 9; i < 15; ++i) {  // bias random walk: row i has -1/s at column i (state a) and +1/s at column 15 + i (state b)
    const int ca = i, cb = kD + i;
    const double ja = J[30 * i + ca], jb = J[30 * i + cb];
    const int ga = gi(ca), gb = gi(cb);
    g[ga] += ja * r0[i];
    g[gb] += jb * r0[i];
    H[static_cast<size_t>(ga) * n + ga] += ja * ja;
    H[static_cast<size_t>(gb) * n + gb] += jb * jb;
    H[static_cast<size_t>(ga) * n + gb] += ja * jb;
    H[static_cast<size_t>(gb) * n + ga] += ja * jb;
  }
}
void accumulate_pose_prior(int ia, const double* r0, const double* J /* 6 x 15 */, std::vector<double>& H, std::vector<double>& g, int n) {
  const int base = ia * kD;
  for (int blk = 0; blk < 2; ++blk)
    for (int c1 = 0; c1 < 3; ++c1) {
      const int g1 = base + 3 * blk + c1;
      double s = 0;
      for (int i = 0; i < 3; ++i) s += J[kD * (3 * blk + i) + 3 * blk + c1] * r0[3 * blk + i];
      g[g1] += s;
      for (int c2 = c1; c2 < 3; ++c2) {
        const int g2 = base + 3 * blk + c2;
        double h = 0;
        for (int i = 0; i < 3; ++i) h += J[kD * (3 * blk + i) + 3 * blk + c1] * J[kD * (3 * blk + i) + 3 * blk + c2];
        H[static_cast<size_t>(g1) * n + g2] += h;
        if (c2 != c1) H[static_cast<size_t>(g2) * n + g1] += h;
      }
    }
}

// The Gaussian prior on x[0] (the initial priors :712-745, the reset's :756-770, or the fixed-lag window's marginal):
// 1/2 d^T H0 d + b0^T d with d = local(lin0, x[0]).  Its derivative with respect to an increment of x[0] goes through
// d(local)/d(increment) = diag(Jr^-1(d_rot), I): Log(R0^T R Exp(e)) = d_rot + Jr^-1(d_rot) e -- what
// gtsam::PriorFactor<Pose3>::evaluateError returns as H -- so a prior whose state has been pulled away from its mean is
// linearised like the reference's (round 6; the plain H0 / H0 d used until then is the d_rot -> 0 limit).
// Adds J^T H0 J to H (row stride n) and J^T (b0 + H0 d) to g, both at offset 0.
void add_state0_prior(const dliom_imu_window& w, const State& x0, double* H, double* g, int n) {
  double d0[kD], t[kD];
  local(w.lin0, x0, d0);
  const M3 Ji = RightJacobianInverse({d0[0], d0[1], d0[2]});
  for (int i = 0; i < kD; ++i) {
    double acc = w.b0[i];
    for (int j = 0; j < kD; ++j) acc += w.H0[kD * i + j] * d0[j];
    t[i] = acc;
  }
  double HJ[kD * kD];  // H0 J: columns 0..2 mixed by Jr^-1
  for (int i = 0; i < kD; ++i) {
    for (int c = 0; c < 3; ++c)
      HJ[kD * i + c] = w.H0[kD * i + 0] * Ji.m[0 * 3 + c] + w.H0[kD * i + 1] * Ji.m[1 * 3 + c] + w.H0[kD * i + 2] * Ji.m[2 * 3 + c];
    for (int c = 3; c < kD; ++c) HJ[kD * i + c] = w.H0[kD * i + c];
  }
  for (int c = 0; c < kD; ++c) {  // J^T (H0 J): rows 0..2 mixed the same way
    for (int r = 0; r < 3; ++r)
      H[static_cast<size_t>(r) * n + c] += Ji.m[0 * 3 + r] * HJ[kD * 0 + c] + Ji.m[1 * 3 + r] * HJ[kD * 1 + c] + Ji.m[2 * 3 + r] * HJ[kD * 2 + c];
    for (int r = 3; r < kD; ++r) H[static_cast<size_t>(r) * n + c] += HJ[kD * r + c];
  }
  for (int r = 0; r < 3; ++r) g[r] += Ji.m[0 * 3 + r] * t[0] + Ji.m[1 * 3 + r] * t[1] + Ji.m[2 * 3 + r] * t[2];
  for (int r = 3; r < kD; ++r) g[r] += t[r];
}

// ---- the chain solver -------------------------------------------------------------------------------------------------
// The window's normal equations are block tridiagonal (every factor touches one state or two neighbours; the prior sits on
// the first block): block Cholesky from the oldest state on, S_i = D_i - W_{i-1} W_{i-1}^T = L_i L_i^T, W_i = B_i L_i^-T,
// y_i = L_i^-1 (-g_i - W_{i-1} y_{i-1}), then delta from the newest state back.  What a state's elimination needs from
// the older part of the chain is (W_{i-1}, y_{i-1}) only, so a new key, a factor on a recent state or a moved
// linearisation point re-eliminates the chain from there on and nothing before it: linear in the window for a full
// Gauss-Newton step, a handful of blocks per scan for the reference's rule.
State estimate_at(const dliom_imu_window& w, size_t i) { return retract(w.x[i], &w.delta[kD * i]); }

bool chol15(double* a) {  // in place, lower, row major 15 x 15
  for (int j = 0; j < kD; ++j) {
    double d = a[kD * j + j];
    for (int k = 0; k < j; ++k) d -= a[kD * j + k] * a[kD * j + k];
    if (!(d > 0)) return false;
    d = std::sqrt(d);
    a[kD * j + j] = d;
    for (int i = j + 1; i < kD; ++i) {
      double t = a[kD * i + j];
      for (int k = 0; k < j; ++k) t -= a[kD * i + k] * a[kD * j + k];
      a[kD * i + j] = t / d;
    }
  }
  return true;
}
inline void lower_solve15(const double* L, double* b) {  // b <- L^-1 b
  for (int i = 0; i < kD; ++i) {
    double t = b[i];
    for (int k = 0; k < i; ++k) t -= L[kD * i + k] * b[k];
    b[i] = t / L[kD * i + i];
  }
}
inline void upper_solve15(const double* L, double* b) {  // b <- L^-T b
  for (int i = kD - 1; i >= 0; --i) {
    double t = b[i];
    for (int k = i + 1; k < kD; ++k) t -= L[kD * k + i] * b[k];
    b[i] = t / L[kD * i + i];
  }
}

void chain_mark(dliom_imu_window& w, int state) {  // something on `state` changed: its block and everything after it
  w.clean_until = std::min(w.clean_until, std::max(0, state));
}

// A new key: its increment, its factor to the state in front of it (whitening computed once), its block.
bool chain_push_state(dliom_imu_window& w) {
  const int N = static_cast<int>(w.x.size());
  w.delta.resize(static_cast<size_t>(N) * kD, 0.0);
  w.blk.resize(static_cast<size_t>(N));
  if (N >= 2 && static_cast<int>(w.fac.size()) < N - 1) {
    std::vector<double> Linv;
    if (!whitening(w.between[static_cast<size_t>(N) - 2], &Linv)) return false;
    dliom_imu_window::ChainFactor f;
    std::memcpy(f.Linv, Linv.data(), sizeof f.Linv);
    f.stale = true;
    w.fac.push_back(f);
    chain_mark(w, N - 2);  // the older state's diagonal block gains Ja^T Ja
  }
  chain_mark(w, N - 1);
  return true;
}

// theta <- theta (+) delta for the KEYS whose increment exceeds the threshold -- ISAM2's relinearisation, which looks at
// every variable by itself: X (rotation, translation), V and B of a state are three keys there (X(i), V(i), B(i),
// local_trajectory_builder_3d.cc:703-707), each moved when ITS increment's largest component reaches the threshold
// (threshold 0: every key that moved at all = plain Gauss-Newton).  The retraction acts on the components separately, so
// a state may hold a moved pose next to an unmoved velocity.  A moved key makes both of its state's IMU factors stale.
void chain_relinearize(dliom_imu_window& w, double threshold) {
  const int N = static_cast<int>(w.x.size());
  const int first[3] = {0, 6, 9}, size[3] = {6, 3, 6};  // X, V, B
  for (int i = 0; i < N; ++i) {
    double* d = &w.delta[static_cast<size_t>(kD) * i];
    double part[kD] = {0};
    bool any = false;
    for (int k = 0; k < 3; ++k) {
      double m = 0.0;
      for (int c = first[k]; c < first[k] + size[k]; ++c) m = std::max(m, std::fabs(d[c]));
      if (!(m > threshold)) continue;
      for (int c = first[k]; c < first[k] + size[k]; ++c) {
        part[c] = d[c];
        d[c] = 0.0;
      }
      any = true;
      ++w.relinearizations;
    }
    if (!any) continue;
    w.x[i] = retract(w.x[i], part);
    if (i > 0) w.fac[static_cast<size_t>(i) - 1].stale = true;
    if (i + 1 < N) w.fac[static_cast<size_t>(i)].stale = true;
    chain_mark(w, i - 1);
  }
}

// Forward elimination of the states [clean_until, N).
bool chain_eliminate(dliom_imu_window& w) {
  const int N = static_cast<int>(w.x.size());
  std::vector<double> H(30 * 30), g(30), D(kD * kD), gd(kD);
  for (int i = w.clean_until; i < N; ++i) {
    // the factor in front of state i is fresh (state i - 1 was eliminated with it, or chain_relinearize / chain_push_state
    // would have marked i - 1); the factor behind it may be stale
    if (i + 1 < N && w.fac[i].stale) {
      dliom_imu_window::ChainFactor& f = w.fac[i];
      const std::vector<double> Linv(f.Linv, f.Linv + 81);
      double r[15], J[15 * 30];
      imu_factor_jacobian(w, w.between[i], Linv, w.x[i], w.x[i + 1], r, J);
      std::fill(H.begin(), H.end(), 0.0);
      std::fill(g.begin(), g.end(), 0.0);
      accumulate_imu_factor(0, r, J, H, g, 30);
      for (int a = 0; a < kD; ++a) {
        for (int b = 0; b < kD; ++b) {
          f.A[kD * a + b] = H[30 * a + b];
          f.B[kD * a + b] = H[30 * (kD + a) + b];
          f.C[kD * a + b] = H[30 * (kD + a) + kD + b];
        }
        f.ga[a] = g[a];
        f.gb[a] = g[kD + a];
      }
      f.stale = false;
    }
    // D_i, g_i: the factors on state i alone, the prior (state 0), the two IMU factors' shares
    std::fill(D.begin(), D.end(), 0.0);
    std::fill(gd.begin(), gd.end(), 0.0);
    if (i == 0) add_state0_prior(w, w.x[0], D.data(), gd.data(), kD);
    for (const auto& f : w.pose_priors)
      if (f.index == i) {
        double r[6], J[6 * kD];
        pose_prior_jacobian(f, w.x[i], r, J);
        accumulate_pose_prior(0, r, J, D, gd, kD);
      }
    for (const auto& f : w.gravity)
      if (f.index == i) {
        double r[2], J[2 * kD];
        gravity_residual(f, w.x[i], r, J);
        add_factor_with_jacobian(0, 2, r, J, D, gd, kD);
      }
    if (i > 0) {
      const dliom_imu_window::ChainFactor& f = w.fac[static_cast<size_t>(i) - 1];
      for (int a = 0; a < kD * kD; ++a) D[a] += f.C[a];
      for (int a = 0; a < kD; ++a) gd[a] += f.gb[a];
    }
    if (i + 1 < N) {
      const dliom_imu_window::ChainFactor& f = w.fac[i];
      for (int a = 0; a < kD * kD; ++a) D[a] += f.A[a];
      for (int a = 0; a < kD; ++a) gd[a] += f.ga[a];
    }
    dliom_imu_window::ChainBlock& b = w.blk[i];
    double rhs[kD];
    for (int a = 0; a < kD; ++a) rhs[a] = -gd[a];
    if (i > 0) {
      const dliom_imu_window::ChainBlock& p = w.blk[static_cast<size_t>(i) - 1];
      for (int a = 0; a < kD; ++a) {
        double t = 0.0;
        for (int k = 0; k < kD; ++k) t += p.W[kD * a + k] * p.y[k];
        rhs[a] -= t;
        for (int c = 0; c <= a; ++c) {
          double u = 0.0;
          for (int k = 0; k < kD; ++k) u += p.W[kD * a + k] * p.W[kD * c + k];
          D[kD * a + c] -= u;
        }
      }
    }
    for (int a = 0; a < kD; ++a) D[kD * a + a] += 1e-12;
    std::memcpy(b.L, D.data(), sizeof b.L);  // the lower triangle is what chol15 reads
    if (!chol15(b.L)) return false;
    lower_solve15(b.L, rhs);
    std::memcpy(b.y, rhs, sizeof b.y);
    for (int a = 0; a < kD; ++a) {
      b.rdiag[a] = 1.0 / b.L[kD * a + a];
      for (int c = 0; c < kD; ++c) b.Lt[kD * a + c] = b.L[kD * c + a];
    }
    if (i + 1 < N) {  // W L^T = B: every row of B by forward substitution
      const double* B = w.fac[i].B;
      for (int a = 0; a < kD; ++a) {
        double row[kD];
        std::memcpy(row, B + kD * a, sizeof row);
        lower_solve15(b.L, row);
        std::memcpy(b.W + kD * a, row, sizeof row);
      }
    }
    ++w.blocks_eliminated;
  }
  w.clean_until = N;
  return true;
}

// Increments from the newest key back.  `first_changed`: the oldest state whose elimination was redone for this solve;
// in front of it (L, W, y) are what the previous solve used, so once a key's increment comes out as it was -- to 1e-13 of
// its tangent units, ten orders below ISAM2's own wildfire threshold of 1e-3 -- the older keys' increments stand and the
// walk stops: a scan costs the keys its information still reaches, not the graph.
bool chain_back_substitute(dliom_imu_window& w, int first_changed) {  // false: an increment is not finite (a NaN residual)
  const int N = static_cast<int>(w.x.size());
  bool finite = true;
  for (int i = N - 1; i >= 0; --i) {
    const dliom_imu_window::ChainBlock& b = w.blk[i];
    double t[kD];
    std::memcpy(t, b.y, sizeof t);
    if (i + 1 < N) {
      const double* dn = &w.delta[static_cast<size_t>(kD) * (i + 1)];
      for (int a = 0; a < kD; ++a) {
        const double da = dn[a];
        const double* row = b.W + kD * a;
        for (int c = 0; c < kD; ++c) t[c] -= row[c] * da;
      }
    }
    for (int r = kD - 1; r >= 0; --r) {  // L^T x = t along the rows of L^T
      double acc = t[r];
      const double* row = b.Lt + kD * r;
      for (int k = r + 1; k < kD; ++k) acc -= row[k] * t[k];
      t[r] = acc * b.rdiag[r];
    }
    double* d = &w.delta[static_cast<size_t>(kD) * i];
    double change = 0.0;
    for (int c = 0; c < kD; ++c) {
      finite = finite && std::isfinite(t[c]);
      change = std::max(change, std::fabs(t[c] - d[c]));
    }
    std::memcpy(d, t, sizeof t);
    if (!finite) return false;
    if (i < first_changed && change <= 1e-13) break;
  }
  return finite;
}

// `iterations` x ISAM2::update(): move the linearisation points that have to move, re-eliminate what changed, solve.
// Fixed-lag mode: every point moves every time (Gauss-Newton) and the last increment is folded into the states.
bool gauss_newton(dliom_imu_window& w, int iterations) {
  const double threshold = w.full_graph() ? w.o.relinearize_threshold : 0.0;
  for (int it = 0; it < iterations; ++it) {
    chain_relinearize(w, threshold);
    if (w.clean_until >= static_cast<int>(w.x.size())) continue;  // nothing changed: the increments stand
    const int first_changed = w.clean_until;
    if (!chain_eliminate(w) || !chain_back_substitute(w, first_changed)) return false;
  }
  if (!w.full_graph()) chain_relinearize(w, 0.0);
  return true;
}

// Marginalises x[0]: the factors touching it (its prior, the IMU factor to x[1], pose priors / gravity factors on
// it) become a Gaussian prior on x[1], linearised at the current estimate.
#ifdef DLIOM_TEST_HOOKS
// tests/cpp/imu_window_marginalize_fail.cc compiles this file by itself with the hook; libdliom.so never has it
int dliom_test_fail_marginalize = 0;
#endif

bool marginalize_oldest(dliom_imu_window& w) {
#ifdef DLIOM_TEST_HOOKS
  if (dliom_test_fail_marginalize != 0) return false;
#endif
  // sub-problem of x[0], x[1] with only those factors
  std::vector<State> x2 = {w.x[0], w.x[1]};
  const int n = 2 * kD;
  std::vector<double> H(static_cast<size_t>(n) * n, 0.0), g(n, 0.0);
  add_state0_prior(w, w.x[0], H.data(), g.data(), n);
  std::vector<double> Linv;
  if (!whitening(w.between[0], &Linv)) return false;
  const Preint P = w.between[0];
  {
    double r[15], J[15 * 30];
    imu_factor_jacobian(w, P, Linv, x2[0], x2[1], r, J);
    accumulate_imu_factor(0, r, J, H, g, n);
  }
  for (const auto& f : w.pose_priors)
    if (f.index == 0) {
      double r[6], J[6 * kD];
      pose_prior_jacobian(f, x2[0], r, J);
      accumulate_pose_prior(0, r, J, H, g, n);
    }
  for (const auto& f : w.gravity)
    if (f.index == 0) {
      double r[2], J[2 * kD];
      gravity_residual(f, x2[0], r, J);
      add_factor_with_jacobian(0, 2, r, J, H, g, n);
    }
  // Schur complement of the first block
  std::vector<double> Haa(kD * kD);
  for (int i = 0; i < kD; ++i)
    for (int j = 0; j < kD; ++j) Haa[kD * i + j] = H[static_cast<size_t>(i) * n + j];
  for (int i = 0; i < kD; ++i) Haa[kD * i + i] += 1e-12;
  if (!cholesky(Haa, kD)) return false;
  // X = Haa^-1 [Hab | ga]
  double X[kD][kD + 1];
  for (int c = 0; c <= kD; ++c) {
    double col[kD];
    for (int i = 0; i < kD; ++i) col[i] = c < kD ? H[static_cast<size_t>(i) * n + kD + c] : g[i];
    chol_solve(Haa, kD, col);
    for (int i = 0; i < kD; ++i) X[i][c] = col[i];
  }
  for (int i = 0; i < kD; ++i) {
    double s = g[kD + i];
    for (int k = 0; k < kD; ++k) s -= H[static_cast<size_t>(kD + i) * n + k] * X[k][kD];
    w.b0[i] = s;
    for (int j = 0; j < kD; ++j) {
      double h = H[static_cast<size_t>(kD + i) * n + kD + j];
      for (int k = 0; k < kD; ++k) h -= H[static_cast<size_t>(kD + i) * n + k] * X[k][j];
      w.H0[kD * i + j] = h;
    }
  }
  for (int i = 0; i < kD; ++i)  // symmetrise
    for (int j = i + 1; j < kD; ++j) w.H0[kD * i + j] = w.H0[kD * j + i] = 0.5 * (w.H0[kD * i + j] + w.H0[kD * j + i]);
  w.lin0 = w.x[1];
  w.x.erase(w.x.begin());
  w.between.erase(w.between.begin());
  w.delta.erase(w.delta.begin(), w.delta.begin() + kD);  // folded into x by gauss_newton (fixed-lag mode): zeros
  w.fac.erase(w.fac.begin());
  w.blk.erase(w.blk.begin());
  w.clean_until = 0;  // the new first state carries the marginal prior now
  std::vector<dliom_imu_window::PosePrior> pp;
  for (auto f : w.pose_priors)
    if (f.index > 0) {
      --f.index;
      pp.push_back(f);
    }
  w.pose_priors.swap(pp);
  std::vector<dliom_imu_window::Gravity> gg;
  for (auto f : w.gravity)
    if (f.index > 0) {
      --f.index;
      gg.push_back(f);
    }
  w.gravity.swap(gg);
  return true;
}

State predicted(const dliom_imu_window& w, const State& s, const Preint& P) {
  M3 dR;
  V3 dp, dv;
  corrected(P, s.ba, s.bg, &dR, &dp, &dv);
  const V3 g{0, 0, -w.o.gravity};
  State r = s;
  r.R = s.R * dR;
  r.p = s.p + P.dt * s.v + (0.5 * P.dt * P.dt) * g + s.R * dp;
  r.v = s.v + P.dt * g + s.R * dv;
  return r;
}

void write_state(const State& s, double pose7[7], double vel3[3], double bias6[6]) {
  if (pose7 != nullptr) {
    pose7[0] = s.p.x;
    pose7[1] = s.p.y;
    pose7[2] = s.p.z;
    matrix_to_quat(s.R, pose7 + 3);
  }
  if (vel3 != nullptr) {
    vel3[0] = s.v.x;
    vel3[1] = s.v.y;
    vel3[2] = s.v.z;
  }
  if (bias6 != nullptr) {
    bias6[0] = s.ba.x;
    bias6[1] = s.ba.y;
    bias6[2] = s.ba.z;
    bias6[3] = s.bg.x;
    bias6[4] = s.bg.y;
    bias6[5] = s.bg.z;
  }
}

// LocalTrajectoryBuilder3D::EstimateGravity (:1106-1154).  `prev` is prev_pose_ / prev_vel_ (the newest optimised state),
// `running` the preintegration since it (imu_integrator_opt_).  Kept as written: the velocities in g_est_Vs_ are rotated
// into their frames IN PLACE on every call (:1135-1136), i.e. again on every later call while they stay in the deque.
bool estimate_gravity(dliom_imu_window& w, const State& prev, const Preint& running) {
  const int win = w.o.frames_for_online_gravity_estimate;
  w.g_frames.push_back(GFrame{prev.R, prev.p, running.dt, running.dp, running.dv});
  w.g_vs.push_back(prev.v);
  if (static_cast<int>(w.g_frames.size()) <= win + 1) return false;
  w.g_frames.pop_front();
  w.g_vs.pop_front();
  std::vector<GFrame> tmp(w.g_frames.begin(), w.g_frames.end());
  const M3 RwT = transpose(w.g_frames.front().R);
  const V3 pw = w.g_frames.front().p;
  for (size_t i = 0; i < tmp.size(); ++i) {
    tmp[i].R = RwT * w.g_frames[i].R;            // T_w_inv * tsf.transform
    tmp[i].p = RwT * (w.g_frames[i].p - pw);
    w.g_vs[i] = transpose(w.g_frames[i].R) * w.g_vs[i];
  }
  const std::vector<V3> vs(w.g_vs.begin(), w.g_vs.end());
  const V3 tlb{w.o.lidar_in_imu_translation[0], w.o.lidar_in_imu_translation[1], w.o.lidar_in_imu_translation[2]};
  V3 g_B;
  if (!estimate_gravity_vector(tmp, tlb, vs, w.o.gravity, &g_B)) return false;
  w.g_est_G = w.g_frames.front().R * (-1.0 * g_B);
  return w.g_est_G.z + w.o.gravity < 0.5;
}

}  // namespace

#define DLIOM_TRY_STATUS(expr)       \
  do {                               \
    const int _s = (expr);           \
    if (_s != DLIOM_OK) return _s;   \
  } while (0)

extern "C" {

int dliom_gravity_estimate(int num_frames, const double* poses7, const double* delta_t, const double* delta_p,
                           const double* delta_v, const double* velocities, const double lidar_in_imu_translation[3],
                           double gravity_norm, double gravity_out[3], int* accepted) {
  if (num_frames < 0 || (num_frames > 0 && (poses7 == nullptr || delta_t == nullptr || delta_p == nullptr || delta_v == nullptr ||
                                             velocities == nullptr)) ||
      lidar_in_imu_translation == nullptr || gravity_out == nullptr || accepted == nullptr)
    return DLIOM_ERR_INVALID_ARGUMENT;
  std::vector<GFrame> f(static_cast<size_t>(num_frames));
  std::vector<V3> vs(static_cast<size_t>(num_frames));
  for (int i = 0; i < num_frames; ++i) {
    f[i].R = quat_to_matrix(poses7 + 7 * i + 3);
    f[i].p = {poses7[7 * i], poses7[7 * i + 1], poses7[7 * i + 2]};
    f[i].dt = delta_t[i];
    f[i].dP = {delta_p[3 * i], delta_p[3 * i + 1], delta_p[3 * i + 2]};
    f[i].dV = {delta_v[3 * i], delta_v[3 * i + 1], delta_v[3 * i + 2]};
    vs[i] = {velocities[3 * i], velocities[3 * i + 1], velocities[3 * i + 2]};
  }
  V3 g{0, 0, 0};
  const V3 tlb{lidar_in_imu_translation[0], lidar_in_imu_translation[1], lidar_in_imu_translation[2]};
  *accepted = estimate_gravity_vector(f, tlb, vs, gravity_norm, &g) ? 1 : 0;
  gravity_out[0] = g.x;
  gravity_out[1] = g.y;
  gravity_out[2] = g.z;
  return DLIOM_OK;
}

int dliom_imu_window_default_options(dliom_imu_window_options* o) {
  if (o == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  // trajectory_builder_3d.lua:86-101 (imu block) and local_trajectory_builder_3d.cc:79-92
  o->acc_noise = 3.9939570888238808e-01;
  o->gyr_noise = 1.5636343949698187e-03;
  o->acc_bias_noise = 6.4356659353532566e-05;
  o->gyr_bias_noise = 3.5640318696367613e-05;
  o->gravity = 9.80511;
  o->integration_sigma = 1e-4;
  o->prior_pose_noise = 1e-2;
  o->prior_velocity_sigma = 1e4;
  o->prior_bias_sigma = 1e-2;
  o->ceres_pose_noise_t = 5e-2;
  o->ceres_pose_noise_r = 5e-2;
  o->ceres_pose_noise_t_drift = 3e-1;
  o->ceres_pose_noise_r_drift = 1e-1;
  o->prior_gravity_noise = 1e-2;
  o->window_size = 4;
  o->iterations = 2;
  o->enable_gravity_factor = 0;                // trajectory_builder_3d.lua:31 (dlio/config/basic_config_3d.lua:80 sets true)
  o->frames_for_online_gravity_estimate = 7;   // :29
  o->lidar_in_imu_translation[0] = o->lidar_in_imu_translation[1] = o->lidar_in_imu_translation[2] = 0.0;
  o->graph_reset_every = 0;
  o->tangent_preintegration = 1;  // what the reference's GTSAM 4.0.2 build integrates (README.MD:13-15: no flag, default ON)
  o->relinearize_threshold = 0.1;  // ISAM2Params::relinearizeThreshold, :676-679 (reference-rule mode only)
  return DLIOM_OK;
}

int dliom_imu_window_create(const dliom_imu_window_options* options, dliom_imu_window** out) {
  // window_size: 0 = the reference's rule (every key kept until the graph reset, which then has to exist) or a fixed lag
  // of 2 .. 4096 states (the chain solver is linear in it)
  const bool window_ok = options != nullptr && ((options->window_size == 0 && options->graph_reset_every >= 2 &&
                                                 options->relinearize_threshold >= 0.0 && options->relinearize_threshold < 1e300) ||
                                                (options->window_size >= 2 && options->window_size <= 4096));
  if (options == nullptr || out == nullptr || !window_ok ||
      options->iterations < 1 || !(options->acc_noise > 0) || !(options->gyr_noise > 0) || !(options->acc_bias_noise > 0) ||
      !(options->gyr_bias_noise > 0) || options->graph_reset_every < 0 || options->graph_reset_every == 1 ||
      (options->tangent_preintegration != 0 && options->tangent_preintegration != 1))  // a struct filled without
    return DLIOM_ERR_INVALID_ARGUMENT;  // dliom_imu_window_default_options leaves the (round 4) trailing field indeterminate
  // the gravity factor goes on the state frames_for_online_gravity_estimate keys back (:828): it has to be in the window
  if (options->enable_gravity_factor != 0 &&
      (options->frames_for_online_gravity_estimate < 2 ||
       (options->window_size != 0 && options->window_size < options->frames_for_online_gravity_estimate + 1)))
    return DLIOM_ERR_INVALID_ARGUMENT;
  dliom_imu_window* w = new dliom_imu_window;
  w->o = *options;
  w->current.tangent = options->tangent_preintegration != 0;
  *out = w;
  return DLIOM_OK;
}

int dliom_imu_window_destroy(dliom_imu_window* w) {
  delete w;
  return DLIOM_OK;
}

}  // extern "C"

namespace {
// ResetGTSAM + the priors of the gtsam_initialized_ == false branch (:674-686,712-745): the graph becomes ONE state with
// PriorFactor<Pose3> (prior_pose_noise x 6), PriorFactor<Vector3> (1e4), PriorFactor<ConstantBias> (1e-2) at `s`, the
// running preintegration starts over at its biases, key_ = 1.  EstimateGravity's deques are the caller's business: the
// reference never clears them (a fresh dliom_imu_window_initialize does).
void restart_graph_at(dliom_imu_window* w, const State& s) {
  w->x.assign(1, s);
  w->between.clear();
  w->pose_priors.clear();
  w->gravity.clear();
  std::memset(w->H0, 0, sizeof w->H0);
  std::memset(w->b0, 0, sizeof w->b0);
  const double sp = w->o.prior_pose_noise, sv = w->o.prior_velocity_sigma, sb = w->o.prior_bias_sigma;
  for (int i = 0; i < 6; ++i) w->H0[kD * i + i] = 1.0 / (sp * sp);
  for (int i = 6; i < 9; ++i) w->H0[kD * i + i] = 1.0 / (sv * sv);
  for (int i = 9; i < 15; ++i) w->H0[kD * i + i] = 1.0 / (sb * sb);
  w->lin0 = s;
  w->delta.assign(kD, 0.0);
  w->fac.clear();
  w->blk.assign(1, dliom_imu_window::ChainBlock());
  w->clean_until = 0;
  w->current.reset(s.ba, s.bg);
  w->key = 1;
}
}  // namespace

extern "C" {

int dliom_imu_window_initialize(dliom_imu_window* w, const double pose7[7], const double velocity[3], const double bias6[6]) {
  if (w == nullptr || pose7 == nullptr || velocity == nullptr || bias6 == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  State s;
  s.R = quat_to_matrix(pose7 + 3);
  s.p = {pose7[0], pose7[1], pose7[2]};
  s.v = {velocity[0], velocity[1], velocity[2]};
  s.ba = {bias6[0], bias6[1], bias6[2]};
  s.bg = {bias6[3], bias6[4], bias6[5]};
  w->g_frames.clear();  // a fresh start: the estimator's window starts over
  w->g_vs.clear();
  w->g_est_valid = false;
  restart_graph_at(w, s);
  w->initialized = true;
  w->graph_started = false;
  w->num_states = 1;
  return DLIOM_OK;
}

int dliom_imu_window_add_imu(dliom_imu_window* w, const double acc[3], const double gyr[3], double dt) {
  if (w == nullptr || acc == nullptr || gyr == nullptr || !(dt > 0)) return DLIOM_ERR_INVALID_ARGUMENT;
  if (!w->initialized) return DLIOM_ERR_INVALID_ARGUMENT;
  integrate(w->current, {acc[0], acc[1], acc[2]}, {gyr[0], gyr[1], gyr[2]}, dt, w->o.acc_noise, w->o.gyr_noise,
            w->o.integration_sigma);
  return DLIOM_OK;
}

int dliom_imu_window_add_imu_batch(dliom_imu_window* w, int n, const double* acc, const double* gyr, const double* dt) {
  if (w == nullptr || n < 0 || (n > 0 && (acc == nullptr || gyr == nullptr || dt == nullptr))) return DLIOM_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < n; ++i) DLIOM_TRY_STATUS(dliom_imu_window_add_imu(w, acc + 3 * i, gyr + 3 * i, dt[i]));
  return DLIOM_OK;
}

int dliom_imu_window_predict(const dliom_imu_window* w, double pose7[7], double velocity[3]) {
  if (w == nullptr || !w->initialized) return DLIOM_ERR_INVALID_ARGUMENT;
  write_state(predicted(*w, estimate_at(*w, w->x.size() - 1), w->current), pose7, velocity, nullptr);
  return DLIOM_OK;
}

int dliom_imu_window_add_gravity(dliom_imu_window* w, int states_back, const double direction[3]) {
  if (w == nullptr || direction == nullptr || !w->initialized || states_back < 0 || states_back >= static_cast<int>(w->x.size()))
    return DLIOM_ERR_INVALID_ARGUMENT;
  V3 n{direction[0], direction[1], direction[2]};
  const double l = norm(n);
  if (!(l > 0)) return DLIOM_ERR_INVALID_ARGUMENT;
  dliom_imu_window::Gravity f;
  f.index = static_cast<int>(w->x.size()) - 1 - states_back;
  f.nZ = (1.0 / l) * n;
  f.bRef = {0, 0, -1};  // g_ref_B, :780,825
  f.sigma = w->o.prior_gravity_noise;
  w->gravity.push_back(f);
  chain_mark(*w, f.index);
  return DLIOM_OK;
}

}  // extern "C"

namespace {
// "reset graph for speed" (:749-792): when key_ reaches num_range_data the reference throws its graph away and starts a
// new one at the newest state, whose priors are the marginal covariances of X, V and B taken SEPARATELY -- whatever the
// old graph knew about how pose, velocity and bias errors go together is dropped.  A fixed-lag window does not need
// this (its marginal prior keeps the full 15 x 15 block); it is here so that the estimates follow the reference's
// through its resets: tests/test_imu_window.py compares a 70-scan run against a batch solver with the same rule.
// `old`: the graph that is being replaced -- w itself, or (reference-rule mode) the object w's graph was MOVED into so that
// a failed call can move it back instead of having copied ~10 KB a key.
bool reset_graph(dliom_imu_window& w, dliom_imu_window& old) {
  // marginal covariance of the newest state = inverse of the last block of the chain's factor, taken at the
  // linearisation points like ISAM2::marginalCovariance (fixed-lag mode: they are the estimates)
  if (old.clean_until < static_cast<int>(old.x.size()) && !chain_eliminate(old)) return false;
  const double* L = old.blk.back().L;
  double cov[kD][kD];
  for (int c = 0; c < kD; ++c) {
    double e[kD] = {0};
    e[c] = 1.0;
    lower_solve15(L, e);
    upper_solve15(L, e);
    for (int r = 0; r < kD; ++r) cov[r][c] = e[r];
  }
  const State newest = estimate_at(old, old.x.size() - 1);  // prev_pose_ / prev_vel_ / prev_bias_
  std::fill(w.H0, w.H0 + kD * kD, 0.0);
  std::fill(w.b0, w.b0 + kD, 0.0);
  const int first[3] = {0, 6, 9}, size[3] = {6, 3, 6};  // updatedPoseNoise, updatedVelNoise, updatedBiasNoise
  for (int blk = 0; blk < 3; ++blk) {
    const int f = first[blk], m = size[blk];
    std::vector<double> C(static_cast<size_t>(m) * m);
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) C[static_cast<size_t>(i) * m + j] = 0.5 * (cov[f + i][f + j] + cov[f + j][f + i]);
    if (!cholesky(C, m)) return false;
    for (int c = 0; c < m; ++c) {  // information = C^-1, column by column
      std::vector<double> e(m, 0.0);
      e[c] = 1.0;
      chol_solve(C, m, e.data());
      for (int r = 0; r < m; ++r) w.H0[kD * (f + r) + f + c] = e[r];
    }
  }
  for (int i = 0; i < kD; ++i)
    for (int j = i + 1; j < kD; ++j) w.H0[kD * i + j] = w.H0[kD * j + i] = 0.5 * (w.H0[kD * i + j] + w.H0[kD * j + i]);
  w.lin0 = newest;
  w.x.assign(1, newest);
  w.between.clear();
  w.pose_priors.clear();
  w.gravity.clear();
  w.delta.assign(kD, 0.0);
  w.fac.clear();
  w.blk.assign(1, dliom_imu_window::ChainBlock());
  w.clean_until = 0;
  w.key = 1;
  return true;
}
}  // namespace

namespace {
// What a failed add_pose has to put back.  The fixed-lag window is small: a copy of it.  The reference-rule graph holds
// up to num_range_data keys with their eliminations (~10 KB a key): copied only on the scan of a graph reset; otherwise
// the call's own additions are taken back by hand (a linearisation point that moved in the failed call stays where it is
// with a zero increment -- the same estimate -- and the chain is re-eliminated from the oldest state by the next call).
struct AddPoseUndo {
  std::unique_ptr<dliom_imu_window> copy;
  size_t states = 0, priors = 0, gravity = 0;
  std::vector<double> delta;
  std::deque<GFrame> g_frames;
  std::deque<V3> g_vs;
  V3 g_est_G{0, 0, 0};
  bool g_est_valid = false;
  int64_t relinearizations = 0;
  std::vector<State> x;  // linearisation points (15 doubles a key: cheap next to the eliminations)
};
void take_back(dliom_imu_window& w, AddPoseUndo& u) {
  if (u.copy) {
    w = std::move(*u.copy);
    return;
  }
  w.x.resize(u.states);
  w.between.resize(u.states - 1);
  w.fac.resize(u.states - 1);
  w.blk.resize(u.states);
  w.pose_priors.resize(u.priors);
  w.gravity.resize(u.gravity);
  w.x = u.x;
  w.delta = u.delta;
  for (auto& f : w.fac) f.stale = true;
  w.clean_until = 0;
  w.g_frames.swap(u.g_frames);
  w.g_vs.swap(u.g_vs);
  w.g_est_G = u.g_est_G;
  w.g_est_valid = u.g_est_valid;
  w.relinearizations = u.relinearizations;
}
}  // namespace

extern "C" {

int dliom_imu_window_add_pose(dliom_imu_window* w, const double matched_pose7[7], int degenerate, double pose7[7],
                              double velocity[3], double bias6[6]) {
  if (w == nullptr || matched_pose7 == nullptr || !w->initialized) return DLIOM_ERR_INVALID_ARGUMENT;
  if (!(w->current.dt > 0)) return DLIOM_ERR_INVALID_ARGUMENT;  // no IMU since the last pose
  w->graph_started = true;
  const bool reset_due = w->o.graph_reset_every > 0 && w->key == w->o.graph_reset_every;
  AddPoseUndo undo;  // a failed solve leaves the window exactly as it was
  if (!w->full_graph()) {
    undo.copy.reset(new dliom_imu_window(*w));
  } else if (reset_due) {
    // the graph moves into the undo object (vectors change hands, nothing is copied); what outlives a reset -- options,
    // the running preintegration, the gravity estimator's deques, counters -- is put back into *w right away
    undo.copy.reset(new dliom_imu_window(std::move(*w)));
    const dliom_imu_window& old = *undo.copy;
    w->g_frames = old.g_frames;
    w->g_vs = old.g_vs;
    w->x.clear();
    w->between.clear();
    w->pose_priors.clear();
    w->gravity.clear();
    w->delta.clear();
    w->fac.clear();
    w->blk.clear();
  } else {
    undo.states = w->x.size();
    undo.priors = w->pose_priors.size();
    undo.gravity = w->gravity.size();
    undo.delta = w->delta;
    undo.x = w->x;
    undo.g_frames = w->g_frames;
    undo.g_vs = w->g_vs;
    undo.g_est_G = w->g_est_G;
    undo.g_est_valid = w->g_est_valid;
    undo.relinearizations = w->relinearizations;
  }
  // prev_state_: the reference predicts from the estimate it read after the previous scan, also across a reset
  dliom_imu_window& graph = (w->full_graph() && reset_due) ? *undo.copy : *w;  // where the graph is right now
  const State prev = estimate_at(graph, graph.x.size() - 1);
  if (reset_due) {
    if (!reset_graph(*w, graph)) {
      take_back(*w, undo);
      return DLIOM_ERR_SOLVER;
    }
    if (w->o.enable_gravity_factor != 0) {  // :772-782: EstimateGravity() here too (its deques get this frame twice)
      w->g_est_valid = estimate_gravity(*w, prev, w->current);
      if (w->g_est_valid) {
        dliom_imu_window::Gravity gf;
        gf.index = 0;
        gf.nZ = (1.0 / norm(w->g_est_G)) * w->g_est_G;
        gf.bRef = {0, 0, -1};
        gf.sigma = w->o.prior_gravity_noise;
        w->gravity.push_back(gf);
        ++w->gravity_factors;
        if (!gauss_newton(*w, 1)) {  // "optimize once" (:788)
          take_back(*w, undo);
          return DLIOM_ERR_SOLVER;
        }
      }
    }
  }
  // new state at the IMU prediction (:833-838), IMU factor + bias random walk to it, pose prior on it
  State next = predicted(*w, prev, w->current);
  next.ba = prev.ba;
  next.bg = prev.bg;
  w->x.push_back(next);
  w->between.push_back(w->current);
  if (!chain_push_state(*w)) {
    take_back(*w, undo);
    return DLIOM_ERR_SOLVER;
  }
  dliom_imu_window::PosePrior f;
  f.index = static_cast<int>(w->x.size()) - 1;
  f.R = quat_to_matrix(matched_pose7 + 3);
  f.p = {matched_pose7[0], matched_pose7[1], matched_pose7[2]};
  // sigmas as the reference fills them, (t, t, t, r, r, r), against GTSAM's (rotation, translation) tangent order
  f.sigma_rot = degenerate ? w->o.ceres_pose_noise_t_drift : w->o.ceres_pose_noise_t;
  f.sigma_trans = degenerate ? w->o.ceres_pose_noise_r_drift : w->o.ceres_pose_noise_r;
  w->pose_priors.push_back(f);
  // gravity factor for the state frames_for_online_gravity_estimate keys back (:819-831); key_ of the new state is
  // w->key (1 for the first scan after initialisation or a reset)
  bool gravity_added = false;
  if (w->o.enable_gravity_factor != 0) {
    w->g_est_valid = estimate_gravity(*w, prev, w->current);
    const int win = w->o.frames_for_online_gravity_estimate;
    if (w->g_est_valid && w->key - win >= 0 && static_cast<int>(w->x.size()) - 1 - win >= 0) {
      dliom_imu_window::Gravity gf;
      gf.index = static_cast<int>(w->x.size()) - 1 - win;
      gf.nZ = (1.0 / norm(w->g_est_G)) * w->g_est_G;
      gf.bRef = {0, 0, -1};
      gf.sigma = w->o.prior_gravity_noise;
      w->gravity.push_back(gf);
      chain_mark(*w, gf.index);
      gravity_added = true;
    }
  }
  if (!gauss_newton(*w, w->o.iterations)) {
    // nothing of this scan stays: the new key, its factors, the estimator's entry and a graph reset are taken back and
    // the running preintegration is kept, so that the caller may try again (or re-initialise) without IMU samples
    // counted twice
    take_back(*w, undo);
    return DLIOM_ERR_SOLVER;
  }
  if (gravity_added) ++w->gravity_factors;
  if (!w->full_graph())
    while (static_cast<int>(w->x.size()) > w->o.window_size)
      if (!marginalize_oldest(*w)) {  // a failed Schur complement is a failed solve: nothing of this scan stays either
        take_back(*w, undo);
        return DLIOM_ERR_SOLVER;
      }
  const State s = estimate_at(*w, w->x.size() - 1);
  w->current.reset(s.ba, s.bg);  // resetIntegrationAndSetBias(prev_bias_), :852
  ++w->num_states;
  ++w->key;
  write_state(s, pose7, velocity, bias6);
  if (norm(s.v) > 30.0 || norm(s.ba) > 1.0 || norm(s.bg) > 1.0) {  // FailureDetection, :856-859,896-913
    // ResetParams(): gtsam_initialized_ = false and nothing else -- prev_state_ / prev_bias_ stay what this scan made them,
    // and the next WindowOptimize starts a new graph there (dliom_imu_window_window_optimize does; a caller of the add_pose
    // primitive re-initialises, or goes on adding keys to the graph as it is)
    w->graph_started = false;
    return DLIOM_ERR_DIVERGED;
  }
  return DLIOM_OK;
}

// LocalTrajectoryBuilder3D::WindowOptimize(matched_pose, is_drift) as the reference calls it once per scan (:693-863),
// its first call included: while gtsam_initialized_ is false (:712-745) the call only STARTS the graph -- priors on X(0),
// V(0), B(0) at prev_state_ / prev_bias_ as InitializeIMU left them, the preintegration since then dropped
// (resetIntegrationAndSetBias, :739), key_ = 1 -- and the scan's matched pose is not used: the caller's opt_pose is
// prev_state_ (:555-557).  Every later call is dliom_imu_window_add_pose.
int dliom_imu_window_window_optimize(dliom_imu_window* w, const double matched_pose7[7], int is_drift, double pose7[7],
                                     double velocity[3], double bias6[6]) {
  if (w == nullptr || matched_pose7 == nullptr || !w->initialized) return DLIOM_ERR_INVALID_ARGUMENT;
  if (w->graph_started) return dliom_imu_window_add_pose(w, matched_pose7, is_drift, pose7, velocity, bias6);
  const State s = estimate_at(*w, w->x.size() - 1);  // prev_state_ / prev_bias_: InitializeIMU's, or what a diverged scan left
  restart_graph_at(w, s);
  w->graph_started = true;
  write_state(s, pose7, velocity, bias6);
  return DLIOM_OK;
}

int dliom_imu_window_solver_stats(const dliom_imu_window* w, int64_t* relinearizations, int64_t* blocks_eliminated) {
  if (w == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  if (relinearizations != nullptr) *relinearizations = w->relinearizations;
  if (blocks_eliminated != nullptr) *blocks_eliminated = w->blocks_eliminated;
  return DLIOM_OK;
}

// Diagnostic: the Jacobian of the IMU factor (+ bias random walk) between the window's two newest states, analytic (what
// the solver uses) and by central differences of the residual (what it used until round 3), 15 x 30 row major each.
int dliom_diag_imu_factor_jacobians(dliom_imu_window* w, double* analytic, double* numeric) {
  if (w == nullptr || analytic == nullptr || numeric == nullptr || !w->initialized || w->x.size() < 2) return DLIOM_ERR_INVALID_ARGUMENT;
  const int i = static_cast<int>(w->x.size()) - 2;
  const Preint& P = w->between[static_cast<size_t>(i)];
  std::vector<double> Linv;
  if (!whitening(P, &Linv)) return DLIOM_ERR_SOLVER;
  double r[15];
  imu_factor_jacobian(*w, P, Linv, w->x[i], w->x[i + 1], r, analytic);
  const double eps = 1e-6;
  for (int c = 0; c < 30; ++c) {
    const int si = c < kD ? i : i + 1;
    const State keep = w->x[si];
    double d[kD] = {0}, rp[15], rm[15];
    d[c % kD] = eps;
    w->x[si] = retract(keep, d);
    imu_residual(*w, P, Linv, w->x[i], w->x[i + 1], rp);
    d[c % kD] = -eps;
    w->x[si] = retract(keep, d);
    imu_residual(*w, P, Linv, w->x[i], w->x[i + 1], rm);
    w->x[si] = keep;
    for (int k = 0; k < 15; ++k) numeric[30 * k + c] = (rp[k] - rm[k]) / (2 * eps);
  }
  return DLIOM_OK;
}

int dliom_imu_window_state(const dliom_imu_window* w, int states_back, double pose7[7], double velocity[3], double bias6[6]) {
  if (w == nullptr || !w->initialized || states_back < 0 || states_back >= static_cast<int>(w->x.size()))
    return DLIOM_ERR_INVALID_ARGUMENT;
  write_state(estimate_at(*w, w->x.size() - 1 - states_back), pose7, velocity, bias6);
  return DLIOM_OK;
}

int dliom_imu_window_size(const dliom_imu_window* w) { return w == nullptr ? 0 : static_cast<int>(w->x.size()); }

int dliom_imu_window_gravity_estimate(const dliom_imu_window* w, double gravity_in_global[3], int* valid, int64_t* factors_added) {
  if (w == nullptr || gravity_in_global == nullptr || valid == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  gravity_in_global[0] = w->g_est_G.x;
  gravity_in_global[1] = w->g_est_G.y;
  gravity_in_global[2] = w->g_est_G.z;
  *valid = w->g_est_valid ? 1 : 0;
  if (factors_added != nullptr) *factors_added = w->gravity_factors;
  return DLIOM_OK;
}

}  // extern "C"
