// ORACLE (test infrastructure, NOT product code).
//
// IMU preintegration as the reference holds it in-tree:
//   mapping/internal/3d/initialization/integration_base.h
//     :27-52,70-98   construction / resetIntegration (noise matrix)
//     :106-123       push_back (the first sample only seeds acc_0 / gyr_0)
//     :140-155       repropagate
//     :157-248       midPointIntegration (state, F, V, jacobian and covariance propagation)
//     :250-278       propagate
//     :280-316       evaluate (commented out in the reference; VINS-Mono's residual)
// PARITY UNPINNED: the reference has no test for this class, and its steady-state window uses
// GTSAM's PreintegratedImuMeasurements instead.  Dense products are plain
// row-by-column loops here, not Eigen's blocked GEMM; the product (csrc/imu_preintegration.cc) is
// written the same way so the two can be compared bit for bit.
// Row/column layout of F / jacobian / covariance follows the CODE of midPointIntegration
// (position 0, rotation 3, velocity 6, accelerometer bias 9, gyroscope bias 12), which is VINS-Mono's;
// the fork's StateOrder enum (O_R = 0, O_P = 3) does not match it and is only used by the
// commented-out evaluate().
#ifndef ORACLE_OM_IMU_H_
#define ORACLE_OM_IMU_H_

#include <array>
#include <cmath>
#include <vector>

#include "om_math.h"

namespace oracle {

struct Mat3 {
  double m[3][3];
};
inline Mat3 Mat3Identity() { return Mat3{{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}}; }
inline Mat3 Skew(const Vec3d& v) { return Mat3{{{0, -v.z, v.y}, {v.z, 0, -v.x}, {-v.y, v.x, 0}}}; }
inline Mat3 Mul(const Mat3& a, const Mat3& b) {
  Mat3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.;
      for (int k = 0; k < 3; ++k) s += a.m[i][k] * b.m[k][j];
      r.m[i][j] = s;
    }
  return r;
}
inline Mat3 Add(const Mat3& a, const Mat3& b) {
  Mat3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] + b.m[i][j];
  return r;
}
inline Mat3 Scale(const Mat3& a, double s) {
  Mat3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] * s;
  return r;
}
// Eigen QuaternionBase::toRotationMatrix()
inline Mat3 ToRotationMatrix(const Quatd& q) {
  const double tx = 2. * q.x, ty = 2. * q.y, tz = 2. * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  return Mat3{{{1. - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1. - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1. - (txx + tyy)}}};
}

using Mat15 = std::array<std::array<double, 15>, 15>;

struct ImuNoise {
  double acc_n, gyr_n, acc_w, gyr_w;
};

class IntegrationBase {
 public:
  IntegrationBase(const Vec3d& ba, const Vec3d& bg, const ImuNoise& n) { Reset(ba, bg, n); }

  void Reset(const Vec3d& ba, const Vec3d& bg, const ImuNoise& n) {  // :70-98
    dt_ = -1.;
    sum_dt = 0.;
    linearized_ba = ba;
    linearized_bg = bg;
    for (int i = 0; i < 15; ++i)
      for (int j = 0; j < 15; ++j) {
        jacobian[i][j] = i == j ? 1. : 0.;
        covariance[i][j] = 0.;
      }
    delta_p = Vec3d(0, 0, 0);
    delta_q = Quatd();
    delta_v = Vec3d(0, 0, 0);
    // 18 noise variances (diagonal): acc_n, gyr_n, acc_n, gyr_n, acc_w, gyr_w -- three each
    const double v[6] = {n.acc_n * n.acc_n, n.gyr_n * n.gyr_n, n.acc_n * n.acc_n, n.gyr_n * n.gyr_n, n.acc_w * n.acc_w,
                         n.gyr_w * n.gyr_w};
    for (int b = 0; b < 6; ++b)
      for (int k = 0; k < 3; ++k) noise[3 * b + k] = v[b];
    dt_buf.clear();
    acc_buf.clear();
    gyr_buf.clear();
  }

  void push_back(double dt, const Vec3d& acc, const Vec3d& gyr) {  // :106-123
    if (dt_ < 0.) {
      dt_ = 1e-6;
      acc_0 = acc;
      gyr_0 = gyr;
      linearized_acc = acc_0;
      linearized_gyr = gyr_0;
      return;
    }
    dt_buf.push_back(dt);
    acc_buf.push_back(acc);
    gyr_buf.push_back(gyr);
    Propagate(dt, acc, gyr);
  }

  void repropagate(const Vec3d& ba, const Vec3d& bg) {  // :140-155
    sum_dt = 0.;
    acc_0 = linearized_acc;
    gyr_0 = linearized_gyr;
    delta_p = Vec3d(0, 0, 0);
    delta_q = Quatd();
    delta_v = Vec3d(0, 0, 0);
    linearized_ba = ba;
    linearized_bg = bg;
    for (int i = 0; i < 15; ++i)
      for (int j = 0; j < 15; ++j) {
        jacobian[i][j] = i == j ? 1. : 0.;
        covariance[i][j] = 0.;
      }
    for (size_t i = 0; i < dt_buf.size(); ++i) Propagate(dt_buf[i], acc_buf[i], gyr_buf[i]);
  }

  // VINS-Mono's residual (the reference keeps it commented out, :280-316), rows P, R, V, BA, BG.
  // G is the gravity vector in the world frame, e.g. (0, 0, 9.8).
  std::array<double, 15> evaluate(const Vec3d& Pi, const Quatd& Qi, const Vec3d& Vi, const Vec3d& Bai, const Vec3d& Bgi,
                                  const Vec3d& Pj, const Quatd& Qj, const Vec3d& Vj, const Vec3d& Baj, const Vec3d& Bgj,
                                  const Vec3d& G) const {
    auto block = [&](int r, int c) {
      Mat3 b;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) b.m[i][j] = jacobian[r + i][c + j];
      return b;
    };
    auto mulv = [](const Mat3& a, const Vec3d& v) {
      return Vec3d(a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
                   a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z);
    };
    const Vec3d dba = Bai - linearized_ba, dbg = Bgi - linearized_bg;
    const Vec3d th = mulv(block(3, 12), dbg);
    const Quatd corrected_q = delta_q * Quatd(1., th.x / 2., th.y / 2., th.z / 2.);
    const Vec3d corrected_v = delta_v + mulv(block(6, 9), dba) + mulv(block(6, 12), dbg);
    const Vec3d corrected_p = delta_p + mulv(block(0, 9), dba) + mulv(block(0, 12), dbg);
    const Quatd qi_inv = Qi.conjugate();  // unit quaternion
    const Vec3d rp = qi_inv * (0.5 * sum_dt * sum_dt * G + Pj - Pi - sum_dt * Vi) - corrected_p;
    const Quatd dq = corrected_q.conjugate() * (qi_inv * Qj);
    const Vec3d rv = qi_inv * (sum_dt * G + Vj - Vi) - corrected_v;
    const Vec3d rba = Baj - Bai, rbg = Bgj - Bgi;
    return {rp.x, rp.y, rp.z, 2. * dq.x, 2. * dq.y, 2. * dq.z, rv.x, rv.y, rv.z, rba.x, rba.y, rba.z, rbg.x, rbg.y, rbg.z};
  }

  double sum_dt = 0.;
  Vec3d delta_p, delta_v;
  Quatd delta_q;
  Vec3d linearized_ba, linearized_bg;
  Mat15 jacobian, covariance;

 private:
  void Propagate(double dt, const Vec3d& acc_1, const Vec3d& gyr_1) {  // :250-278 + 157-248
    dt_ = dt;
    const Vec3d un_acc_0 = delta_q * (acc_0 - linearized_ba);
    const Vec3d un_gyr = 0.5 * (gyr_0 + gyr_1) - linearized_bg;
    const Quatd result_q = delta_q * Quatd(1., un_gyr.x * dt / 2., un_gyr.y * dt / 2., un_gyr.z * dt / 2.);
    const Vec3d un_acc_1 = result_q * (acc_1 - linearized_ba);
    const Vec3d un_acc = 0.5 * (un_acc_0 + un_acc_1);
    const Vec3d result_p = delta_p + dt * delta_v + (0.5 * dt * dt) * un_acc;
    const Vec3d result_v = delta_v + dt * un_acc;

    const Vec3d w_x = 0.5 * (gyr_0 + gyr_1) - linearized_bg;
    const Mat3 R_w = Skew(w_x), R_a0 = Skew(acc_0 - linearized_ba), R_a1 = Skew(acc_1 - linearized_ba);
    const Mat3 Rq = ToRotationMatrix(delta_q), Rr = ToRotationMatrix(result_q), I = Mat3Identity();
    const Mat3 I_minus = Add(I, Scale(R_w, -dt));
    Mat15 F{};
    auto set = [](Mat15& M, int r, int c, const Mat3& b) {
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[r + i][c + j] = b.m[i][j];
    };
    set(F, 0, 0, I);
    set(F, 0, 3, Add(Scale(Mul(Rq, R_a0), -0.25 * dt * dt), Scale(Mul(Mul(Rr, R_a1), I_minus), -0.25 * dt * dt)));
    set(F, 0, 6, Scale(I, dt));
    set(F, 0, 9, Scale(Add(Rq, Rr), -0.25 * dt * dt));
    set(F, 0, 12, Scale(Mul(Rr, R_a1), -0.25 * dt * dt * -dt));
    set(F, 3, 3, I_minus);
    set(F, 3, 12, Scale(I, -dt));
    set(F, 6, 3, Add(Scale(Mul(Rq, R_a0), -0.5 * dt), Scale(Mul(Mul(Rr, R_a1), I_minus), -0.5 * dt)));
    set(F, 6, 6, I);
    set(F, 6, 9, Scale(Add(Rq, Rr), -0.5 * dt));
    set(F, 6, 12, Scale(Mul(Rr, R_a1), -0.5 * dt * -dt));
    set(F, 9, 9, I);
    set(F, 12, 12, I);
    std::array<std::array<double, 18>, 15> V{};
    auto setv = [&](int r, int c, const Mat3& b) {
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[r + i][c + j] = b.m[i][j];
    };
    const Mat3 v03 = Scale(Mul(Rr, R_a1), 0.25 * -1. * dt * dt * 0.5 * dt);
    const Mat3 v63 = Scale(Mul(Rr, R_a1), 0.5 * -1. * dt * 0.5 * dt);
    setv(0, 0, Scale(Rq, 0.25 * dt * dt));
    setv(0, 3, v03);
    setv(0, 6, Scale(Rr, 0.25 * dt * dt));
    setv(0, 9, v03);
    setv(3, 3, Scale(I, 0.5 * dt));
    setv(3, 9, Scale(I, 0.5 * dt));
    setv(6, 0, Scale(Rq, 0.5 * dt));
    setv(6, 3, v63);
    setv(6, 6, Scale(Rr, 0.5 * dt));
    setv(6, 9, v63);
    setv(9, 12, Scale(I, dt));
    setv(12, 15, Scale(I, dt));
    // jacobian = F jacobian ; covariance = F covariance F^T + V noise V^T
    Mat15 nj{}, fc{}, nc{};
    for (int i = 0; i < 15; ++i)
      for (int j = 0; j < 15; ++j) {
        double s = 0., t = 0.;
        for (int k = 0; k < 15; ++k) {
          s += F[i][k] * jacobian[k][j];
          t += F[i][k] * covariance[k][j];
        }
        nj[i][j] = s;
        fc[i][j] = t;
      }
    for (int i = 0; i < 15; ++i)
      for (int j = 0; j < 15; ++j) {
        double s = 0.;
        for (int k = 0; k < 15; ++k) s += fc[i][k] * F[j][k];
        double t = 0.;
        for (int k = 0; k < 18; ++k) t += (V[i][k] * noise[k]) * V[j][k];
        nc[i][j] = s + t;
      }
    jacobian = nj;
    covariance = nc;
    delta_p = result_p;
    delta_q = result_q.normalized();
    delta_v = result_v;
    sum_dt += dt;
    acc_0 = acc_1;
    gyr_0 = gyr_1;
  }

  double dt_ = -1.;
  Vec3d acc_0, gyr_0, linearized_acc, linearized_gyr;
  double noise[18];
  std::vector<double> dt_buf;
  std::vector<Vec3d> acc_buf, gyr_buf;
};

}  // namespace oracle

#endif  // ORACLE_OM_IMU_H_
