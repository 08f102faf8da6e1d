// IMU preintegration between two scans on the host (15 residuals over a handful of samples: not
// GPU work).  Mid-point integration with first-order bias Jacobians and covariance propagation as
// the reference keeps it in-tree for initialisation,
//   mapping/internal/3d/initialization/integration_base.h:106-123 (push_back), :140-155
//   (repropagate), :157-248 (midPointIntegration), :250-278 (propagate), :280-316 (the VINS-Mono
//   residual, commented out there),
// offered as the self-contained replacement SURVEY.md 8(f) rank 4 names for the GTSAM
// PreintegratedImuMeasurements the steady-state window uses
// (local_trajectory_builder_3d.cc:179-199: integrateMeasurement / predict).  PARITY UNPINNED: the
// reference has no test for either; tests compare with the oracle's independent restatement and with
// closed-form motion.  Block layout of jacobian / covariance: P 0, R 3, V 6, BA 9, BG 12 (what
// midPointIntegration's F actually fills).
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/dliom.h"

namespace {

struct V3 {
  double v[3];
};
struct Q4 {
  double w, x, y, z;
};
typedef double M3[3][3];

V3 sub(const V3& a, const V3& b) { return V3{{a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]}}; }
V3 add(const V3& a, const V3& b) { return V3{{a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]}}; }
V3 scl(double s, const V3& a) { return V3{{s * a.v[0], s * a.v[1], s * a.v[2]}}; }

// Eigen Quaterniond product, SSE2 evaluation order (host_math.h::qmul_d).
Q4 qmul(const Q4& a, const Q4& b) {
  const double t1x = a.w * b.x + a.y * b.z, t1y = a.w * b.y + a.y * b.w;
  const double t2x = a.z * b.x - a.x * b.z, t2y = a.z * b.y - a.x * b.w;
  const double u1z = a.w * b.z - a.y * b.x, u1w = a.w * b.w - a.y * b.y;
  const double u2z = a.z * b.z + a.x * b.x, u2w = a.z * b.w + a.x * b.y;
  return Q4{u1w - u2z, t1x - t2y, t1y + t2x, u1z + u2w};
}
Q4 qnormalized(const Q4& q) {
  const double z2 = (q.x * q.x + q.z * q.z) + (q.y * q.y + q.w * q.w);
  if (z2 > 0.) {
    const double n = std::sqrt(z2);
    return Q4{q.w / n, q.x / n, q.y / n, q.z / n};
  }
  return q;
}
V3 qrot(const Q4& q, const V3& v) {  // QuaternionBase::_transformVector
  double uv[3] = {q.y * v.v[2] - q.z * v.v[1], q.z * v.v[0] - q.x * v.v[2], q.x * v.v[1] - q.y * v.v[0]};
  for (double& u : uv) u += u;
  const double c[3] = {q.y * uv[2] - q.z * uv[1], q.z * uv[0] - q.x * uv[2], q.x * uv[1] - q.y * uv[0]};
  return V3{{(v.v[0] + q.w * uv[0]) + c[0], (v.v[1] + q.w * uv[1]) + c[1], (v.v[2] + q.w * uv[2]) + c[2]}};
}
void rotation_matrix(const Q4& q, M3 r) {  // QuaternionBase::toRotationMatrix
  const double tx = 2. * q.x, ty = 2. * q.y, tz = 2. * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  r[0][0] = 1. - (tyy + tzz); r[0][1] = txy - twz;        r[0][2] = txz + twy;
  r[1][0] = txy + twz;        r[1][1] = 1. - (txx + tzz); r[1][2] = tyz - twx;
  r[2][0] = txz - twy;        r[2][1] = tyz + twx;        r[2][2] = 1. - (txx + tyy);
}
void skew(const V3& a, M3 r) {
  r[0][0] = 0;       r[0][1] = -a.v[2]; r[0][2] = a.v[1];
  r[1][0] = a.v[2];  r[1][1] = 0;       r[1][2] = -a.v[0];
  r[2][0] = -a.v[1]; r[2][1] = a.v[0];  r[2][2] = 0;
}
void mul3(const M3 a, const M3 b, M3 r) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.;
      for (int k = 0; k < 3; ++k) s += a[i][k] * b[k][j];
      r[i][j] = s;
    }
}

}  // namespace

struct dliom_imu_integrator {
  double dt = -1., sum_dt = 0.;
  V3 acc_0, gyr_0, linearized_acc, linearized_gyr, ba, bg, dp, dv;
  Q4 dq{1, 0, 0, 0};
  double J[15][15], C[15][15], noise[18];
  std::vector<double> dt_buf;
  std::vector<V3> acc_buf, gyr_buf;

  void clear_state() {
    sum_dt = 0.;
    dp = V3{{0, 0, 0}};
    dv = V3{{0, 0, 0}};
    dq = Q4{1, 0, 0, 0};
    for (int i = 0; i < 15; ++i)
      for (int j = 0; j < 15; ++j) {
        J[i][j] = i == j ? 1. : 0.;
        C[i][j] = 0.;
      }
  }
  void reset(const double* a, const double* g, const dliom_imu_noise& n) {  // :70-98
    dt = -1.;
    std::memcpy(ba.v, a, 24);
    std::memcpy(bg.v, g, 24);
    clear_state();
    const double var[6] = {n.acc_n * n.acc_n, n.gyr_n * n.gyr_n, n.acc_n * n.acc_n, n.gyr_n * n.gyr_n,
                           n.acc_w * n.acc_w, n.gyr_w * n.gyr_w};
    for (int b = 0; b < 6; ++b)
      for (int k = 0; k < 3; ++k) noise[3 * b + k] = var[b];
    dt_buf.clear();
    acc_buf.clear();
    gyr_buf.clear();
  }
  void propagate(double h, const V3& acc_1, const V3& gyr_1) {  // :157-278
    dt = h;
    const V3 un_acc_0 = qrot(dq, sub(acc_0, ba));
    const V3 un_gyr = sub(scl(0.5, add(gyr_0, gyr_1)), bg);
    const Q4 rq = qmul(dq, Q4{1., un_gyr.v[0] * h / 2., un_gyr.v[1] * h / 2., un_gyr.v[2] * h / 2.});
    const V3 un_acc_1 = qrot(rq, sub(acc_1, ba));
    const V3 un_acc = scl(0.5, add(un_acc_0, un_acc_1));
    const V3 rp = add(add(dp, scl(h, dv)), scl(0.5 * h * h, un_acc));
    const V3 rv = add(dv, scl(h, un_acc));
    M3 Rw, Ra0, Ra1, Rq, Rr, RqA0, RrA1, Im, RrA1Im;
    skew(un_gyr, Rw);
    skew(sub(acc_0, ba), Ra0);
    skew(sub(acc_1, ba), Ra1);
    rotation_matrix(dq, Rq);
    rotation_matrix(rq, Rr);
    mul3(Rq, Ra0, RqA0);
    mul3(Rr, Ra1, RrA1);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Im[i][j] = (i == j ? 1. : 0.) + Rw[i][j] * -h;
    mul3(RrA1, Im, RrA1Im);
    double F[15][15] = {}, V[15][18] = {};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        const double id = i == j ? 1. : 0.;
        F[i][j] = id;
        F[i][3 + j] = RqA0[i][j] * (-0.25 * h * h) + RrA1Im[i][j] * (-0.25 * h * h);
        F[i][6 + j] = id * h;
        F[i][9 + j] = (Rq[i][j] + Rr[i][j]) * (-0.25 * h * h);
        F[i][12 + j] = RrA1[i][j] * (-0.25 * h * h * -h);
        F[3 + i][3 + j] = Im[i][j];
        F[3 + i][12 + j] = id * -h;
        F[6 + i][3 + j] = RqA0[i][j] * (-0.5 * h) + RrA1Im[i][j] * (-0.5 * h);
        F[6 + i][6 + j] = id;
        F[6 + i][9 + j] = (Rq[i][j] + Rr[i][j]) * (-0.5 * h);
        F[6 + i][12 + j] = RrA1[i][j] * (-0.5 * h * -h);
        F[9 + i][9 + j] = id;
        F[12 + i][12 + j] = id;
        V[i][j] = Rq[i][j] * (0.25 * h * h);
        V[i][3 + j] = RrA1[i][j] * (0.25 * -1. * h * h * 0.5 * h);
        V[i][6 + j] = Rr[i][j] * (0.25 * h * h);
        V[i][9 + j] = V[i][3 + j];
        V[3 + i][3 + j] = id * (0.5 * h);
        V[3 + i][9 + j] = id * (0.5 * h);
        V[6 + i][j] = Rq[i][j] * (0.5 * h);
        V[6 + i][3 + j] = RrA1[i][j] * (0.5 * -1. * h * 0.5 * h);
        V[6 + i][6 + j] = Rr[i][j] * (0.5 * h);
        V[6 + i][9 + j] = V[6 + i][3 + j];
        V[9 + i][12 + j] = id * h;
        V[12 + i][15 + j] = id * h;
      }
    double nj[15][15], fc[15][15], nc[15][15];
    for (int i = 0; i < 15; ++i)
      for (int j = 0; j < 15; ++j) {
        double s = 0., t = 0.;
        for (int k = 0; k < 15; ++k) {
          s += F[i][k] * J[k][j];
          t += F[i][k] * C[k][j];
        }
        nj[i][j] = s;
        fc[i][j] = t;
      }
    for (int i = 0; i < 15; ++i)
      for (int j = 0; j < 15; ++j) {
        double s = 0., t = 0.;
        for (int k = 0; k < 15; ++k) s += fc[i][k] * F[j][k];
        for (int k = 0; k < 18; ++k) t += (V[i][k] * noise[k]) * V[j][k];
        nc[i][j] = s + t;
      }
    std::memcpy(J, nj, sizeof(nj));
    std::memcpy(C, nc, sizeof(nc));
    dp = rp;
    dq = qnormalized(rq);
    dv = rv;
    sum_dt += h;
    acc_0 = acc_1;
    gyr_0 = gyr_1;
  }
};

extern "C" {

int dliom_imu_integrator_create(const double ba[3], const double bg[3], const dliom_imu_noise* noise,
                                dliom_imu_integrator** out) {
  if (ba == nullptr || bg == nullptr || noise == nullptr || out == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  *out = new dliom_imu_integrator;
  (*out)->reset(ba, bg, *noise);
  return DLIOM_OK;
}
int dliom_imu_integrator_destroy(dliom_imu_integrator* m) {
  if (m == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  delete m;
  return DLIOM_OK;
}
int dliom_imu_integrator_reset(dliom_imu_integrator* m, const double ba[3], const double bg[3],
                               const dliom_imu_noise* noise) {
  if (m == nullptr || ba == nullptr || bg == nullptr || noise == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  m->reset(ba, bg, *noise);
  return DLIOM_OK;
}
int dliom_imu_integrator_push_back(dliom_imu_integrator* m, double dt, const double acc[3], const double gyr[3]) {
  if (m == nullptr || acc == nullptr || gyr == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  V3 a, g;
  std::memcpy(a.v, acc, 24);
  std::memcpy(g.v, gyr, 24);
  if (m->dt < 0.) {  // the first sample only seeds the mid-point rule (:108-115)
    m->dt = 1e-6;
    m->acc_0 = m->linearized_acc = a;
    m->gyr_0 = m->linearized_gyr = g;
    return DLIOM_OK;
  }
  m->dt_buf.push_back(dt);
  m->acc_buf.push_back(a);
  m->gyr_buf.push_back(g);
  m->propagate(dt, a, g);
  return DLIOM_OK;
}
int dliom_imu_integrator_repropagate(dliom_imu_integrator* m, const double ba[3], const double bg[3]) {
  if (m == nullptr || ba == nullptr || bg == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  m->acc_0 = m->linearized_acc;
  m->gyr_0 = m->linearized_gyr;
  std::memcpy(m->ba.v, ba, 24);
  std::memcpy(m->bg.v, bg, 24);
  m->clear_state();
  for (size_t i = 0; i < m->dt_buf.size(); ++i) m->propagate(m->dt_buf[i], m->acc_buf[i], m->gyr_buf[i]);
  return DLIOM_OK;
}
int dliom_imu_integrator_get(const dliom_imu_integrator* m, dliom_imu_preintegration* out) {
  if (m == nullptr || out == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  out->sum_dt = m->sum_dt;
  std::memcpy(out->delta_p, m->dp.v, 24);
  std::memcpy(out->delta_v, m->dv.v, 24);
  out->delta_q[0] = m->dq.w;
  out->delta_q[1] = m->dq.x;
  out->delta_q[2] = m->dq.y;
  out->delta_q[3] = m->dq.z;
  std::memcpy(out->linearized_ba, m->ba.v, 24);
  std::memcpy(out->linearized_bg, m->bg.v, 24);
  std::memcpy(out->jacobian, m->J, sizeof(m->J));
  std::memcpy(out->covariance, m->C, sizeof(m->C));
  return DLIOM_OK;
}

// state = [P(3), Q(w,x,y,z), V(3), Ba(3), Bg(3)]
int dliom_imu_integrator_evaluate(const dliom_imu_integrator* m, const double si[16], const double sj[16],
                                  const double gravity[3], double residuals[15]) {
  if (m == nullptr || si == nullptr || sj == nullptr || gravity == nullptr || residuals == nullptr)
    return DLIOM_ERR_INVALID_ARGUMENT;
  auto v3 = [](const double* p) { return V3{{p[0], p[1], p[2]}}; };
  auto mulv = [&](int r, int c, const V3& x) {
    V3 o;
    for (int i = 0; i < 3; ++i) o.v[i] = m->J[r + i][c] * x.v[0] + m->J[r + i][c + 1] * x.v[1] + m->J[r + i][c + 2] * x.v[2];
    return o;
  };
  const V3 Pi = v3(si), Vi = v3(si + 7), Bai = v3(si + 10), Bgi = v3(si + 13);
  const V3 Pj = v3(sj), Vj = v3(sj + 7), Baj = v3(sj + 10), Bgj = v3(sj + 13), G = v3(gravity);
  const Q4 Qi{si[3], si[4], si[5], si[6]}, Qj{sj[3], sj[4], sj[5], sj[6]};
  const V3 dba = sub(Bai, m->ba), dbg = sub(Bgi, m->bg);
  const V3 th = mulv(3, 12, dbg);
  const Q4 cq = qmul(m->dq, Q4{1., th.v[0] / 2., th.v[1] / 2., th.v[2] / 2.});
  const V3 cv = add(add(m->dv, mulv(6, 9, dba)), mulv(6, 12, dbg));
  const V3 cp = add(add(m->dp, mulv(0, 9, dba)), mulv(0, 12, dbg));
  const Q4 qi_inv{Qi.w, -Qi.x, -Qi.y, -Qi.z};
  const double T = m->sum_dt;
  const V3 rp = sub(qrot(qi_inv, sub(sub(add(scl(0.5 * T * T, G), Pj), Pi), scl(T, Vi))), cp);
  const Q4 dq = qmul(Q4{cq.w, -cq.x, -cq.y, -cq.z}, qmul(qi_inv, Qj));
  const V3 rv = sub(qrot(qi_inv, sub(add(scl(T, G), Vj), Vi)), cv);
  const V3 rba = sub(Baj, Bai), rbg = sub(Bgj, Bgi);
  const double out[15] = {rp.v[0], rp.v[1], rp.v[2], 2. * dq.x, 2. * dq.y, 2. * dq.z, rv.v[0], rv.v[1], rv.v[2],
                          rba.v[0], rba.v[1], rba.v[2], rbg.v[0], rbg.v[1], rbg.v[2]};
  std::memcpy(residuals, out, sizeof(out));
  return DLIOM_OK;
}

// The state that makes the first nine residuals vanish (biases carried over): the pose / velocity
// prediction the front end feeds to the matchers (the role of PreintegratedImuMeasurements::predict,
// local_trajectory_builder_3d.cc:197).
int dliom_imu_integrator_predict(const dliom_imu_integrator* m, const double si[16], const double gravity[3],
                                 double sj[16]) {
  if (m == nullptr || si == nullptr || gravity == nullptr || sj == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  const Q4 Qi{si[3], si[4], si[5], si[6]};
  const double T = m->sum_dt;
  const V3 Pi{{si[0], si[1], si[2]}}, Vi{{si[7], si[8], si[9]}}, G{{gravity[0], gravity[1], gravity[2]}};
  const V3 Pj = add(sub(add(Pi, scl(T, Vi)), scl(0.5 * T * T, G)), qrot(Qi, m->dp));
  const V3 Vj = add(sub(Vi, scl(T, G)), qrot(Qi, m->dv));
  const Q4 Qj = qnormalized(qmul(Qi, m->dq));
  std::memcpy(sj, Pj.v, 24);
  sj[3] = Qj.w;
  sj[4] = Qj.x;
  sj[5] = Qj.y;
  sj[6] = Qj.z;
  std::memcpy(sj + 7, Vj.v, 24);
  std::memcpy(sj + 10, si + 10, 48);
  return DLIOM_OK;
}

}  // extern "C"
