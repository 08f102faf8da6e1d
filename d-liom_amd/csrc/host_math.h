// Host-side fixed-size math for the candidate generator and the LM driver.
//
// The float paths reproduce, operation for operation, what the reference
// computes through Eigen 3.3 on an x86-64/SSE2 build:
// these values feed the device kernels, and voxel indices must be bit-exact.
//   transform/rigid_transform.h:206-219  Rigid3 composition / point transform
//   transform/transform.h:33-37,85-99    GetAngle, AngleAxisVectorToRotationQuaternion
#ifndef DLIOM_CSRC_HOST_MATH_H_
#define DLIOM_CSRC_HOST_MATH_H_

#include <cmath>

namespace dliom {

struct F3 {
  float x, y, z;
};
struct QF {
  float w, x, y, z;
};
struct PoseF {
  F3 t;
  QF q;
};

// Eigen's fixed-size-3 reduction order: a0 + (a1 + a2).
inline float sqnorm3(const F3& v) { return v.x * v.x + (v.y * v.y + v.z * v.z); }
inline float norm3(const F3& v) { return std::sqrt(sqnorm3(v)); }

// Quaternionf product, SSE lane order of Eigen/src/Geometry/arch/Geometry_SSE.h.
inline QF qmul(const QF& a, const QF& b) {
  QF r;
  r.x = (a.x * b.w - a.z * b.y) + (a.y * b.z + a.w * b.x);
  r.y = (a.y * b.w - a.x * b.z) + (a.z * b.x + a.w * b.y);
  r.z = (a.z * b.w - a.y * b.x) + (a.x * b.y + a.w * b.z);
  r.w = (a.w * b.w - a.x * b.x) - (a.z * b.z + a.y * b.y);
  return r;
}
// coeffs().normalized() with the 4-float packet reduction (x2+z2)+(y2+w2).
inline QF qnormalized(const QF& q) {
  const float z2 = (q.x * q.x + q.z * q.z) + (q.y * q.y + q.w * q.w);
  if (z2 > 0.f) {
    const float n = std::sqrt(z2);
    return QF{q.w / n, q.x / n, q.y / n, q.z / n};
  }
  return q;
}
// QuaternionBase::_transformVector: uv = 2 (u x v); v + w uv + u x uv.
inline F3 qrot(const QF& q, const F3& v) {
  F3 uv{q.y * v.z - q.z * v.y, q.z * v.x - q.x * v.z, q.x * v.y - q.y * v.x};
  uv = F3{uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};
  const F3 c{q.y * uv.z - q.z * uv.y, q.z * uv.x - q.x * uv.z, q.x * uv.y - q.y * uv.x};
  return F3{(v.x + q.w * uv.x) + c.x, (v.y + q.w * uv.y) + c.y, (v.z + q.w * uv.z) + c.z};
}
inline F3 add3(const F3& a, const F3& b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }

// transform.h:85-99 for T = float (sin/cos evaluated in double).
inline QF angle_axis_to_quaternion(const F3& aa) {
  float scale = 0.5f;
  float w = 1.f;
  if (static_cast<double>(sqnorm3(aa)) > 1e-8) {
    const float n = norm3(aa);
    scale = static_cast<float>(std::sin(n / 2.) / n);
    w = static_cast<float>(std::cos(n / 2.));
  }
  return QF{w, scale * aa.x, scale * aa.y, scale * aa.z};
}
// transform.h:33-37 for float.
inline float rotation_angle(const QF& q) {
  const F3 v{q.x, q.y, q.z};
  return 2.f * std::atan2(norm3(v), std::abs(q.w));
}

// ---- double precision rigid transforms (pose bookkeeping around the matchers) ----------------
// transform/rigid_transform.h:125-219 on Eigen::Quaterniond, SSE2 evaluation order of
// Eigen/src/Geometry/arch/Geometry_SSE.h (two Packet2d halves), see SURVEY.md App. A.1.
struct PoseD {
  double t[3];
  double q[4];  // w, x, y, z
};
inline void qmul_d(const double* a, const double* b, double* r) {  // (w,x,y,z)
  const double aw = a[0], ax = a[1], ay = a[2], az = a[3];
  const double bw = b[0], bx = b[1], by = b[2], bz = b[3];
  const double t1x = aw * bx + ay * bz, t1y = aw * by + ay * bw;
  const double t2x = az * bx - ax * bz, t2y = az * by - ax * bw;
  const double u1z = aw * bz - ay * bx, u1w = aw * bw - ay * by;
  const double u2z = az * bz + ax * bx, u2w = az * bw + ax * by;
  r[1] = t1x - t2y;
  r[2] = t1y + t2x;
  r[3] = u1z + u2w;
  r[0] = u1w - u2z;
}
inline void qnormalize_d(double* q) {
  const double z2 = (q[1] * q[1] + q[3] * q[3]) + (q[2] * q[2] + q[0] * q[0]);
  if (z2 > 0.0) {
    const double n = std::sqrt(z2);
    q[0] /= n;
    q[1] /= n;
    q[2] /= n;
    q[3] /= n;
  }
}
inline void qrot_d(const double* q, const double* v, double* out) {
  double uv[3] = {q[2] * v[2] - q[3] * v[1], q[3] * v[0] - q[1] * v[2], q[1] * v[1] - q[2] * v[0]};
  uv[0] += uv[0];
  uv[1] += uv[1];
  uv[2] += uv[2];
  const double c[3] = {q[2] * uv[2] - q[3] * uv[1], q[3] * uv[0] - q[1] * uv[2], q[1] * uv[1] - q[2] * uv[0]};
  out[0] = (v[0] + q[0] * uv[0]) + c[0];
  out[1] = (v[1] + q[0] * uv[1]) + c[1];
  out[2] = (v[2] + q[0] * uv[2]) + c[2];
}
inline PoseD pose_mul(const PoseD& a, const PoseD& b) {  // rigid_transform.h:206-212
  PoseD r;
  double rt[3];
  qrot_d(a.q, b.t, rt);
  for (int i = 0; i < 3; ++i) r.t[i] = rt[i] + a.t[i];
  qmul_d(a.q, b.q, r.q);
  qnormalize_d(r.q);
  return r;
}
inline PoseD pose_inverse(const PoseD& a) {  // rigid_transform.h:167-171
  PoseD r;
  r.q[0] = a.q[0];
  r.q[1] = -a.q[1];
  r.q[2] = -a.q[2];
  r.q[3] = -a.q[3];
  double rt[3];
  qrot_d(r.q, a.t, rt);
  for (int i = 0; i < 3; ++i) r.t[i] = -rt[i];
  return r;
}
inline void pose_to_float7(const PoseD& p, float* out) {  // Rigid3d::cast<float>()
  for (int i = 0; i < 3; ++i) out[i] = static_cast<float>(p.t[i]);
  for (int i = 0; i < 4; ++i) out[3 + i] = static_cast<float>(p.q[i]);
}

}  // namespace dliom

#endif  // DLIOM_CSRC_HOST_MATH_H_
