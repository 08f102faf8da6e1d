// ORACLE (test infrastructure, NOT product code).
//
// CPU restatement of the fixed-size math the reference's scan-to-submap path
// relies on.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may use anything under oracle/.
//
// Follows (paths relative to /root/reference/src/cartographer/cartographer):
//   common/port.h:41-43            RoundToInt = std::lround
//   common/math.h:31-52,74-81      Clamp, Power/Pow2, QuaternionProduct
//   transform/rigid_transform.h:125-219   Rigid3<T>, operator*, inverse
//   transform/transform.h:33-37,85-99     GetAngle, AngleAxisVectorToRotationQuaternion
//
// Eigen (pinned commit f3a22f35b044, bazel/repositories.bzl:139 -- a 3.3-series
// snapshot) is NOT in /root/reference and not installed here.  The scalar
// evaluation orders below restate Eigen 3.3's behaviour on an x86-64/SSE2
// build (the only way the reference is ever built):
//   * Vector3 reductions (squaredNorm/dot): redux_novec_unroller splits
//     [0,3) into [0,1) + [1,3)  ->  x*x + (y*y + z*z)
//   * 4-vector float/double reductions (Quaternion::normalized): packet
//     reduction  ->  (x*x + z*z) + (y*y + w*w)   (coeff order x,y,z,w)
//   * Quaternionf * Quaternionf: Geometry_SSE.h quat_product<SSE,float>
//   * Quaterniond * Quaterniond: Geometry_SSE.h quat_product<SSE,double>
//   * Quaternion * Vector3 (_transformVector): uv = 2*(u x v); v + w*uv + u x uv
//   * generic (Jet) quaternion ops: the plain scalar formulas
// These orders cannot be re-verified offline; by SURVEY.md §8 R1 the oracle's
// order IS the parity definition for the HIP path.
#ifndef ORACLE_OM_MATH_H_
#define ORACLE_OM_MATH_H_

#include <cmath>
#include <cstdint>
#include <type_traits>

namespace oracle {

using uint16 = uint16_t;

// common/port.h:41-43
inline int RoundToInt(const float x) { return static_cast<int>(std::lround(x)); }
inline int RoundToInt(const double x) { return static_cast<int>(std::lround(x)); }

// common/math.h:31-40
template <typename T>
T Clamp(const T value, const T min, const T max) {
  if (value > max) return max;
  if (value < min) return min;
  return value;
}

// common/math.h:43-52 : Power(base, 2) = base * (base * T(1))
template <typename T>
constexpr T Pow2(T a) {
  return a * (a * T(1));
}

template <typename T>
struct Vec3 {
  T x, y, z;
  Vec3() : x(T(0)), y(T(0)), z(T(0)) {}
  Vec3(const T& x_, const T& y_, const T& z_) : x(x_), y(y_), z(z_) {}
  T& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  const T& operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  template <typename U>
  Vec3<U> cast() const {
    return Vec3<U>(U(x), U(y), U(z));
  }
  // Eigen redux order for fixed size 3 (see header comment).
  T squaredNorm() const { return x * x + (y * y + z * z); }
  T norm() const {
    using std::sqrt;
    return sqrt(squaredNorm());
  }
  T dot(const Vec3& o) const { return x * o.x + (y * o.y + z * o.z); }
  Vec3 cross(const Vec3& o) const {
    return Vec3(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x);
  }
};

template <typename T>
Vec3<T> operator+(const Vec3<T>& a, const Vec3<T>& b) {
  return Vec3<T>(a.x + b.x, a.y + b.y, a.z + b.z);
}
template <typename T>
Vec3<T> operator-(const Vec3<T>& a, const Vec3<T>& b) {
  return Vec3<T>(a.x - b.x, a.y - b.y, a.z - b.z);
}
template <typename T>
Vec3<T> operator-(const Vec3<T>& a) {
  return Vec3<T>(-a.x, -a.y, -a.z);
}
template <typename T, typename S>
Vec3<T> operator*(const S& s, const Vec3<T>& a) {
  return Vec3<T>(s * a.x, s * a.y, s * a.z);
}

using Vec3f = Vec3<float>;
using Vec3d = Vec3<double>;

struct Vec3i {
  int x, y, z;
  Vec3i() : x(0), y(0), z(0) {}
  Vec3i(int x_, int y_, int z_) : x(x_), y(y_), z(z_) {}
  int& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  const int& operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  bool operator==(const Vec3i& o) const { return x == o.x && y == o.y && z == o.z; }
};
inline Vec3i operator+(const Vec3i& a, const Vec3i& b) {
  return Vec3i(a.x + b.x, a.y + b.y, a.z + b.z);
}
inline Vec3i operator-(const Vec3i& a, const Vec3i& b) {
  return Vec3i(a.x - b.x, a.y - b.y, a.z - b.z);
}

namespace detail {
template <typename T>
struct QuatOps {
  // Generic scalar formula (Eigen Quaternion.h quat_product<Arch,..,Scalar>).
  static void Product(const T* a, const T* b, T* r) {  // (w,x,y,z)
    r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    r[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    r[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    r[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
  }
  static T SquaredNorm(const T* q) {  // generic 4-redux: (x2+y2)+(z2+w2)
    return (q[1] * q[1] + q[2] * q[2]) + (q[3] * q[3] + q[0] * q[0]);
  }
};
template <>
struct QuatOps<float> {
  // Geometry_SSE.h quat_product<Architecture::SSE, ., ., float>, lane by lane.
  static void Product(const float* a, const float* b, float* r) {
    const float aw = a[0], ax = a[1], ay = a[2], az = a[3];
    const float bw = b[0], bx = b[1], by = b[2], bz = b[3];
    r[1] = (ax * bw - az * by) + (ay * bz + aw * bx);
    r[2] = (ay * bw - ax * bz) + (az * bx + aw * by);
    r[3] = (az * bw - ay * bx) + (ax * by + aw * bz);
    r[0] = (aw * bw - ax * bx) - (az * bz + ay * by);
  }
  static float SquaredNorm(const float* q) {  // predux<Packet4f>
    return (q[1] * q[1] + q[3] * q[3]) + (q[2] * q[2] + q[0] * q[0]);
  }
};
template <>
struct QuatOps<double> {
  // Geometry_SSE.h quat_product<Architecture::SSE, ., ., double>.
  static void Product(const double* a, const double* b, double* r) {
    const double aw = a[0], ax = a[1], ay = a[2], az = a[3];
    const double bw = b[0], bx = b[1], by = b[2], bz = b[3];
    // t1 = ww*xy + yy*zw ; t2 = zz*xy - xx*zw ; res.xy = t1 +/- swap(t2)
    const double t1x = aw * bx + ay * bz, t1y = aw * by + ay * bw;
    const double t2x = az * bx - ax * bz, t2y = az * by - ax * bw;
    r[1] = t1x - t2y;
    r[2] = t1y + t2x;
    // t1 = ww*zw - yy*xy ; t2 = zz*zw + xx*xy ; res.zw = t1 -/+ swap(t2)
    const double u1z = aw * bz - ay * bx, u1w = aw * bw - ay * by;
    const double u2z = az * bz + ax * bx, u2w = az * bw + ax * by;
    r[3] = u1z + u2w;
    r[0] = u1w - u2z;
  }
  static double SquaredNorm(const double* q) {  // 2 x Packet2d, then predux
    return (q[1] * q[1] + q[3] * q[3]) + (q[2] * q[2] + q[0] * q[0]);
  }
};
}  // namespace detail

template <typename T>
struct Quat {
  T w, x, y, z;
  Quat() : w(T(1)), x(T(0)), y(T(0)), z(T(0)) {}
  Quat(const T& w_, const T& x_, const T& y_, const T& z_)
      : w(w_), x(x_), y(y_), z(z_) {}
  template <typename U>
  Quat<U> cast() const {
    return Quat<U>(U(w), U(x), U(y), U(z));
  }
  Vec3<T> vec() const { return Vec3<T>(x, y, z); }
  Quat conjugate() const { return Quat(w, -x, -y, -z); }
  T squaredNorm() const {
    const T q[4] = {w, x, y, z};
    return detail::QuatOps<T>::SquaredNorm(q);
  }
  T norm() const {
    using std::sqrt;
    return sqrt(squaredNorm());
  }
  // MatrixBase::normalized(): n / sqrt(z) if z > 0 else n.
  Quat normalized() const {
    using std::sqrt;
    const T z2 = squaredNorm();
    if (z2 > T(0)) {
      const T n = sqrt(z2);
      return Quat(w / n, x / n, y / n, z / n);
    }
    return *this;
  }
  // QuaternionBase::_transformVector
  Vec3<T> operator*(const Vec3<T>& v) const {
    Vec3<T> uv = vec().cross(v);
    uv = uv + uv;
    return (v + w * uv) + vec().cross(uv);
  }
  Quat operator*(const Quat& o) const {
    const T a[4] = {w, x, y, z};
    const T b[4] = {o.w, o.x, o.y, o.z};
    T r[4];
    detail::QuatOps<T>::Product(a, b, r);
    return Quat(r[0], r[1], r[2], r[3]);
  }
};
using Quatf = Quat<float>;
using Quatd = Quat<double>;

// transform/rigid_transform.h:125-219
template <typename T>
struct Rigid3 {
  Vec3<T> translation;
  Quat<T> rotation;
  Rigid3() {}
  Rigid3(const Vec3<T>& t, const Quat<T>& q) : translation(t), rotation(q) {}
  static Rigid3 Translation(const Vec3<T>& t) { return Rigid3(t, Quat<T>()); }
  static Rigid3 Rotation(const Quat<T>& q) { return Rigid3(Vec3<T>(), q); }
  template <typename U>
  Rigid3<U> cast() const {
    return Rigid3<U>(translation.template cast<U>(), rotation.template cast<U>());
  }
  // rigid_transform.h:167-171
  Rigid3 inverse() const {
    const Quat<T> r = rotation.conjugate();
    const Vec3<T> t = -(r * translation);
    return Rigid3(t, r);
  }
};
// rigid_transform.h:206-212
template <typename T>
Rigid3<T> operator*(const Rigid3<T>& lhs, const Rigid3<T>& rhs) {
  return Rigid3<T>(lhs.rotation * rhs.translation + lhs.translation,
                   (lhs.rotation * rhs.rotation).normalized());
}
// rigid_transform.h:214-219
template <typename T>
Vec3<T> operator*(const Rigid3<T>& rigid, const Vec3<T>& point) {
  return rigid.rotation * point + rigid.translation;
}
using Rigid3f = Rigid3<float>;
using Rigid3d = Rigid3<double>;

// transform/transform.h:33-37
template <typename T>
T GetAngle(const Rigid3<T>& transform) {
  return T(2) * std::atan2(transform.rotation.vec().norm(),
                           std::abs(transform.rotation.w));
}

// transform/transform.h:85-99.  For T=float the sin/cos are evaluated in
// double (norm / 2. promotes) and rounded back to float on assignment.
template <typename T>
Quat<T> AngleAxisVectorToRotationQuaternion(const Vec3<T>& angle_axis) {
  T scale = T(0.5);
  T w = T(1.);
  constexpr double kCutoffAngle = 1e-8;  // compared against the SQUARED norm
  if (angle_axis.squaredNorm() > kCutoffAngle) {
    const T norm = angle_axis.norm();
    scale = T(std::sin(norm / 2.) / norm);
    w = T(std::cos(norm / 2.));
  }
  const Vec3<T> q_xyz = scale * angle_axis;
  return Quat<T>(w, q_xyz.x, q_xyz.y, q_xyz.z);
}

// Eigen AngleAxis -> Quaternion (used only to build test poses):
// q.w = cos(a/2), q.vec = sin(a/2) * axis
inline Quatd AngleAxisToQuat(double angle, const Vec3d& unit_axis) {
  const double s = std::sin(0.5 * angle);
  return Quatd(std::cos(0.5 * angle), s * unit_axis.x, s * unit_axis.y,
               s * unit_axis.z);
}

}  // namespace oracle

#endif  // ORACLE_OM_MATH_H_
