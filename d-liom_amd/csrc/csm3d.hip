// CeresScanMatcher3D on gfx950.
//
// Replaces mapping/internal/3d/scan_matching/ceres_scan_matcher_3d.cc:71-123 together with the
// residual functors it stacks:
//   occupied_space_cost_function_3d.h:50-80, interpolated_grid.h:51-146 (device kernel below),
//   translation_delta_cost_functor_3d.h:39-45, rotation_delta_cost_functor_3d.h:43-66 (6 scalar
//   residuals, evaluated on the host), optimization/ceres_pose.cc:30-44 (t[3], q[4] = w,x,y,z).
//
// The reference hands Ceres a (N_hi+N_lo+6) x 6 Jacobian and lets DENSE_QR factor it.  Here one
// kernel evaluates every occupied-space residual together with its ANALYTIC Jacobian row and
// reduces straight into the 6x6 normal equations (21 unique J^T J entries + 6 J^T r + cost):
// the Jacobian never exists in memory.  The stacked J is tall-skinny N x 6 -> 21 length-N dot
// products; that is wave-reduction work, not an MFMA-shaped GEMM.
// The Levenberg-Marquardt outer loop restates Ceres 1.13's trust-region minimizer
// (jacobi scaling, radius update, tolerances) on those 6x6 systems.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "device_common.h"

namespace dliom {

constexpr int kCsmBlock = 256;
constexpr int kAcc = 28;  // 21 JtJ (upper triangle, row-major) + 6 Jtr + 1 sum r^2
static_assert(kAcc % 4 == 0 && kCsmBlock == 256, "the block reductions fold four waves x kAcc / 4 sums");

struct CsmCloudArg {
  GridView g;
  const float* x;
  const float* y;
  const float* z;
  int n;
  double scale;  // occupied_space_weight / sqrt(n)
};
struct CsmPose {
  double t[3];
  double q[4];      // w,x,y,z (not normalised, as in the reference's Jet evaluation)
  double plus[12];  // d q / d local, 4 x nloc row-major (QuaternionParameterization::ComputeJacobian)
  int nloc;         // 3, or 1 for the yaw-only parameterisation
};
struct CsmArgs {
  CsmCloudArg cloud[DLIOM_MAX_CLOUDS];
  int num_clouds;
  int total_points;
  CsmPose pose;
};

__device__ __forceinline__ double lut_probability(unsigned v, float k_scale, float k_offset,
                                                  float k_unknown) {
  v &= 0x7FFFu;
  const float p = v == 0u ? k_unknown : static_cast<float>(static_cast<int>(v)) * k_scale + k_offset;
  return static_cast<double>(p);
}

// The eight voxels around (ix, iy, iz): ix / ix + 1 etc.  The index arithmetic per axis is done once (two values each)
// and combined with ORs -- eight calls of grid_value() were eight times the whole sequence, with 64-bit
// multiply-adds (quarter rate) in it; csm_lm_kernel is bound by instruction issue.  32-bit offsets: bits <= 7
// (table and pool below 4 GiB); bits = 8 grids take grid_value().
__device__ __forceinline__ void grid_corner_values(const GridView& g, int ix, int iy, int iz, unsigned (&v)[8]) {
  if (g.dense != nullptr) {  // uniform
    // The dense mirror of the matcher (grid.hip: the same values with the update marker stripped and 1 for "unknown")
    // where it covers the eight cells: ONE load per voxel instead of leaf table -> leaf, and the eight of them in one or
    // two 128-byte bricks.  "Unknown" reads 1 there instead of 0; lut_probability() gives both the same float (0.1f:
    // kMinProbability is what value 1 stands for, and 1 * k_scale + k_offset rounds to it -- checked on the host where
    // the constants are made, setup_problem()).
    const int mx = ix + g.dense_off[0], my = iy + g.dense_off[1], mz = iz + g.dense_off[2];
    const int S = g.dense_stride;
    if (mx >= 0 && my >= 0 && mz >= 0 && mx + 1 < S && my + 1 < S && mz + 1 < S) {
      const unsigned B = static_cast<unsigned>(g.dense_bricks);
      unsigned bx[2], by[2], bz[2], cx[2], cy[2], cz[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const unsigned ux = static_cast<unsigned>(mx + k), uy = static_cast<unsigned>(my + k), uz = static_cast<unsigned>(mz + k);
        bx[k] = (ux >> 2) << 7;
        by[k] = ((uy >> 2) * B) << 7;
        bz[k] = ((uz >> 2) * B * B) << 7;
        cx[k] = (ux & 3u) << 1;
        cy[k] = (uy & 3u) << 3;
        cz[k] = (uz & 3u) << 5;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {  // k = x << 2 | y << 1 | z
        const int kx = k >> 2, ky = (k >> 1) & 1, kz = k & 1;
        v[k] = *reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(g.dense) + ((bz[kz] + by[ky] + bx[kx]) | cz[kz] | cy[ky] | cx[kx]));
      }
      return;
    }
  }
  if (g.log2_leaves > 10) {  // uniform
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = grid_value(g, ix + (k >> 2), iy + ((k >> 1) & 1), iz + (k & 1));
    return;
  }
  const unsigned lb = static_cast<unsigned>(g.log2_leaves);
  unsigned tx[2], ty[2], tz[2], cx[2], cy[2], cz[2];
  bool in_x[2], in_y[2], in_z[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const unsigned sx = static_cast<unsigned>(ix + k + g.half), sy = static_cast<unsigned>(iy + k + g.half),
                   sz = static_cast<unsigned>(iz + k + g.half);
    in_x[k] = sx < g.grid_size;
    in_y[k] = sy < g.grid_size;
    in_z[k] = sz < g.grid_size;
    tx[k] = sx >> 3;
    ty[k] = (sy >> 3) << lb;
    tz[k] = (sz >> 3) << (2u * lb);
    cx[k] = (sx & 7u) << 1;
    cy[k] = (sy & 7u) << 4;
    cz[k] = (sz & 7u) << 7;
  }
  unsigned slot[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {  // k = x << 2 | y << 1 | z
    const int kx = k >> 2, ky = (k >> 1) & 1, kz = k & 1;
    const bool inside = in_x[kx] && in_y[ky] && in_z[kz];
    const unsigned tidx = inside ? (tz[kz] | ty[ky] | tx[kx]) : 0u;
    slot[k] = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(g.table) + (tidx << 2));
    slot[k] = inside ? slot[k] : 0u;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int kx = k >> 2, ky = (k >> 1) & 1, kz = k & 1;
    v[k] = *reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(g.pool) + ((slot[k] << 10) | cz[kz] | cy[ky] | cx[kx]));
  }
}

// 1 / d for d near the grid resolution: a float seed (IEEE float division) and two Newton steps -- a double division
// is a dependent sequence of ~35 instructions, three of them per point.  (Within an ulp of the correctly rounded
// reciprocal; the cost functor's parity bar is 1e-9.)
__device__ __forceinline__ double fast_reciprocal(double d) {
  double y = static_cast<double>(1.0f / static_cast<float>(d));
  y = fma(y, fma(-d, y, 1.0), y);
  y = fma(y, fma(-d, y, 1.0), y);
  return y;
}

// One point: residual r = s (1 - P(T p)) and its 6 tangent-space derivatives.
// The double-precision arithmetic is written in explicit fused multiply-adds (round 4: a third fewer instructions --
// csm_lm_kernel is bound by instruction issue; the values differ from the reference's separately rounded operations in
// the last bits, far inside the 1e-9 bar against the oracle's Jets).  Explicit, not `#pragma clang fp contract`: which
// products a contracting compiler fuses depends on what the function is inlined into, and the one-launch kernel, the
// evaluation kernel and the grid-barrier kernel must produce the same bits.  The float block in between -- which cell,
// which side of its centre: interpolated_grid.h:123-139 -- is the reference's arithmetic to the bit.
__device__ __forceinline__ double cross_term(double a, double b, double c, double d) { return fma(a, b, -(c * d)); }  // a b - c d
// interpolated_grid.h:88-102, one axis: (a - b) n^3 2 + (b - a) n^2 3 + a, with c3 = 2 n^3 and c2 = 3 n^2
__device__ __forceinline__ double smooth_mix(double a, double b, double c3, double c2) { return fma(a - b, c3, fma(b - a, c2, a)); }
// ... and its derivative by n: (a - b) n^2 6 + (b - a) n 6, with e2 = 6 n^2 and e1 = 6 n
__device__ __forceinline__ double smooth_mix_dn(double a, double b, double e2, double e1) { return fma(a - b, e2, (b - a) * e1); }

__device__ __forceinline__ void csm_point_v(const CsmPose& a, const CsmCloudArg& c, double vx, double vy, double vz,
                                            float k_scale, float k_offset, float k_unknown,
                                            double* r_out, double jrow[6]) {
  const double qw = a.q[0], ux = a.q[1], uy = a.q[2], uz = a.q[3];
  // Eigen _transformVector on doubles: uv = 2 (u x v); world = (v + w uv) + u x uv, then + t
  double uvx = cross_term(uy, vz, uz, vy), uvy = cross_term(uz, vx, ux, vz), uvz = cross_term(ux, vy, uy, vx);
  uvx = uvx + uvx;
  uvy = uvy + uvy;
  uvz = uvz + uvz;
  const double wx = (fma(qw, uvx, vx) + cross_term(uy, uvz, uz, uvy)) + a.t[0];
  const double wy = (fma(qw, uvy, vy) + cross_term(uz, uvx, ux, uvz)) + a.t[1];
  const double wz = (fma(qw, uvz, vz) + cross_term(ux, uvy, uy, uvx)) + a.t[2];

  // interpolated_grid.h:123-139: cell of the point (double -> float), centre in float, step down
  // where the float centre exceeds the double coordinate; x2 = x1 + resolution in float.
  const float res = c.g.resolution;
  const int ix0 = cell_of(static_cast<float>(wx), res);
  const int iy0 = cell_of(static_cast<float>(wy), res);
  const int iz0 = cell_of(static_cast<float>(wz), res);
  float lx = static_cast<float>(ix0) * res, ly = static_cast<float>(iy0) * res,
        lz = static_cast<float>(iz0) * res;
  if (static_cast<double>(lx) > wx) lx -= res;
  if (static_cast<double>(ly) > wy) ly -= res;
  if (static_cast<double>(lz) > wz) lz -= res;
  const double x1 = lx, y1 = ly, z1 = lz;
  const double x2 = static_cast<double>(lx + res), y2 = static_cast<double>(ly + res),
               z2 = static_cast<double>(lz + res);
  const int ix = cell_of(lx, res), iy = cell_of(ly, res), iz = cell_of(lz, res);
  unsigned corner[8];
  grid_corner_values(c.g, ix, iy, iz, corner);
  const double q111 = lut_probability(corner[0], k_scale, k_offset, k_unknown);
  const double q112 = lut_probability(corner[1], k_scale, k_offset, k_unknown);
  const double q121 = lut_probability(corner[2], k_scale, k_offset, k_unknown);
  const double q122 = lut_probability(corner[3], k_scale, k_offset, k_unknown);
  const double q211 = lut_probability(corner[4], k_scale, k_offset, k_unknown);
  const double q212 = lut_probability(corner[5], k_scale, k_offset, k_unknown);
  const double q221 = lut_probability(corner[6], k_scale, k_offset, k_unknown);
  const double q222 = lut_probability(corner[7], k_scale, k_offset, k_unknown);

  // Jet / scalar multiplies by the reciprocal (ceres/jet.h operator/(Jet, T)).
  const double inv_dx = fast_reciprocal(x2 - x1), inv_dy = fast_reciprocal(y2 - y1), inv_dz = fast_reciprocal(z2 - z1);
  const double nx = (wx - x1) * inv_dx, ny = (wy - y1) * inv_dy, nz = (wz - z1) * inv_dz;
  const double nxx = nx * nx, nyy = ny * ny, nzz = nz * nz;
  const double x3 = 2. * (nx * nxx), x2c = 3. * nxx, xe2 = 6. * nxx, xe1 = 6. * nx;
  const double y3 = 2. * (ny * nyy), y2c = 3. * nyy, ye2 = 6. * nyy, ye1 = 6. * ny;
  const double z3 = 2. * (nz * nzz), z2c = 3. * nzz, ze2 = 6. * nzz, ze1 = 6. * nz;
  // interpolated_grid.h:88-102, z then y then x; d(.)/dn alongside.
  const double q11 = smooth_mix(q111, q112, z3, z2c), q12 = smooth_mix(q121, q122, z3, z2c);
  const double q21 = smooth_mix(q211, q212, z3, z2c), q22 = smooth_mix(q221, q222, z3, z2c);
  const double d11 = smooth_mix_dn(q111, q112, ze2, ze1), d12 = smooth_mix_dn(q121, q122, ze2, ze1);
  const double d21 = smooth_mix_dn(q211, q212, ze2, ze1), d22 = smooth_mix_dn(q221, q222, ze2, ze1);
  const double q1 = smooth_mix(q11, q12, y3, y2c), q2 = smooth_mix(q21, q22, y3, y2c);
  const double q1_z = smooth_mix(d11, d12, y3, y2c), q2_z = smooth_mix(d21, d22, y3, y2c);
  const double q1_y = smooth_mix_dn(q11, q12, ye2, ye1), q2_y = smooth_mix_dn(q21, q22, ye2, ye1);
  const double P = smooth_mix(q1, q2, x3, x2c);
  const double P_nx = smooth_mix_dn(q1, q2, xe2, xe1);
  const double P_ny = smooth_mix(q1_y, q2_y, x3, x2c);
  const double P_nz = smooth_mix(q1_z, q2_z, x3, x2c);
  const double gx = P_nx * inv_dx, gy = P_ny * inv_dy, gz = P_nz * inv_dz;  // dP/dworld

  const double s = c.scale;
  *r_out = s * (1. - P);
  // dr/dworld = -s * grad P ; dworld/dt = I
  const double ax = -s * gx, ay = -s * gy, az = -s * gz;
  jrow[0] = ax;
  jrow[1] = ay;
  jrow[2] = az;
  // dworld/dq for f(w,u) = v + 2 w (u x v) + 2 u x (u x v):
  //   d/dw   = 2 (u x v) = uv
  //   d/du_k = 2 w (e_k x v) + 2 e_k x (u x v) + 2 u x (e_k x v)
  const double hx = 0.5 * uvx, hy = 0.5 * uvy, hz = 0.5 * uvz;  // u x v (exact halving)
  auto dot3 = [](double p0, double p1, double p2, double q0, double q1v, double q2v) { return fma(p2, q2v, fma(p1, q1v, p0 * q0)); };
  double dq[4];
  dq[0] = dot3(ax, ay, az, uvx, uvy, uvz);
  {
    // e_x x v = (0, -vz, vy); e_x x h = (0, -hz, hy); u x (e_x x v) = (uy vy + uz vz, -ux vy, -ux vz)
    const double gxv = fma(uy, vy, uz * vz), gyv = -(ux * vy), gzv = -(ux * vz);
    const double dxw = 2. * gxv, dyw = 2. * (fma(qw, -vz, -hz) + gyv), dzw = 2. * (fma(qw, vy, hy) + gzv);
    dq[1] = dot3(ax, ay, az, dxw, dyw, dzw);
  }
  {
    // e_y x v = (vz, 0, -vx); e_y x h = (hz, 0, -hx); u x (e_y x v) = (-uy vx, ux vx + uz vz, -uy vz)
    const double gxv = -(uy * vx), gyv = fma(ux, vx, uz * vz), gzv = -(uy * vz);
    const double dxw = 2. * (fma(qw, vz, hz) + gxv), dyw = 2. * gyv, dzw = 2. * (fma(qw, -vx, -hx) + gzv);
    dq[2] = dot3(ax, ay, az, dxw, dyw, dzw);
  }
  {
    // e_z x v = (-vy, vx, 0); e_z x h = (-hy, hx, 0); u x (e_z x v) = (-uz vx, -uz vy, ux vx + uy vy)
    const double gxv = -(uz * vx), gyv = -(uz * vy), gzv = fma(ux, vx, uy * vy);
    const double dxw = 2. * (fma(qw, -vy, -hy) + gxv), dyw = 2. * (fma(qw, vx, hx) + gyv), dzw = 2. * gzv;
    dq[3] = dot3(ax, ay, az, dxw, dyw, dzw);
  }
  // tangent space: J_local = J_ambient(1x4) * plus(4 x nloc)
  jrow[3] = jrow[4] = jrow[5] = 0.;
  for (int c2 = 0; c2 < a.nloc; ++c2) {
    double acc = 0.;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc = fma(dq[k], a.plus[k * a.nloc + c2], acc);
    jrow[3 + c2] = acc;
  }
}

__device__ __forceinline__ void csm_point(const CsmPose& a, const CsmCloudArg& c, int i,
                                          float k_scale, float k_offset, float k_unknown,
                                          double* r_out, double jrow[6]) {
  csm_point_v(a, c, static_cast<double>(c.x[i]), static_cast<double>(c.y[i]), static_cast<double>(c.z[i]), k_scale,
              k_offset, k_unknown, r_out, jrow);
}

// ---- the workgroup's 28 sums: a reduce-scatter over the lanes, then the four waves' results (round 4) ----------------
// Until round 4 all 28 x 256 partial sums went to LDS (57 KB), wave w folded sums w, w + 4, ... with a six-step butterfly
// each (84 cross-lane exchanges per lane), three barriers: 3 700 cycles per evaluation of csm_lm_kernel.  Now every wave
// halves its 28 (padded to 32) values per step -- lanes l and l ^ 32 exchange 16 of them and keep the sums of the other
// 16, then l ^ 16 with 8, ... -- so that after five steps lane l holds ONE quantity (number l >> 1) summed over half the
// wave and the sixth step completes it: 32 additions and 32 exchanges instead of 168 and 84, v_permlane32/16_swap and
// DPP instead of ds_bpermute.  The waves' 4 x 28 results meet in LDS (2 KB, double-buffered: ONE barrier per call) and
// every thread adds them.  The order of the additions is fixed, and the same in csm_eval_kernel, csm_lm_kernel and
// csm_lm_grid_kernel: they still produce the same bits (tested).
struct Halves {
  unsigned lo, hi;
};
__device__ __forceinline__ Halves split_double(double v) {
  const unsigned long long u = static_cast<unsigned long long>(__double_as_longlong(v));
  return Halves{static_cast<unsigned>(u), static_cast<unsigned>(u >> 32)};
}
__device__ __forceinline__ double join_double(unsigned lo, unsigned hi) {
  return __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo));
}
// lanes 32..63 of x <-> lanes 0..31 of y (v_permlane32_swap): afterwards x + y is, in lanes 0..31, x's pair sum (lane l
// and l + 32) and in lanes 32..63 y's
__device__ __forceinline__ void swap_halves32(double& x, double& y) {
  const Halves a = split_double(x), b = split_double(y);
  const auto lo = __builtin_amdgcn_permlane32_swap(a.lo, b.lo, false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(a.hi, b.hi, false, false);
  x = join_double(lo[0], hi[0]);
  y = join_double(lo[1], hi[1]);
}
// odd rows of x <-> even rows of y (rows of 16 lanes; v_permlane16_swap): x + y is x's pair sum (lane l and l ^ 16) in
// the even rows and y's in the odd rows
__device__ __forceinline__ void swap_rows16(double& x, double& y) {
  const Halves a = split_double(x), b = split_double(y);
  const auto lo = __builtin_amdgcn_permlane16_swap(a.lo, b.lo, false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(a.hi, b.hi, false, false);
  x = join_double(lo[0], hi[0]);
  y = join_double(lo[1], hi[1]);
}
template <int kDppCtrl>
__device__ __forceinline__ double dpp_move_double(double v) {
  const Halves h = split_double(v);
  const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(h.lo), kDppCtrl, 0xf, 0xf, false));
  const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(h.hi), kDppCtrl, 0xf, 0xf, false));
  return join_double(lo, hi);
}
__device__ __forceinline__ double from_lane_xor8(double v) { return dpp_move_double<0x128>(v); }  // row_ror:8
__device__ __forceinline__ double from_lane_xor2(double v) { return dpp_move_double<0x4E>(v); }   // quad_perm:[2,3,0,1]
__device__ __forceinline__ double from_lane_xor1(double v) { return dpp_move_double<0xB1>(v); }   // quad_perm:[1,0,3,2]
__device__ __forceinline__ double from_lane_xor4(double v) {                                      // ds_swizzle, xor mask 4
  const Halves h = split_double(v);
  return join_double(static_cast<unsigned>(__builtin_amdgcn_ds_swizzle(static_cast<int>(h.lo), 0x101F)),
                     static_cast<unsigned>(__builtin_amdgcn_ds_swizzle(static_cast<int>(h.hi), 0x101F)));
}
// keeps a (lanes whose `bit` is clear) or b (set), adds the partner's contribution to the kept one
template <class Exchange>
__device__ __forceinline__ double keep_and_add(double a, double b, bool bit_set, Exchange from_partner) {
  const double send = bit_set ? a : b, keep = bit_set ? b : a;
  return keep + from_partner(send);
}
// part: [2][kCsmBlock / 64][32] doubles of LDS; parity: 0 / 1 alternating from call to call (0 for a single call)
__device__ __forceinline__ void block_reduce28(const double (&acc)[kAcc], double* part, int parity, double (&sums)[kAcc]) {
  static_assert(kAcc == 28 && kCsmBlock == 256, "written for 28 sums and four waves");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double r1[16], r2[8], r3[4], r4[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    double a = acc[i], b = i + 16 < kAcc ? acc[i + 16] : 0.;
    swap_halves32(a, b);
    r1[i] = a + b;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    double a = r1[i], b = r1[i + 8];
    swap_rows16(a, b);
    r2[i] = a + b;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) r3[i] = keep_and_add(r2[i], r2[i + 4], (lane & 8) != 0, from_lane_xor8);
#pragma unroll
  for (int i = 0; i < 2; ++i) r4[i] = keep_and_add(r3[i], r3[i + 2], (lane & 4) != 0, from_lane_xor4);
  const double r5 = keep_and_add(r4[0], r4[1], (lane & 2) != 0, from_lane_xor2);
  const double total = r5 + from_lane_xor1(r5);  // quantity number lane >> 1, over the wave's 64 lanes
  double* mine = part + (parity & 1) * (kCsmBlock / 64) * 32;
  if ((lane & 1) == 0) mine[wave * 32 + (lane >> 1)] = total;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kAcc; ++k) sums[k] = (mine[k] + mine[32 + k]) + (mine[64 + k] + mine[96 + k]);
}

// Every thread strides over the stacked clouds, accumulates its 28 sums in registers, then the
// block reduces them in a FIXED order (block_reduce28; deterministic run to run).
__global__ __launch_bounds__(kCsmBlock) void csm_eval_kernel(CsmArgs a, float k_scale, float k_offset,
                                                             float k_unknown,
                                                             double* __restrict__ partials,
                                                             double* __restrict__ final_out, unsigned* done_word,
                                                             unsigned done_seq) {
  double acc[kAcc];
#pragma unroll
  for (int k = 0; k < kAcc; ++k) acc[k] = 0.;
  const int stride = gridDim.x * blockDim.x;
  for (int ci = 0; ci < a.num_clouds; ++ci) {  // uniform: the cloud descriptor stays in SGPRs
    const CsmCloudArg& c = a.cloud[ci];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < c.n; i += stride) {
      double r, j[6];
      csm_point(a.pose, c, i, k_scale, k_offset, k_unknown, &r, j);
      int idx = 0;
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int q = p; q < 6; ++q) acc[idx++] += j[p] * j[q];
#pragma unroll
      for (int p = 0; p < 6; ++p) acc[21 + p] += j[p] * r;
      acc[27] += r * r;
    }
  }
  __shared__ double part[2 * (kCsmBlock / 64) * 32];
  double sums[kAcc];
  block_reduce28(acc, part, 0, sums);
  if (threadIdx.x < kAcc) {
    double mine = sums[0];
#pragma unroll
    for (int k = 1; k < kAcc; ++k)
      if (static_cast<int>(threadIdx.x) == k) mine = sums[k];
    partials[blockIdx.x * kAcc + threadIdx.x] = mine;
    if (gridDim.x == 1) final_out[threadIdx.x] = mine;  // small problems: this block's sums ARE the result
  }
  if (gridDim.x == 1 && done_word != nullptr) {  // completion word for the host's poll (internal.h, wait_done)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence_system();
      *reinterpret_cast<volatile unsigned*>(done_word) = done_seq;
    }
  }
}

// One wave per accumulated quantity: lane l sums blocks l, l+64, ... then a fixed butterfly --
// the summation order never changes from run to run.
__global__ void csm_final_reduce_kernel(const double* __restrict__ partials, int num_blocks,
                                        double* __restrict__ out, unsigned* arrivals, unsigned* done_word, unsigned done_seq) {
  const int k = blockIdx.x;
  const int lane = threadIdx.x;
  double s = 0.;
  for (int b = lane; b < num_blocks; b += 64) s += partials[b * kAcc + k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) {
    out[k] = s;
    if (done_word != nullptr) {  // the last of the kAcc waves to get here tells the host (and re-arms the counter)
      __threadfence_system();
      if (atomicAdd(arrivals, 1u) == static_cast<unsigned>(kAcc) - 1u) {
        *arrivals = 0u;
        __threadfence_system();
        *reinterpret_cast<volatile unsigned*>(done_word) = done_seq;
      }
    }
  }
}

// ---------------------------------------------------------------------------------- host side
// index of (r, c) = (c, r) in the upper triangle of a symmetric 6 x 6 matrix stored row by row (0 .. 20: the order of the
// 21 J^T J sums of the evaluation kernels)
__host__ __device__ constexpr int tri(int r, int c) { return r <= c ? r * 6 - r * (r - 1) / 2 + (c - r) : c * 6 - c * (c - 1) / 2 + (r - c); }
struct Normal {
  double H[21];  // J^T J, upper triangle row-major (tri(r, c)): the trust-region loop of csm_lm_kernel keeps two of these
                 // and a scaled copy in registers, and the full 6 x 6 form did not fit the 256 it can address directly
  double g[6];   // J^T r
  double cost;   // 1/2 sum r^2
};

struct CsmProblem {
  dliom_ctx* ctx;
  const dliom_csm_options* o;
  CsmArgs args;  // clouds, scales; t/q/plus filled per evaluation
  double target_t[3];
  double init_q[4];
  int nloc;
  int num_blocks;
  double* d_partials;
  double* d_out;
  unsigned* d_arrivals;  // zeroed word: csm_final_reduce_kernel's last-wave detection (null: full synchronise)
  int evaluations;
};

__host__ __device__ static void quat_product(const double z[4], const double w[4], double zw[4]) {
  zw[0] = z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3];
  zw[1] = z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2];
  zw[2] = z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1];
  zw[3] = z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0];
}

// QuaternionParameterization / YawOnlyQuaternionPlus (rotation_parameterization.h:27-39)
__host__ __device__ static void plus_jacobian(const double q[4], int nloc, double j[12]) {
  if (nloc == 3) {
    j[0] = -q[1]; j[1] = -q[2]; j[2] = -q[3];
    j[3] = q[0];  j[4] = q[3];  j[5] = -q[2];
    j[6] = -q[3]; j[7] = q[0];  j[8] = q[1];
    j[9] = q[2];  j[10] = -q[1]; j[11] = q[0];
  } else {
    j[0] = -q[3];
    j[1] = -q[2];
    j[2] = q[1];
    j[3] = q[0];
  }
}
__host__ __device__ static void plus(const double x[7], const double* delta, int nloc, double out[7]) {
  for (int i = 0; i < 3; ++i) out[i] = x[i] + delta[i];
  const double* d = delta + 3;
  if (nloc == 3) {
    const double n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    if (n2 > 0.0) {
      // q_delta = [cos |d|, sin |d| / |d| d] (QuaternionParameterization::Plus).  An LM step turns by a fraction of a
      // degree: both factors are series in |d|^2 there (to 1e-17 below |d| = 0.25) -- no square root, no division, no
      // sin / cos, which were a dependent chain of ~2 000 cycles on the critical path of every iteration of
      // csm_lm_kernel, twice (candidate and gradient projection).  Host and device run this same code.
      double sn, cs;
      if (n2 < 0.0625) {
        sn = 1.0 + n2 * (-1.0 / 6.0 + n2 * (1.0 / 120.0 + n2 * (-1.0 / 5040.0 + n2 * (1.0 / 362880.0 + n2 * (-1.0 / 39916800.0 +
             n2 * (1.0 / 6227020800.0 + n2 * (-1.0 / 1307674368000.0 + n2 * (1.0 / 355687428096000.0))))))));
        cs = 1.0 + n2 * (-0.5 + n2 * (1.0 / 24.0 + n2 * (-1.0 / 720.0 + n2 * (1.0 / 40320.0 + n2 * (-1.0 / 3628800.0 +
             n2 * (1.0 / 479001600.0 + n2 * (-1.0 / 87178291200.0 + n2 * (1.0 / 20922789888000.0))))))));
      } else {
        const double n = sqrt(n2);
        sn = sin(n) / n;
        cs = cos(n);
      }
      const double qd[4] = {cs, sn * d[0], sn * d[1], sn * d[2]};
      quat_product(qd, x + 3, out + 3);
    } else {
      for (int i = 0; i < 4; ++i) out[3 + i] = x[3 + i];
    }
  } else {
    double c = d[0];
    if (c > 0.5) c = 0.5;
    if (c < -0.5) c = -0.5;
    const double qd[4] = {sqrt(1. - c * c), 0., 0., c};
    quat_product(qd, x + 3, out + 3);
  }
}

// The 28 occupied-space sums -> full normal equations, plus the two prior residual blocks
// (translation_delta_cost_functor_3d.h:39-45, rotation_delta_cost_functor_3d.h:43-54).
template <int nloc>
__host__ __device__ static void finish_normal(const double* sums28, const double x[7], const double* plusj, double wt,
                                              double wr, const double target_t[3], const double init_q[4], Normal* out) {
  int idx = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = r; c < 6; ++c) {
      out->H[tri(r, c)] = sums28[idx];
      ++idx;
    }
#pragma unroll
  for (int r = 0; r < 6; ++r) out->g[r] = sums28[21 + r];
  double sumsq = sums28[27];
  // translation_delta_cost_functor_3d.h:39-45 (only if weight > 0: ceres_scan_matcher_3d.cc:104-110)
  if (wt > 0.) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double r = wt * (x[i] - target_t[i]);
      out->H[tri(i, i)] += wt * wt;
      out->g[i] += wt * r;
      sumsq += r * r;
    }
  }
  // rotation_delta_cost_functor_3d.h:43-54: r = w * (q_init^-1 (x) q).xyz, linear in q
  if (wr > 0.) {
    const double z[4] = {init_q[0], -init_q[1], -init_q[2], -init_q[3]};
    double d[4];
    quat_product(z, x + 3, d);
    // rows of d(delta_k)/dq for k = 1..3 (common/math.h:74-81)
    const double D[3][4] = {{z[1], z[0], -z[3], z[2]}, {z[2], z[3], z[0], -z[1]}, {z[3], -z[2], z[1], z[0]}};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double r = wr * d[k + 1];
      double jl[3] = {0, 0, 0};
#pragma unroll
      for (int c = 0; c < nloc; ++c) {
        double s = 0.;
#pragma unroll
        for (int m = 0; m < 4; ++m) s += (wr * D[k][m]) * plusj[m * nloc + c];
        jl[c] = s;
      }
#pragma unroll
      for (int c1 = 0; c1 < nloc; ++c1) {
        out->g[3 + c1] += jl[c1] * r;
#pragma unroll
        for (int c2 = c1; c2 < nloc; ++c2) out->H[tri(3 + c1, 3 + c2)] += jl[c1] * jl[c2];
      }
      sumsq += r * r;
    }
  }
  out->cost = 0.5 * sumsq;
}

// One evaluation at x = [t, q]: device occupied-space sums + host prior residuals.
static int evaluate(CsmProblem* p, const double x[7], Normal* out) {
  dliom_ctx* ctx = p->ctx;
  CsmArgs& a = p->args;
  for (int i = 0; i < 3; ++i) a.pose.t[i] = x[i];
  for (int i = 0; i < 4; ++i) a.pose.q[i] = x[3 + i];
  a.pose.nloc = p->nloc;
  plus_jacobian(x + 3, p->nloc, a.pose.plus);
  const float kMin = 0.1f, kMax = 1.f - 0.1f;
  const float k_scale = (kMax - kMin) / 32766.f;
  const float k_offset = kMin - k_scale;
  const int span = ctx->begin_span(DLIOM_KERNEL_CSM_EVAL);
  // the 28 results go straight into pinned host memory (device-visible): no copy command
  double* host = static_cast<double*>(ctx->pinned);
  // ... and a completion word behind them, which the host polls: ten evaluations a match, 5 us of synchronise each
  unsigned* done = p->d_arrivals != nullptr ? ctx->done_word : nullptr;
  const unsigned seq = done != nullptr ? (++ctx->done_seq == 0u ? ++ctx->done_seq : ctx->done_seq) : 0u;
  hipLaunchKernelGGL(csm_eval_kernel, dim3(p->num_blocks), dim3(kCsmBlock), 0, ctx->stream, a, k_scale,
                     k_offset, kMin, p->d_partials, host, done, seq);
  if (p->num_blocks > 1)
    hipLaunchKernelGGL(csm_final_reduce_kernel, dim3(kAcc), dim3(64), 0, ctx->stream, p->d_partials,
                       p->num_blocks, host, p->d_arrivals, done, seq);
  ctx->end_span(span);
  DLIOM_HIP_TRY(hipGetLastError());
  if (done != nullptr)
    DLIOM_TRY(wait_done(ctx, ctx->stream, done, seq));
  else
    DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  ++p->evaluations;
  if (p->nloc == 3)
    finish_normal<3>(host, x, a.pose.plus, p->o->translation_weight, p->o->rotation_weight, p->target_t, p->init_q, out);
  else
    finish_normal<1>(host, x, a.pose.plus, p->o->translation_weight, p->o->rotation_weight, p->target_t, p->init_q, out);
  return DLIOM_OK;
}

// 1 / sqrt(s) to double precision without a double square root or division (each a dependent sequence of ~30
// instructions on the device, and a pivot of the Cholesky factorisation below waits for it): a seed from the exponent
// bits (the classic shift-and-subtract, 3.4 % off at worst) and five Newton steps in fused multiply-adds -- only
// integer operations and IEEE multiply-adds, so host and device get the same bits.
__host__ __device__ static inline double inv_sqrt(double s) {
  if (!(s > 1e-300 && s < 1e300)) return 1.0 / sqrt(s);
  long long bits;
  memcpy(&bits, &s, sizeof(bits));
  bits = 0x5FE6EB50C7B537A9ll - (bits >> 1);
  double y;
  memcpy(&y, &bits, sizeof(y));
  const double h = 0.5 * s;
#pragma unroll
  for (int it = 0; it < 5; ++it) {  // relative error 3.4e-2 -> 1.8e-3 -> 4.6e-6 -> 3.2e-11 -> 1.5e-21 (-> rounding)
    const double e = fma(-(h * y), y, 0.5);  // 1/2 - s y^2 / 2
    y = fma(y, e, y);
  }
  return y;
}

// (A + diag(d2)) y = b for the leading n x n block, Cholesky; false if not positive definite.
// (n is a template argument and every loop is unrolled: in csm_lm_kernel the matrices must live in registers, a
// dynamically indexed local array would sit in scratch memory at ~1 us per dependent access.)
template <int n>
__host__ __device__ static bool solve_spd(const double* A, const double* d2, const double* b, double* y) {
  // One reciprocal square root per pivot (inv_sqrt), multiplications elsewhere: a double division or square root is a
  // ~30-instruction dependent sequence on the device, and this solve runs on the critical path of every LM iteration
  // (csm_lm_kernel).
  // Host and device run this same code, so the launch-per-evaluation loop and the one-launch kernel stay identical.
  double Lm[36], rd[6];
#pragma unroll
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = A[tri(i, j)] + (i == j ? d2[i] : 0.0);
#pragma unroll
      for (int k = 0; k < j; ++k) s -= Lm[i * 6 + k] * Lm[j * 6 + k];
      if (i == j) {
        if (!(s > 0.0)) return false;
        rd[i] = inv_sqrt(s);
        Lm[i * 6 + i] = s * rd[i];
      } else {
        Lm[i * 6 + j] = s * rd[j];
      }
    }
  }
  double z[6];
#pragma unroll
  for (int i = 0; i < n; ++i) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= Lm[i * 6 + k] * z[k];
    z[i] = s * rd[i];
  }
#pragma unroll
  for (int i = n - 1; i >= 0; --i) {
    double s = z[i];
#pragma unroll
    for (int k = i + 1; k < n; ++k) s -= Lm[k * 6 + i] * y[k];
    y[i] = s * rd[i];
  }
#pragma unroll
  for (int i = 0; i < n; ++i)
    if (!(fabs(y[i]) <= 1.7976931348623157e308)) return false;  // !isfinite
  return true;
}

#ifdef DLIOM_EXPERIMENTS
__device__ unsigned long long dbg_lm2[256];
#endif
#if defined(DLIOM_EXPERIMENTS) && defined(__HIP_DEVICE_COMPILE__)
#define DLIOM_LM2_STAMP(i) if (threadIdx.x == 0 && blockIdx.x == 0 && iteration < 30) dbg_lm2[8 * iteration + (i)] = __builtin_readcyclecounter()
#else
#define DLIOM_LM2_STAMP(i)
#endif
// Ceres 1.13 trust-region minimizer (LEVENBERG_MARQUARDT) on the normal equations.
// `ev(x, &normal)` evaluates the normal equations at x (0 = ok) and counts in ev.evaluations; on the host it launches
// the evaluation kernel per call, inside csm_lm_kernel it is the workgroup's own reduction -- the SAME loop either way.
struct LmConfig {
  int max_num_iterations;
  int use_nonmonotonic_steps;
  int nloc;
};
template <int nloc, class Eval>
__host__ __device__ static int minimize(Eval& ev, const LmConfig& o, double x[7], dliom_csm_summary* sum) {
  constexpr int ne = 3 + nloc;
  const double kFunctionTol = 1e-6, kGradientTol = 1e-10, kParameterTol = 1e-8;
  const double kMinRelDecrease = 1e-3, kMinDiag = 1e-6, kMaxDiag = 1e32;
  const double kMaxRadius = 1e16, kMinRadius = 1e-32;
  const int kMaxInvalid = 5;
  double radius = 1e4, decrease_factor = 2.0;
  auto norm7 = [](const double* v) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) s += v[i] * v[i];
    return sqrt(s);
  };
  auto grad_max_norm = [&](const double* xx, const Normal& nrm) {
    double neg[6] = {0, 0, 0, 0, 0, 0}, proj[7];
#pragma unroll
    for (int i = 0; i < ne; ++i) neg[i] = -nrm.g[i];
    plus(xx, neg, nloc, proj);
    double m = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) m = fmax(m, fabs(xx[i] - proj[i]));
    return m;
  };
  sum->initial_cost = sum->final_cost = 0.0;
  sum->num_successful_steps = sum->num_unsuccessful_steps = sum->num_iterations = 0;
  sum->num_residual_evaluations = sum->num_jacobian_evaluations = sum->termination_type = 0;
  Normal cur;
  {
    const int es = ev(x, &cur);
    if (es != DLIOM_OK) return es;
  }

  double x_cost = cur.cost, minimum_cost = cur.cost;
  double best_x[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) best_x[i] = x[i];
  sum->initial_cost = x_cost;
  double min_iter_cost = x_cost;
  int num_iter_records = 1;
  double scale[6] = {1, 1, 1, 1, 1, 1};
#pragma unroll
  for (int i = 0; i < ne; ++i) scale[i] = 1.0 / (1.0 + sqrt(cur.H[tri(i, i)]));
  double x_norm = norm7(x);
  double gmax = grad_max_norm(x, cur);
  auto finish = [&](int type) {
#pragma unroll
    for (int i = 0; i < 7; ++i) x[i] = best_x[i];
    sum->final_cost = fmin(sum->initial_cost, min_iter_cost);
    sum->num_iterations = num_iter_records;
    sum->num_residual_evaluations = ev.evaluations;
    sum->num_jacobian_evaluations = ev.evaluations;
    sum->termination_type = type;
    return type == 2 ? DLIOM_ERR_SOLVER : DLIOM_OK;
  };
  if (gmax <= kGradientTol) return finish(0);

  // trust_region_step_evaluator.cc state
  const int max_nonmono = o.use_nonmonotonic_steps ? 5 : 0;
  double ev_min = x_cost, ev_cur = x_cost, ev_ref = x_cost, ev_cand = x_cost;
  double acc_ref = 0.0, acc_cand = 0.0;
  int nonmono = 0;

  int iteration = 0, invalid = 0;
  bool last_ok = false;
  for (;;) {
    if (last_ok) {
      ++sum->num_successful_steps;
      if (x_cost < minimum_cost) {
        minimum_cost = x_cost;
#pragma unroll
        for (int i = 0; i < 7; ++i) best_x[i] = x[i];
      }
    } else if (iteration > 0) {
      ++sum->num_unsuccessful_steps;
    }
    if (iteration >= o.max_num_iterations) return finish(1);
    if (radius <= kMinRadius) return finish(0);
    // The gradient test of an accepted step (max |x - Plus(x, -g)| <= tolerance -> converged, like the radius test above)
    // is evaluated BELOW, behind the solve: its projection and the factorisation are two independent dependent chains and
    // the device, which issues in order, interleaves them only inside one block.  A converged run discards the solve.
    const bool accepted_step = last_ok;
    ++iteration;
    last_ok = false;
    DLIOM_LM2_STAMP(0);
    if (accepted_step) gmax = grad_max_norm(x, cur);

    // scaled normal equations Hs = S H S, gs = S g; LM diagonal from diag(Hs)
    double Hs[21], gs[6], d2[6], y[6], step[6];  // (Hs: upper triangle like Normal::H)
#pragma unroll
    for (int r = 0; r < ne; ++r) {
      gs[r] = cur.g[r] * scale[r];
#pragma unroll
      for (int c = r; c < ne; ++c) Hs[tri(r, c)] = cur.H[tri(r, c)] * scale[r] * scale[c];
    }
    const double inv_radius = 1.0 / radius;
#pragma unroll
    for (int r = 0; r < ne; ++r)
      d2[r] = fmin(fmax(Hs[tri(r, r)], kMinDiag), kMaxDiag) * inv_radius;
    DLIOM_LM2_STAMP(1);
    bool valid = solve_spd<ne>(Hs, d2, gs, y);
    DLIOM_LM2_STAMP(2);
    if (accepted_step && gmax <= kGradientTol) {
      --iteration;  // (the test belongs in front of this iteration)
      return finish(0);
    }
    double model_cost_change = 0.0;
    if (valid) {
#pragma unroll
      for (int r = 0; r < ne; ++r) step[r] = -y[r];
      // -(J s)^T (r + J s / 2) = -s^T gs - 1/2 s^T Hs s
      double sg = 0.0, shs = 0.0;
#pragma unroll
      for (int r = 0; r < ne; ++r) {
        sg += step[r] * gs[r];
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < ne; ++c) t += Hs[tri(r, c)] * step[c];
        shs += step[r] * t;
      }
      model_cost_change = -sg - 0.5 * shs;
      valid = model_cost_change > 0.0;
    }
    if (!valid) {
      if (++invalid >= kMaxInvalid) return finish(2);
      radius *= 0.5;
      min_iter_cost = fmin(min_iter_cost, x_cost);
      ++num_iter_records;
      continue;
    }
    invalid = 0;
    double delta[6] = {0, 0, 0, 0, 0, 0}, cand[7];
#pragma unroll
    for (int r = 0; r < ne; ++r) delta[r] = step[r] * scale[r];
    plus(x, delta, nloc, cand);
    DLIOM_LM2_STAMP(3);
    Normal cn;
    {
      const int es = ev(cand, &cn);
      if (es != DLIOM_OK) return es;
    }
    DLIOM_LM2_STAMP(4);
    const double cand_cost = cn.cost;
    double step_norm = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) step_norm += (x[i] - cand[i]) * (x[i] - cand[i]);
    step_norm = sqrt(step_norm);
    if (step_norm <= kParameterTol * (x_norm + kParameterTol)) return finish(0);
    if (fabs(x_cost - cand_cost) <= kFunctionTol * x_cost) return finish(0);
    const double rel = (ev_cur - cand_cost) / model_cost_change;
    const double hist = (ev_ref - cand_cost) / (acc_ref + model_cost_change);
    const double quality = fmax(rel, hist);
    DLIOM_LM2_STAMP(5);
    if (quality > kMinRelDecrease) {
#pragma unroll
      for (int i = 0; i < 7; ++i) x[i] = cand[i];
      x_norm = norm7(x);
      cur = cn;
      x_cost = cand_cost;
      last_ok = true;
      const double tq = 2.0 * quality - 1.0;
      radius = radius / fmax(1.0 / 3.0, 1.0 - tq * tq * tq);  // (pow(., 3) is a few hundred instructions on the device)
      radius = fmin(kMaxRadius, radius);
      decrease_factor = 2.0;
      ev_cur = cand_cost;
      acc_cand += model_cost_change;
      acc_ref += model_cost_change;
      if (ev_cur < ev_min) {
        ev_min = ev_cur;
        nonmono = 0;
        ev_cand = ev_cur;
        acc_cand = 0.0;
      } else {
        ++nonmono;
        if (ev_cur > ev_cand) {
          ev_cand = ev_cur;
          acc_cand = 0.0;
        }
      }
      if (nonmono == max_nonmono) {
        ev_ref = ev_cand;
        acc_ref = acc_cand;
      }
      min_iter_cost = fmin(min_iter_cost, x_cost);
    } else {
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      min_iter_cost = fmin(min_iter_cost, cand_cost);
    }
    DLIOM_LM2_STAMP(6);
    ++num_iter_records;
  }
}

struct HostEval {
  CsmProblem* p;
  int evaluations = 0;
  int operator()(const double x[7], Normal* out) {
    const int s = evaluate(p, x, out);
    evaluations = p->evaluations;
    return s;
  }
};

// ---- the whole Levenberg-Marquardt loop in ONE launch (small clouds: the reference's own configuration matches
// ~170 + ~210 points after the adaptive voxel filters, where ten launch + synchronise round trips of ~25 us each
// were 90 % of CeresScanMatcher3D's time).  One workgroup: every thread runs the same minimize<> loop on the same
// numbers (uniform control flow); an evaluation is the workgroup's strided accumulation + the fixed-order
// reduction of csm_eval_kernel (block_reduce28), so for clouds that fit one workgroup of that kernel the sums are bit-identical.
struct LmKernelParams {
  LmConfig cfg;
  double translation_weight, rotation_weight;
  double target_t[3];
  double init_q[4];
  double x0[7];
  float k_scale, k_offset, k_unknown;
  unsigned* done_word;  // csm_lm_kernel: completion word the host polls (null: full synchronise)
  unsigned done_seq;
};
struct LmKernelOut {
  double x[7];
  dliom_csm_summary summary;
  int status;
};

#ifdef DLIOM_EXPERIMENTS
__device__ unsigned long long dbg_lm[128];
#define DLIOM_LM_STAMP(i) if (threadIdx.x == 0 && evaluations < 24) dbg_lm[5 * evaluations + (i)] = __builtin_readcyclecounter()
#else
#define DLIOM_LM_STAMP(i)
#endif
// (Round 4 tried two and three groups of 256 threads, one point per thread, the products handed to group 0 through LDS
// in the evaluation kernel's order of additions: NOT faster.  An evaluation is bound by instruction issue -- ~750 vector
// instructions per point without contraction, 4 cycles each, on the four SIMDs of the ONE compute unit a workgroup
// lives on -- so eight waves with one point each issue what four waves with two points did, the exchange costs three
// more barriers, and the trust-region step, which every wave runs redundantly, has to share its SIMD: 414 000 cycles
// per match against 325 000.  What did help: fewer instructions, see plus(), inv_sqrt() and csm_point_v.)
template <int NLOC>
struct DeviceEval {
  const CsmArgs* a;  // clouds (kernel argument)
  const LmKernelParams* prm;
  double* part;              // block_reduce28's LDS: [2][4][32]
  double pts[6];             // two-cloud fast path: this thread's point of cloud 0 and of cloud 1 (index clamped)
  int evaluations = 0;
  __device__ int operator()(const double x[7], Normal* out) {
    DLIOM_LM_STAMP(0);
    CsmPose pose;
    for (int i = 0; i < 3; ++i) pose.t[i] = x[i];
    for (int i = 0; i < 4; ++i) pose.q[i] = x[3 + i];
    pose.nloc = NLOC;
    plus_jacobian(x + 3, NLOC, pose.plus);
    double acc[kAcc];
#pragma unroll
    for (int k = 0; k < kAcc; ++k) acc[k] = 0.;
    auto add = [&acc](double r, const double* j) {
      int idx = 0;
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int q = p; q < 6; ++q) acc[idx++] += j[p] * j[q];
#pragma unroll
      for (int p = 0; p < 6; ++p) acc[21 + p] += j[p] * r;
      acc[27] += r * r;
    };
    const int n0 = a->cloud[0].n, n1 = a->cloud[1].n;
    if (a->num_clouds == 2 && n0 > 0 && n1 > 0 && n0 <= kCsmBlock && n1 <= kCsmBlock) {
      // The reference's own case (two adaptively filtered clouds of ~150 and ~200 points): one point of each cloud per
      // thread.  Both are evaluated in ONE basic block (index clamped, sums predicated) so that the two chains of
      // dependent loads (point -> leaf table -> leaf) overlap; same additions in the same order as the loop below.
      const int t = threadIdx.x;
      double r0, j0[6], r1, j1[6];
      // the thread's two points were loaded once, before the first evaluation (pts: registers across the whole loop)
      csm_point_v(pose, a->cloud[0], pts[0], pts[1], pts[2], prm->k_scale, prm->k_offset, prm->k_unknown, &r0, j0);
      csm_point_v(pose, a->cloud[1], pts[3], pts[4], pts[5], prm->k_scale, prm->k_offset, prm->k_unknown, &r1, j1);
      if (t < n0) add(r0, j0);
      if (t < n1) add(r1, j1);
    } else {
      for (int ci = 0; ci < a->num_clouds; ++ci) {
        const CsmCloudArg& c = a->cloud[ci];
        for (int i = threadIdx.x; i < c.n; i += kCsmBlock) {
          double r, j[6];
          csm_point(pose, c, i, prm->k_scale, prm->k_offset, prm->k_unknown, &r, j);
          add(r, j);
        }
      }
    }
    DLIOM_LM_STAMP(1);
    double sums[kAcc];
    block_reduce28(acc, part, evaluations, sums);  // (csm_eval_kernel's order of additions)
    DLIOM_LM_STAMP(2);
    finish_normal<NLOC>(sums, x, pose.plus, prm->translation_weight, prm->rotation_weight, prm->target_t, prm->init_q, out);
    DLIOM_LM_STAMP(3);
    ++evaluations;
    return DLIOM_OK;
  }
};

template <int NLOC>
__global__ __launch_bounds__(kCsmBlock) void csm_lm_kernel(CsmArgs a, LmKernelParams prm, LmKernelOut* out) {
  __shared__ double part[2 * (kCsmBlock / 64) * 32];
  DeviceEval<NLOC> ev;
  ev.a = &a;
  ev.prm = &prm;
  ev.part = part;
  {
    const int n0 = a.cloud[0].n, n1 = a.cloud[1].n;
    const bool pair = a.num_clouds == 2 && n0 > 0 && n1 > 0 && n0 <= kCsmBlock && n1 <= kCsmBlock;
    const int t = threadIdx.x;
    const int i0 = pair ? (t < n0 ? t : n0 - 1) : 0, i1 = pair ? (t < n1 ? t : n1 - 1) : 0;
    ev.pts[0] = pair ? static_cast<double>(a.cloud[0].x[i0]) : 0.0;
    ev.pts[1] = pair ? static_cast<double>(a.cloud[0].y[i0]) : 0.0;
    ev.pts[2] = pair ? static_cast<double>(a.cloud[0].z[i0]) : 0.0;
    ev.pts[3] = pair ? static_cast<double>(a.cloud[1].x[i1]) : 0.0;
    ev.pts[4] = pair ? static_cast<double>(a.cloud[1].y[i1]) : 0.0;
    ev.pts[5] = pair ? static_cast<double>(a.cloud[1].z[i1]) : 0.0;
  }
  double x[7];
  for (int i = 0; i < 7; ++i) x[i] = prm.x0[i];
  dliom_csm_summary sum;
  const int status = minimize<NLOC>(ev, prm.cfg, x, &sum);
#ifdef DLIOM_EXPERIMENTS
  if (threadIdx.x == 0) {
    dbg_lm[126] = __builtin_readcyclecounter();
    dbg_lm[127] = static_cast<unsigned long long>(ev.evaluations);
  }
#endif
  if (threadIdx.x == 0) {
    for (int i = 0; i < 7; ++i) out->x[i] = x[i];
    out->summary = sum;
    out->status = status;
    if (prm.done_word != nullptr) {
      __threadfence_system();
      *reinterpret_cast<volatile unsigned*>(prm.done_word) = prm.done_seq;
    }
  }
}

#ifdef DLIOM_EXPERIMENTS
}  // namespace dliom
extern "C" int dliom_exp_lm2_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dliom::dbg_lm2), sizeof(unsigned long long) * 256) == hipSuccess ? 0 : -2;
}
extern "C" int dliom_exp_lm_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dliom::dbg_lm), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -2;
}
namespace dliom {
#endif

// ---- the whole Levenberg-Marquardt loop in ONE launch for LARGE clouds: the same grid as csm_eval_kernel (every
// workgroup resident), every thread of every workgroup runs the same minimize<> loop on the same numbers, and an
// evaluation is csm_eval_kernel's accumulation + block reduction, a grid barrier, and csm_final_reduce_kernel's
// reduction done by every workgroup for itself -- the same additions in the same order, hence the same bits as the
// per-evaluation loop, without its launch + synchronise round trip per evaluation (ten of them per match: 0.29 ms per
// 131 072-point match, of which the kernels were 0.17).  The partial sums alternate between two buffers, so one barrier
// per evaluation is enough.  Barrier: agent-scope release / acquire on one counter (MI355X_MICROARCH.md, "Inter-workgroup
// visibility"), every spin bounded.
struct GridSync {
  unsigned* counter;   // zeroed by the host before the launch
  unsigned target;     // arrivals expected so far
  bool timed_out;
};
__device__ __forceinline__ void grid_barrier(GridSync& gs) {
  __syncthreads();  // this workgroup's partial sums are written
  gs.target += gridDim.x;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(gs.counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(gs.counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gs.target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 24)) break;  // ~seconds: a workgroup that never became resident; the match reports an error
    }
  }
  __syncthreads();
}

template <int NLOC>
struct DeviceGridEval {
  const CsmArgs* a;
  const LmKernelParams* prm;
  double* part;              // block_reduce28's LDS: [2][4][32]
  double* tot;               // [kAcc]
  double* partials;          // 2 x gridDim.x x kAcc
  GridSync gs;
  int evaluations = 0;
  __device__ int operator()(const double x[7], Normal* out) {
    CsmPose pose;
    for (int i = 0; i < 3; ++i) pose.t[i] = x[i];
    for (int i = 0; i < 4; ++i) pose.q[i] = x[3 + i];
    pose.nloc = NLOC;
    plus_jacobian(x + 3, NLOC, pose.plus);
    double acc[kAcc];
#pragma unroll
    for (int k = 0; k < kAcc; ++k) acc[k] = 0.;
    const int stride = gridDim.x * blockDim.x;
    for (int ci = 0; ci < a->num_clouds; ++ci) {
      const CsmCloudArg& c = a->cloud[ci];
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < c.n; i += stride) {
        double r, j[6];
        csm_point(pose, c, i, prm->k_scale, prm->k_offset, prm->k_unknown, &r, j);
        int idx = 0;
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
          for (int q = p; q < 6; ++q) acc[idx++] += j[p] * j[q];
#pragma unroll
        for (int p = 0; p < 6; ++p) acc[21 + p] += j[p] * r;
        acc[27] += r * r;
      }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* mine = partials + static_cast<size_t>(evaluations & 1) * gridDim.x * kAcc;
    {
      double block_sums[kAcc];
      block_reduce28(acc, part, evaluations, block_sums);  // csm_eval_kernel's block reduction
      if (threadIdx.x < kAcc) {
        double v = block_sums[0];
#pragma unroll
        for (int k = 1; k < kAcc; ++k)
          if (static_cast<int>(threadIdx.x) == k) v = block_sums[k];
        mine[blockIdx.x * kAcc + threadIdx.x] = v;
      }
    }
    grid_barrier(gs);
    for (int k = wave; k < kAcc; k += kCsmBlock / 64) {  // csm_final_reduce_kernel's reduction
      double sk = 0.;
      for (int b = lane; b < static_cast<int>(gridDim.x); b += 64) sk += mine[b * kAcc + k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) sk += __shfl_xor(sk, off, 64);
      if (lane == 0) tot[k] = sk;
    }
    __syncthreads();
    double sums[kAcc];
#pragma unroll
    for (int k = 0; k < kAcc; ++k) sums[k] = tot[k];
    finish_normal<NLOC>(sums, x, pose.plus, prm->translation_weight, prm->rotation_weight, prm->target_t, prm->init_q, out);
    ++evaluations;
    return DLIOM_OK;
  }
};

template <int NLOC>
__global__ __launch_bounds__(kCsmBlock) void csm_lm_grid_kernel(CsmArgs a, LmKernelParams prm, double* partials,
                                                                unsigned* counter, LmKernelOut* out) {
  __shared__ double part[2 * (kCsmBlock / 64) * 32];
  __shared__ double tot[kAcc];
  DeviceGridEval<NLOC> ev;
  ev.a = &a;
  ev.prm = &prm;
  ev.part = part;
  ev.tot = tot;
  ev.partials = partials;
  ev.gs = GridSync{counter, 0u, false};
  double x[7];
  for (int i = 0; i < 7; ++i) x[i] = prm.x0[i];
  dliom_csm_summary sum;
  const int status = minimize<NLOC>(ev, prm.cfg, x, &sum);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int i = 0; i < 7; ++i) out->x[i] = x[i];
    out->summary = sum;
    // every workgroup must have taken part in every barrier: the counter tells
    const unsigned arrived = __hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    out->status = arrived >= ev.gs.target ? status : DLIOM_ERR_HIP;
  }
}

static int setup_problem(dliom_ctx* ctx, const dliom_csm_options* o, const double target_t[3],
                         const double init7[7], int k, const dliom_cloud* const* clouds,
                         const dliom_grid* const* grids, CsmProblem* p) {
  if (o->num_occupied_space_weights != k || k <= 0 || k > DLIOM_MAX_CLOUDS) return DLIOM_ERR_WEIGHTS;
  p->ctx = ctx;
  p->o = o;
  p->nloc = o->only_optimize_yaw ? 1 : 3;
  p->evaluations = 0;
  int total = 0;
  for (int i = 0; i < k; ++i) {
    if (!(o->occupied_space_weight[i] > 0.)) return DLIOM_ERR_WEIGHTS;
    if (clouds[i] == nullptr || grids[i] == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
    if (clouds[i]->n <= 0) return DLIOM_ERR_EMPTY_CLOUD;
    CsmCloudArg& c = p->args.cloud[i];
    c.g = grids[i]->view();
    {
      // the mirror stands in for the leaf table only if "unknown" (0) and the mirror's 1 mean the same probability
      const float kMin = 0.1f, kMax = 1.f - 0.1f;
      const float k_scale = (kMax - kMin) / 32766.f, k_offset = kMin - k_scale;
      if (!(1.f * k_scale + k_offset == kMin)) c.g.dense = nullptr;
    }
    DLIOM_TRY(ensure_morton(ctx, clouds[i]));
    c.x = clouds[i]->d_xs;  // Morton order: neighbouring lanes read neighbouring voxels; the
    c.y = clouds[i]->d_ys;  // reduction order is still fixed, so results stay reproducible
    c.z = clouds[i]->d_zs;
    c.n = static_cast<int>(clouds[i]->n);
    c.scale = o->occupied_space_weight[i] / std::sqrt(static_cast<double>(clouds[i]->n));
    total += c.n;
  }
  p->args.num_clouds = k;
  p->args.total_points = total;
  for (int i = 0; i < 3; ++i) p->target_t[i] = target_t[i];
  for (int i = 0; i < 4; ++i) p->init_q[i] = init7[3 + i];
  p->num_blocks = std::max(1, std::min(512, (total + 2 * kCsmBlock - 1) / (2 * kCsmBlock)));
  DLIOM_TRY(ctx->partials.reserve(static_cast<size_t>(p->num_blocks + 1) * kAcc * sizeof(double)));
  p->d_partials = ctx->partials.as<double>();
  p->d_out = p->d_partials + static_cast<size_t>(p->num_blocks) * kAcc;
  // the word csm_final_reduce_kernel counts its waves in: zeroed once when it is allocated, re-armed by the kernel itself
  if (ctx->csm_arrivals.p == nullptr) {
    DLIOM_TRY(ctx->csm_arrivals.reserve(256));
    DLIOM_HIP_TRY(hipMemsetAsync(ctx->csm_arrivals.p, 0, 256, ctx->stream));
  }
  p->d_arrivals = ctx->csm_arrivals.as<unsigned>();
  return DLIOM_OK;
}

static int stage_clouds(dliom_ctx* ctx, int k, const float* const* pts, const int64_t* n,
                        std::vector<dliom_cloud>* staged, std::vector<const dliom_cloud*>* ptrs) {
  size_t total = 0;
  std::vector<size_t> off(k);
  for (int i = 0; i < k; ++i) {
    if (n[i] < 0 || (n[i] > 0 && pts[i] == nullptr)) return DLIOM_ERR_INVALID_ARGUMENT;
    if (n[i] == 0) return DLIOM_ERR_EMPTY_CLOUD;
    off[i] = total;
    total += (staged_cloud_bytes(n[i]) + 255) & ~static_cast<size_t>(255);
  }
  DLIOM_TRY(ctx->points.reserve(total));
  staged->resize(k);
  ptrs->resize(k);
  for (int i = 0; i < k; ++i) {
    DLIOM_TRY(stage_cloud(ctx, pts[i], n[i], &(*staged)[i], off[i]));
    (*ptrs)[i] = &(*staged)[i];
  }
  return DLIOM_OK;
}

}  // namespace dliom

using namespace dliom;

extern "C" {

int dliom_csm3d_match_cloud(dliom_ctx* ctx, const dliom_csm_options* o, const double target_t[3],
                            const double init7[7], int k, const dliom_cloud* const* clouds,
                            const dliom_grid* const* grids, double out7[7], dliom_csm_summary* summary) {
  if (ctx == nullptr || o == nullptr || target_t == nullptr || init7 == nullptr || clouds == nullptr ||
      grids == nullptr || out7 == nullptr)
    return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  CsmProblem p;
  DLIOM_TRY(setup_problem(ctx, o, target_t, init7, k, clouds, grids, &p));
  double x[7];
  std::memcpy(x, init7, sizeof(x));
  dliom_csm_summary local;
  dliom_csm_summary* sum = summary != nullptr ? summary : &local;
  const LmConfig cfg{o->max_num_iterations, o->use_nonmonotonic_steps, p.nloc};
  // total points up to which the one-launch loop is used (0 = never)
  const int persistent_max = ctx->tuning[DLIOM_TUNE_CSM_ONE_LAUNCH_MAX];
  if (p.args.total_points <= persistent_max) {
    LmKernelParams prm;
    prm.cfg = cfg;
    prm.translation_weight = o->translation_weight;
    prm.rotation_weight = o->rotation_weight;
    for (int i = 0; i < 3; ++i) prm.target_t[i] = p.target_t[i];
    for (int i = 0; i < 4; ++i) prm.init_q[i] = p.init_q[i];
    for (int i = 0; i < 7; ++i) prm.x0[i] = x[i];
    const float kMin = 0.1f, kMax = 1.f - 0.1f;
    prm.k_scale = (kMax - kMin) / 32766.f;
    prm.k_offset = kMin - prm.k_scale;
    prm.k_unknown = kMin;
    LmKernelOut* host = reinterpret_cast<LmKernelOut*>(static_cast<char*>(ctx->pinned) + 1024);  // device-visible
    prm.done_word = ctx->done_word;
    prm.done_seq = ctx->done_word != nullptr ? (++ctx->done_seq == 0u ? ++ctx->done_seq : ctx->done_seq) : 0u;
    const int span = ctx->begin_span(DLIOM_KERNEL_CSM_EVAL);
    if (p.nloc == 3)
      hipLaunchKernelGGL(csm_lm_kernel<3>, dim3(1), dim3(kCsmBlock), 0, ctx->stream, p.args, prm, host);
    else
      hipLaunchKernelGGL(csm_lm_kernel<1>, dim3(1), dim3(kCsmBlock), 0, ctx->stream, p.args, prm, host);
    ctx->end_span(span);
    DLIOM_HIP_TRY(hipGetLastError());
    if (prm.done_word != nullptr)
      DLIOM_TRY(wait_done(ctx, ctx->stream, prm.done_word, prm.done_seq));
    else
      DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::memcpy(out7, host->x, sizeof(x));
    *sum = host->summary;
    return host->status;
  }
  const int num_cus = ctx->num_cus;
  // large clouds: the loop in one launch with grid barriers when every workgroup of the evaluation grid is resident at
  // once (256 threads and 57 KB of LDS each: two per CU)
  if (ctx->tuning[DLIOM_TUNE_CSM_GRID_SYNC] != 0 && p.num_blocks > 1 && p.num_blocks <= 2 * num_cus) {
    LmKernelParams prm;
    prm.cfg = cfg;
    prm.translation_weight = o->translation_weight;
    prm.rotation_weight = o->rotation_weight;
    for (int i = 0; i < 3; ++i) prm.target_t[i] = p.target_t[i];
    for (int i = 0; i < 4; ++i) prm.init_q[i] = p.init_q[i];
    for (int i = 0; i < 7; ++i) prm.x0[i] = x[i];
    const float kMin = 0.1f, kMax = 1.f - 0.1f;
    prm.k_scale = (kMax - kMin) / 32766.f;
    prm.k_offset = kMin - prm.k_scale;
    prm.k_unknown = kMin;
    prm.done_word = nullptr;
    prm.done_seq = 0u;
    const size_t part_bytes = (2 * static_cast<size_t>(p.num_blocks) * kAcc * sizeof(double) + 255) & ~static_cast<size_t>(255);
    DLIOM_TRY(ctx->partials.reserve(part_bytes + 256));
    double* partials = ctx->partials.as<double>();
    unsigned* counter = reinterpret_cast<unsigned*>(static_cast<char*>(ctx->partials.p) + part_bytes);
    DLIOM_HIP_TRY(hipMemsetAsync(counter, 0, 4, ctx->stream));
    LmKernelOut* host = reinterpret_cast<LmKernelOut*>(static_cast<char*>(ctx->pinned) + 1024);  // device-visible
    const int span = ctx->begin_span(DLIOM_KERNEL_CSM_EVAL);
    if (p.nloc == 3)
      hipLaunchKernelGGL(csm_lm_grid_kernel<3>, dim3(p.num_blocks), dim3(kCsmBlock), 0, ctx->stream, p.args, prm, partials, counter, host);
    else
      hipLaunchKernelGGL(csm_lm_grid_kernel<1>, dim3(p.num_blocks), dim3(kCsmBlock), 0, ctx->stream, p.args, prm, partials, counter, host);
    ctx->end_span(span);
    DLIOM_HIP_TRY(hipGetLastError());
    DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::memcpy(out7, host->x, sizeof(x));
    *sum = host->summary;
    return host->status;
  }
  HostEval ev{&p};
  const int s = p.nloc == 3 ? minimize<3>(ev, cfg, x, sum) : minimize<1>(ev, cfg, x, sum);
  std::memcpy(out7, x, sizeof(x));
  return s;
}

int dliom_csm3d_match(dliom_ctx* ctx, const dliom_csm_options* o, const double target_t[3],
                      const double init7[7], int k, const float* const* pts, const int64_t* n,
                      const dliom_grid* const* grids, double out7[7], dliom_csm_summary* summary) {
  if (ctx == nullptr || o == nullptr || pts == nullptr || n == nullptr || k <= 0 || k > DLIOM_MAX_CLOUDS)
    return k <= 0 || k > DLIOM_MAX_CLOUDS ? DLIOM_ERR_WEIGHTS : DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  std::vector<dliom_cloud> staged;
  std::vector<const dliom_cloud*> ptrs;
  DLIOM_TRY(stage_clouds(ctx, k, pts, n, &staged, &ptrs));
  return dliom_csm3d_match_cloud(ctx, o, target_t, init7, k, ptrs.data(), grids, out7, summary);
}

int dliom_csm3d_evaluate(dliom_ctx* ctx, const dliom_csm_options* o, const double target_t[3],
                         const double init7[7], const double pose[7], int k, const float* const* pts,
                         const int64_t* n, const dliom_grid* const* grids, double* cost,
                         double gradient[6], double jtj[36]) {
  if (ctx == nullptr || o == nullptr || target_t == nullptr || init7 == nullptr || pose == nullptr ||
      pts == nullptr || n == nullptr || grids == nullptr)
    return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  std::vector<dliom_cloud> staged;
  std::vector<const dliom_cloud*> ptrs;
  DLIOM_TRY(stage_clouds(ctx, k, pts, n, &staged, &ptrs));
  CsmProblem p;
  DLIOM_TRY(setup_problem(ctx, o, target_t, init7, k, ptrs.data(), grids, &p));
  Normal nrm;
  DLIOM_TRY(evaluate(&p, pose, &nrm));
  if (cost != nullptr) *cost = nrm.cost;
  if (gradient != nullptr) std::memcpy(gradient, nrm.g, sizeof(nrm.g));
  if (jtj != nullptr)
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) jtj[r * 6 + c] = nrm.H[tri(r, c)];
  return DLIOM_OK;
}

}  // extern "C"
