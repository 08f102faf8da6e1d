// Per-scan pre-processing of LocalTrajectoryBuilder3D::AddRangeData
// (mapping/internal/3d/local_trajectory_builder_3d.cc:393-487) on the device:
//   VoxelFilter(0.5 * voxel_filter_size) on the timed hits (:393-395)      voxel_filter.hip
//   per-hit de-skew: double slerp + rigid composition, range gate (:421-472)   deskew_kernel
//   returns = hits with min_range <= range <= max_range (order kept)        compaction
//   VoxelFilter(voxel_filter_size) on the returns (:479-484)                  voxel_filter.hip
//   TransformRangeData(., current_pose.inverse()) (:485-487)                  transform_kernel
// dliom_add_range_data() chains them without leaving HBM; dliom_deskew() is the de-skew alone on
// host buffers.
#include <cmath>
#include <cstring>
#include <vector>

#include "device_common.h"
#include "host_math.h"

namespace dliom {

struct DeskewArgs {
  double prev_t[3], prev_q[4];  // pose of the previous state (w,x,y,z)
  double rel_t[3], rel_q[4];    // prev^-1 * predicted pose at the scan stamp
  float cur_t[3], cur_q[4];     // predicted pose cast to float ("not de-skewing" branch)
  double scan_period;
  float ox, oy, oz;             // sensor origin in the tracking frame (origin 0)
  float origins[4][3];          // synchronized_data.origins (RangeDataSynchronizer: up to two lidars; room for 4)
  float min_range, max_range;
  int use_stamps;               // 0: |t_0| < 1e-3, every hit takes the predicted pose
  // How far the DEVICE's per-hit quaternion (double, before the cast to float) can be from the one the reference's host
  // computes with glibc: the two differ only through sin / acos (every other operation is a correctly rounded IEEE
  // operation on both).  q_bound[k]: absolute bound of component k before the normalisation, norm_bound: relative bound
  // the normalisation adds (make_deskew_args, with the derivation).  A hit whose cast could land on another float under
  // that bound is RECORDED and checked against glibc on the host (verify_deskew): equality is proven, not sampled.
  double q_bound[4], norm_bound;
  unsigned launch_tag;  // 0..31, changes with every launch: a record carries it, so that records a FAILED call left behind
                        // (written, never consumed) are recognised and skipped instead of being checked against another
                        // call's poses
};

// Records of hits whose cast is not provably the reference's: a ring in a small persistent buffer of the context,
// [0..6] the last hit's pose (the read-back the chain had already), [8] records written so far (monotonic, never reset),
// [16 + 6 r ..) record r % kDeskewRing = (hit index, bits of its time, bits of the four quaternion floats).
constexpr int kDeskewRing = 160;
constexpr int kDeskewFlagWords = 16 + 6 * kDeskewRing;  // 976 <= 1024: one job of gather_to_pinned

// Eigen Quaterniond product, SSE2 evaluation order (host_math.h::qmul_d), host and device
__host__ __device__ inline void quat_mul_sse_hd(const double* a, const double* b, double* r) {
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

// pose_i = (prev * [s t_rel, slerp(I, q_rel, s)]) (:437-445,869-877) for a hit with relative time t_rel: the rotation as
// doubles (qq, normalised), the translation already cast.  ONE source for the kernel and for the host's check: `sin` and
// `acos` are the device's (OCML) there and glibc's here -- the only operations in which the two can differ.
// *libm_path: the slerp took its sin / acos branch (else every operation is an IEEE operation: nothing to check).
__host__ __device__ inline void deskew_pose(const DeskewArgs& a, float t_rel, double qq[4], float t[3], bool* libm_path) {
  const double s = (a.scan_period + static_cast<double>(t_rel)) / a.scan_period;
  // Eigen::Quaterniond::Identity().slerp(s, q_rel)
  const double one = 1.0 - 2.220446049250313e-16;
  const double d = a.rel_q[0];
  const double abs_d = fabs(d);
  double scale0, scale1;
  *libm_path = !(abs_d >= one);
  if (abs_d >= one) {
    scale0 = 1.0 - s;
    scale1 = s;
  } else {
    const double theta = acos(abs_d);
    const double sin_theta = sin(theta);
    scale0 = sin((1.0 - s) * theta) / sin_theta;
    scale1 = sin(s * theta) / sin_theta;
  }
  if (d < 0.0) scale1 = -scale1;
  const double qi[4] = {scale0 * 1.0 + scale1 * a.rel_q[0], scale0 * 0.0 + scale1 * a.rel_q[1],
                        scale0 * 0.0 + scale1 * a.rel_q[2], scale0 * 0.0 + scale1 * a.rel_q[3]};
  const double ti[3] = {s * a.rel_t[0], s * a.rel_t[1], s * a.rel_t[2]};
  // prev * tmp: rotation (prev.q * qi).normalized(), translation prev.q * ti + prev.t
  quat_mul_sse_hd(a.prev_q, qi, qq);
  const double z2 = (qq[1] * qq[1] + qq[3] * qq[3]) + (qq[2] * qq[2] + qq[0] * qq[0]);
  if (z2 > 0.0) {
    const double nrm = sqrt(z2);
    qq[0] /= nrm;
    qq[1] /= nrm;
    qq[2] /= nrm;
    qq[3] /= nrm;
  }
  const double* u = a.prev_q;
  double uvx = u[2] * ti[2] - u[3] * ti[1], uvy = u[3] * ti[0] - u[1] * ti[2], uvz = u[1] * ti[1] - u[2] * ti[0];
  uvx += uvx;
  uvy += uvy;
  uvz += uvz;
  const double cx = u[2] * uvz - u[3] * uvy, cy = u[3] * uvx - u[1] * uvz, cz = u[1] * uvy - u[2] * uvx;
  t[0] = static_cast<float>(((ti[0] + u[0] * uvx) + cx) + a.prev_t[0]);
  t[1] = static_cast<float>(((ti[1] + u[0] * uvy) + cy) + a.prev_t[1]);
  t[2] = static_cast<float>(((ti[2] + u[0] * uvz) + cz) + a.prev_t[2]);
}

// Could the cast of v land on another float if v were off by up to `bound`?
__host__ __device__ inline bool cast_ambiguous(double v, double bound) {
  const float f = static_cast<float>(v);
  return static_cast<float>(v + bound) != f || static_cast<float>(v - bound) != f;
}

// The float part of one hit under the pose (q, t): hit / origin into the local frame and the range gate (:454-472).
// out_kind: 0 dropped (range < min_range), 1 return, 2 miss (beyond max_range: cropped ray end).
__device__ __forceinline__ void deskew_finish_hit(const DeskewArgs& a, const Quat4 q, float tx, float ty, float tz, const float4 h,
                                                  const float* __restrict__ in_origin, size_t ii, int i, int n,
                                                  float* __restrict__ out_x, float* __restrict__ out_y, float* __restrict__ out_z,
                                                  int out_stride, unsigned char* __restrict__ out_kind,
                                                  float* __restrict__ last_pose7) {
  float hx, hy, hz, ox, oy, oz;
  rotate_point(q, h.x, h.y, h.z, hx, hy, hz);
  hx += tx;
  hy += ty;
  hz += tz;
  // synchronized_data.origins.at(hits[i].origin_index) (:458-459); the index rides along as a float channel
  const int oi_ = in_origin != nullptr ? min(max(static_cast<int>(in_origin[ii]), 0), 3) : 0;
  rotate_point(q, a.origins[oi_][0], a.origins[oi_][1], a.origins[oi_][2], ox, oy, oz);
  ox += tx;
  oy += ty;
  oz += tz;
  const float dx = hx - ox, dy = hy - oy, dz = hz - oz;
  const float range = sqrtf(dx * dx + (dy * dy + dz * dz));
  unsigned char kind = 0;
  if (range >= a.min_range) {
    if (range <= a.max_range) {
      kind = 1;
    } else {
      kind = 2;
      const float f = a.max_range / range;
      hx = ox + f * dx;
      hy = oy + f * dy;
      hz = oz + f * dz;
    }
  }
  const size_t oi = static_cast<size_t>(i) * out_stride;
  out_x[oi] = hx;
  out_y[oi] = hy;
  out_z[oi] = hz;
  out_kind[i] = kind;
  if (i == n - 1) {  // current_pose = hits_poses.back() (:477)
    last_pose7[0] = tx;
    last_pose7[1] = ty;
    last_pose7[2] = tz;
    last_pose7[3] = q.w;
    last_pose7[4] = q.x;
    last_pose7[5] = q.y;
    last_pose7[6] = q.z;
  }
}

// One hit: pose_i = (prev * [s t_rel, slerp(I, q_rel, s)]).cast<float>() (:437-445,869-877), then
// hit/origin into the local frame and the range gate (:454-472).
// Inputs / outputs are strided so that packed host layouts (xyzt stride 4, xyz stride 3) and the
// device SoA layout (stride 1) run the same code.
// flags: the context's record buffer (kDeskewFlagWords words, layout above); only_flags: write nothing but the records
// (the overflow pass: `flags` is then [count | records] of room for every hit).
__global__ void deskew_kernel(DeskewArgs a, const float* __restrict__ in_x, const float* __restrict__ in_y,
                              const float* __restrict__ in_z, const float* __restrict__ in_t,
                              const float* __restrict__ in_origin, int in_stride, int n,
                              float* __restrict__ out_x, float* __restrict__ out_y, float* __restrict__ out_z,
                              int out_stride, unsigned char* __restrict__ out_kind,
                              unsigned* __restrict__ flags, int only_flags, const unsigned* __restrict__ n_dev) {
  if (n_dev != nullptr) n = min(n, static_cast<int>(*n_dev));  // the hits' number is still on the device (stage A)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t ii = static_cast<size_t>(i) * in_stride;
  const float4 h = make_float4(in_x[ii], in_y[ii], in_z[ii], in_t[ii]);
  Quat4 q;
  float tx, ty, tz;
  if (a.use_stamps) {
    double qq[4];
    float t3[3];
    bool libm_path;
    deskew_pose(a, h.w, qq, t3, &libm_path);
    tx = t3[0];
    ty = t3[1];
    tz = t3[2];
    q = Quat4{static_cast<float>(qq[0]), static_cast<float>(qq[1]), static_cast<float>(qq[2]),
              static_cast<float>(qq[3])};
    if (libm_path) {
      bool ambiguous = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) ambiguous = ambiguous || cast_ambiguous(qq[k], a.q_bound[k] + fabs(qq[k]) * a.norm_bound);
      if (ambiguous) {
        if (only_flags) {
          const unsigned r = atomicAdd(flags, 1u);
          unsigned* rec = flags + 1 + 6u * r;
          rec[0] = static_cast<unsigned>(i) | (a.launch_tag << 27);
          rec[1] = __float_as_uint(h.w);
          rec[2] = __float_as_uint(q.w);
          rec[3] = __float_as_uint(q.x);
          rec[4] = __float_as_uint(q.y);
          rec[5] = __float_as_uint(q.z);
        } else {
          const unsigned r = atomicAdd(flags + 8, 1u);
          unsigned* rec = flags + 16 + 6u * (r % static_cast<unsigned>(kDeskewRing));
          rec[0] = static_cast<unsigned>(i) | (a.launch_tag << 27);
          rec[1] = __float_as_uint(h.w);
          rec[2] = __float_as_uint(q.w);
          rec[3] = __float_as_uint(q.x);
          rec[4] = __float_as_uint(q.y);
          rec[5] = __float_as_uint(q.z);
        }
      }
    }
    if (only_flags) return;
  } else {
    if (only_flags) return;
    q = Quat4{a.cur_q[0], a.cur_q[1], a.cur_q[2], a.cur_q[3]};
    tx = a.cur_t[0];
    ty = a.cur_t[1];
    tz = a.cur_t[2];
  }
  deskew_finish_hit(a, q, tx, ty, tz, h, in_origin, ii, i, n, out_x, out_y, out_z, out_stride, out_kind,
                    reinterpret_cast<float*>(flags));
}

// The hits the host's check found different (never observed; glibc's value is the reference's by definition): redone
// with the host's quaternion.  fixes: (hit index, bits of the four floats) x num_fixes.
__global__ void deskew_fix_kernel(DeskewArgs a, const float* __restrict__ in_x, const float* __restrict__ in_y,
                                  const float* __restrict__ in_z, const float* __restrict__ in_t,
                                  const float* __restrict__ in_origin, int in_stride, int n, float* __restrict__ out_x,
                                  float* __restrict__ out_y, float* __restrict__ out_z, int out_stride,
                                  unsigned char* __restrict__ out_kind, float* __restrict__ last_pose7,
                                  const unsigned* __restrict__ fixes, int num_fixes) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= num_fixes) return;
  const int i = static_cast<int>(fixes[5 * f]);
  if (i < 0 || i >= n) return;
  const size_t ii = static_cast<size_t>(i) * in_stride;
  const float4 h = make_float4(in_x[ii], in_y[ii], in_z[ii], in_t[ii]);
  double qq[4];
  float t3[3];
  bool libm_path;
  deskew_pose(a, h.w, qq, t3, &libm_path);  // (the translation has no sin / acos in it: identical on host and device)
  const Quat4 q{__uint_as_float(fixes[5 * f + 1]), __uint_as_float(fixes[5 * f + 2]), __uint_as_float(fixes[5 * f + 3]),
                __uint_as_float(fixes[5 * f + 4])};
  deskew_finish_hit(a, q, t3[0], t3[1], t3[2], h, in_origin, ii, i, n, out_x, out_y, out_z, out_stride, out_kind, last_pose7);
}

__global__ void split_xyzt_kernel(const float4* __restrict__ aos, int n, float* __restrict__ x,
                                  float* __restrict__ y, float* __restrict__ z, float* __restrict__ t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = aos[i];
  x[i] = p.x;
  y[i] = p.y;
  z[i] = p.z;
  t[i] = p.w;
}

// sensor::TransformPointCloud in float (sensor/point_cloud.cc:25-33): rotation * p + translation.
// partial_max[4 * block + k]: per workgroup the largest squared norm (k = 0: the cloud's max ||p||) and the largest
// |x|, |y|, |z| (bounds only: grid.hip proves "this insertion cannot leave the grid's extent" from them).  No atomics:
// one atomicMax per point on one word was the whole kernel (11 us for 46 k points; four words at one per wavefront made
// it 23 -- every one of them serialises at the memory side); the host takes the maximum of <= kTransformBlocks partials.
constexpr int kTransformBlocks = 128;
// A maximum that a NaN wins and keeps (fmaxf drops it): a cloud with a non-finite coordinate must end with non-finite
// bounds, so that grid.hip's insertion_provably_inside() refuses to prove anything about it.
__host__ __device__ __forceinline__ float max_nan_wins(float a, float b) { return (b > a || b != b) ? b : a; }
__global__ __launch_bounds__(256) void transform_kernel(Quat4 q, float tx, float ty, float tz, const float* __restrict__ x,
                                                        const float* __restrict__ y, const float* __restrict__ z, int n,
                                                        float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz,
                                                        float* __restrict__ partial_max, const unsigned* __restrict__ n_dev) {
  if (n_dev != nullptr) n = min(n, static_cast<int>(*n_dev));  // the number of points is still on the device (stage B)
  float m[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float rx, ry, rz;
    rotate_point(q, x[i], y[i], z[i], rx, ry, rz);
    rx += tx;
    ry += ty;
    rz += tz;
    ox[i] = rx;
    oy[i] = ry;
    oz[i] = rz;
    m[0] = max_nan_wins(m[0], rx * rx + (ry * ry + rz * rz));
    m[1] = max_nan_wins(m[1], fabsf(rx));
    m[2] = max_nan_wins(m[2], fabsf(ry));
    m[3] = max_nan_wins(m[3], fabsf(rz));
  }
  __shared__ float part[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m[k] = max_nan_wins(m[k], __shfl_xor(m[k], off, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][k] = m[k];
  }
  __syncthreads();
  if (threadIdx.x < 4)
    partial_max[4 * blockIdx.x + threadIdx.x] = max_nan_wins(max_nan_wins(part[0][threadIdx.x], part[1][threadIdx.x]),
                                                             max_nan_wins(part[2][threadIdx.x], part[3][threadIdx.x]));
}

static int make_deskew_args(const double prev_pose[7], const double predicted_pose[7], double scan_period,
                            const float origin[3], float min_range, float max_range, float first_time,
                            DeskewArgs* a) {
  PoseD prev, cur;
  for (int i = 0; i < 3; ++i) {
    prev.t[i] = prev_pose[i];
    cur.t[i] = predicted_pose[i];
  }
  for (int i = 0; i < 4; ++i) {
    prev.q[i] = prev_pose[3 + i];
    cur.q[i] = predicted_pose[3 + i];
  }
  const PoseD rel = pose_mul(pose_inverse(prev), cur);  // :427
  std::memcpy(a->prev_t, prev.t, sizeof(a->prev_t));
  std::memcpy(a->prev_q, prev.q, sizeof(a->prev_q));
  std::memcpy(a->rel_t, rel.t, sizeof(a->rel_t));
  std::memcpy(a->rel_q, rel.q, sizeof(a->rel_q));
  float cf[7];
  pose_to_float7(cur, cf);
  std::memcpy(a->cur_t, cf, 12);
  std::memcpy(a->cur_q, cf + 3, 16);
  a->scan_period = scan_period;
  a->ox = origin[0];
  a->oy = origin[1];
  a->oz = origin[2];
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 3; ++i) a->origins[k][i] = origin[i];
  a->min_range = min_range;
  a->max_range = max_range;
  a->use_stamps = std::abs(first_time) < 1e-3 ? 0 : 1;  // hits.front().point_time[3] (:429)
  // ---- how far the device's quaternion can be from glibc's (DeskewArgs::q_bound).  Assumptions, both documented
  // library bounds: the device's sin / acos are within 4 ulp (what OpenCL's full profile requires of double sin and acos
  // and OCML is built to), glibc's within 1 ulp.  Two results for the same argument then differ by <= e = 5 * 2^-52
  // relative.  Through deskew_pose, every other operation being a correctly rounded IEEE operation on both sides (two
  // different inputs round to results at most |input difference| + 1 ulp apart):
  //   theta: e;  sin(theta): <= 2 e (theta cot theta <= 1 on (0, pi/2]);  sin(a theta), a in [0, 1]: <= 2 e + 2^-52
  //   scale0, scale1 (quotients, both in [0, 1]):  es = 4 e + 2^-51
  //   qi[0] = scale0 + scale1 d: es (1 + |d|) + 2^-52;   qi[m] = scale1 q_rel[m]: (es + 2^-53) |q_rel[m]|
  //   qq = prev.q * qi (Hamilton product, |prev.q| ~ 1): sum_j |prev.q[j]| dqi[perm] + 7 roundings of values <= 2
  //   normalisation: z2 relative 2 * 2 max_k(db) + 2^-50, its square root half of that, the division one more ulp
  {
    // (e: 5 ulp by the documented bounds; doubled as a margin against an argument range where the shipped OCML is worse
    //  than its specification -- a few more records per scan)
    const double ulp = std::ldexp(1.0, -52), e = 2.0 * 5.0 * ulp, es = 4.0 * e + 2.0 * ulp;
    const double* r = a->rel_q;
    const double dq[4] = {es * (1.0 + std::fabs(r[0])) + ulp, (es + 0.5 * ulp) * std::fabs(r[1]), (es + 0.5 * ulp) * std::fabs(r[2]),
                          (es + 0.5 * ulp) * std::fabs(r[3])};
    const double p0 = std::fabs(a->prev_q[0]), p1 = std::fabs(a->prev_q[1]), p2 = std::fabs(a->prev_q[2]), p3 = std::fabs(a->prev_q[3]);
    const double slack = 7.0 * 2.0 * ulp;
    a->q_bound[0] = p0 * dq[0] + p1 * dq[1] + p2 * dq[2] + p3 * dq[3] + slack;
    a->q_bound[1] = p0 * dq[1] + p1 * dq[0] + p2 * dq[3] + p3 * dq[2] + slack;
    a->q_bound[2] = p0 * dq[2] + p1 * dq[3] + p2 * dq[0] + p3 * dq[1] + slack;
    a->q_bound[3] = p0 * dq[3] + p1 * dq[2] + p2 * dq[1] + p3 * dq[0] + slack;
    const double mx = std::max(std::max(a->q_bound[0], a->q_bound[1]), std::max(a->q_bound[2], a->q_bound[3]));
    a->norm_bound = 2.0 * mx + 4.0 * ulp;
    const double pn = p0 * p0 + p1 * p1 + p2 * p2 + p3 * p3;
    if (!(pn > 0.99 && pn < 1.01) || !std::isfinite(mx)) {  // not a pose the bound was derived for: every hit is checked
      for (double& b : a->q_bound) b = 1.0;
      a->norm_bound = 1.0;
    }
  }
  return DLIOM_OK;
}

// Per launch: the record tag; and in libdliom_hooks.so only (knob 2 of dliom_ctx_set_tuning = 2 or 3): bounds so wide that EVERY hit is recorded -- the
// ring overflows and the records-only pass over all hits runs; with 3 every record is also "fixed" (with the host's own
// floats, equal to the device's): the tests walk both paths and the results must not change.
static void deskew_test_hook(dliom_ctx* ctx, DeskewArgs* a) {
  a->launch_tag = (ctx->deskew_launch_id++) & 31u;
#ifdef DLIOM_TEST_HOOKS
  if (ctx->tuning[DLIOM_TUNE_RESERVED_TEST_HOOK] >= 2) {
    for (double& b : a->q_bound) b = 1.0;
    a->norm_bound = 1.0;
  }
#else
  (void)ctx;
  (void)a;
#endif
}

// The check itself (DeskewArgs::q_bound): every recorded hit once more on the HOST -- deskew_pose with glibc's sin and
// acos, the reference's own -- against the floats the device cast.  Returns the hits that differ as fix records (hit
// index, bits of the four host floats) in *fixes.  recs: `count` records of 6 words.
static void check_deskew_records(const DeskewArgs& a, const unsigned* recs, size_t count, std::vector<unsigned>* fixes,
                                 bool force_fix = false) {
  for (size_t r = 0; r < count; ++r) {
    const unsigned* rec = recs + 6 * r;
    if ((rec[0] >> 27) != a.launch_tag) continue;  // another launch's record (a call that failed before its check)
    float t_rel;
    std::memcpy(&t_rel, rec + 1, 4);
    double qq[4];
    float t3[3];
    bool libm_path;
    deskew_pose(a, t_rel, qq, t3, &libm_path);
    unsigned bits[4];
    bool same = true;
    for (int k = 0; k < 4; ++k) {
      const float f = static_cast<float>(qq[k]);
      std::memcpy(&bits[k], &f, 4);
      same = same && bits[k] == rec[2 + k];
    }
    if (!same || force_fix) {  // (force_fix: libdliom_hooks.so only -- walks the fix path with the host's own floats)
      fixes->push_back(rec[0] & 0x7FFFFFFu);
      for (int k = 0; k < 4; ++k) fixes->push_back(bits[k]);
    }
  }
}

// The context's record buffer (kDeskewFlagWords words, zeroed once; the counter in it only ever grows).
static int deskew_flag_buffer(dliom_ctx* ctx, unsigned** out) {
  if (ctx->deskew_flags.p == nullptr) {
    DLIOM_TRY(ctx->deskew_flags.reserve(4 * kDeskewFlagWords));
    DLIOM_HIP_TRY(hipMemsetAsync(ctx->deskew_flags.p, 0, 4 * kDeskewFlagWords, ctx->stream));
    ctx->deskew_flag_total = 0;
  }
  *out = ctx->deskew_flags.as<unsigned>();
  return DLIOM_OK;
}

// After a de-skew launch whose record buffer `host_flags` (kDeskewFlagWords words) is back on the host: checks the
// recorded hits against glibc, falls back to a records-only pass over every hit when the ring overflowed, and redoes the
// hits that differ with the host's quaternion (deskew_fix_kernel).  *fixed: outputs were rewritten (the caller redoes
// what it derived from them).  Synchronises only on the paths that have something to fix or recount.
struct DeskewLaunch {
  const float *in_x, *in_y, *in_z, *in_t, *in_origin;
  int in_stride, n;
  float *out_x, *out_y, *out_z;
  int out_stride;
  unsigned char* out_kind;
};
static int verify_deskew(dliom_ctx* ctx, const DeskewArgs& a, const DeskewLaunch& l, const unsigned* host_flags, unsigned* d_flags,
                         bool* fixed) {
  *fixed = false;
  if (!a.use_stamps) return DLIOM_OK;
#ifdef DLIOM_TEST_HOOKS
  const bool force_fix = ctx->tuning[DLIOM_TUNE_RESERVED_TEST_HOOK] == 3;
#else
  const bool force_fix = false;
#endif
  const unsigned total = host_flags[8];
  const unsigned k = total - ctx->deskew_flag_total;  // records of this launch (the counter wraps like the subtraction)
  const unsigned first = ctx->deskew_flag_total;
  ctx->deskew_flag_total = total;
  ctx->deskew_records_checked += k;
  if (k == 0) return DLIOM_OK;
  std::vector<unsigned> fixes;
  if (k <= static_cast<unsigned>(kDeskewRing)) {
    std::vector<unsigned> recs(6 * static_cast<size_t>(k));
    for (unsigned r = 0; r < k; ++r)
      std::memcpy(&recs[6 * r], host_flags + 16 + 6 * ((first + r) % static_cast<unsigned>(kDeskewRing)), 24);
    check_deskew_records(a, recs.data(), k, &fixes, force_fix);
  } else {
    // more records than the ring holds (orientations with tiny quaternion components make many casts borderline): a
    // records-only pass over every hit into a list with room for all of them, read back in full
    ++ctx->deskew_overflows;
    const size_t words = 1 + 6 * static_cast<size_t>(l.n);
    DLIOM_TRY(ctx->sort_tmp.reserve(4 * words));
    unsigned* d_list = ctx->sort_tmp.as<unsigned>();
    DLIOM_HIP_TRY(hipMemsetAsync(d_list, 0, 4, ctx->stream));
    hipLaunchKernelGGL(deskew_kernel, dim3(static_cast<unsigned>((l.n + 255) / 256)), dim3(256), 0, ctx->stream, a, l.in_x, l.in_y,
                       l.in_z, l.in_t, l.in_origin, l.in_stride, l.n, l.out_x, l.out_y, l.out_z, l.out_stride, l.out_kind, d_list, 1,
                       static_cast<const unsigned*>(nullptr));
    DLIOM_HIP_TRY(hipGetLastError());
    unsigned count = 0;
    DLIOM_HIP_TRY(hipMemcpyAsync(&count, d_list, 4, hipMemcpyDeviceToHost, ctx->stream));
    DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<unsigned> recs(6 * static_cast<size_t>(count));
    if (count > 0) {
      DLIOM_HIP_TRY(hipMemcpy(recs.data(), d_list + 1, 24 * static_cast<size_t>(count), hipMemcpyDeviceToHost));
      check_deskew_records(a, recs.data(), count, &fixes, force_fix);
    }
  }
  if (fixes.empty()) return DLIOM_OK;
  // never observed: the device's cast differs from glibc's for these hits -- glibc's is the reference's
  const int nf = static_cast<int>(fixes.size() / 5);
  ctx->deskew_fixed_hits += nf;
  DLIOM_TRY(ctx->sort_tmp.reserve(fixes.size() * 4));
  DLIOM_HIP_TRY(hipMemcpyAsync(ctx->sort_tmp.p, fixes.data(), fixes.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(deskew_fix_kernel, dim3(static_cast<unsigned>((nf + 63) / 64)), dim3(64), 0, ctx->stream, a, l.in_x, l.in_y, l.in_z,
                     l.in_t, l.in_origin, l.in_stride, l.n, l.out_x, l.out_y, l.out_z, l.out_stride, l.out_kind,
                     reinterpret_cast<float*>(d_flags), ctx->sort_tmp.as<unsigned>(), nf);
  DLIOM_HIP_TRY(hipGetLastError());
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));  // `fixes` dies with this frame
  *fixed = true;
  return DLIOM_OK;
}

}  // namespace dliom

using namespace dliom;

namespace dliom {

// AddRangeData up to the accumulation (:393-472): VoxelFilter(0.5 size) on the timed hits, per-hit de-skew + range
// gate, the returns compacted in hit order into ctx->misc (x | y | z with stride *stride).  origin_index (one float
// per range, may be null) and origins (num_origins x 3) are the RangeDataSynchronizer's origin table.
static int add_range_data_stage_a(dliom_ctx* ctx, const double prev_pose[7], const double predicted_pose[7],
                                  double scan_period, const float* ranges_xyzt, const float* origin_index, int64_t n,
                                  const float* origins, int num_origins, float min_range, float max_range,
                                  float voxel_filter_size, float current_pose[7], const float** returns,
                                  size_t* stride, int64_t* num_returns) {
  if (n >= (int64_t{1} << 27)) return DLIOM_ERR_INVALID_ARGUMENT;  // (hit indices share a word with the records' launch tag)
  const size_t nn = static_cast<size_t>(n);
  auto al = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  // scratch: raw AoS | raw SoA (4) | filtered hits SoA (4) | de-skewed SoA (3) + kind | returns (3) | origin index in
  // | origin index filtered | pose
  const size_t off_raw = 0, off_b = al(16 * nn), off_c = off_b + al(16 * nn), off_d = off_c + al(16 * nn),
               off_kind = off_d + al(12 * nn), off_e = off_kind + al(nn), off_oi = off_e + al(12 * nn),
               off_of = off_oi + al(4 * nn), off_pose = off_of + al(4 * nn), total = off_pose + 256;
  DLIOM_TRY(ctx->misc.reserve(total));
  char* base = static_cast<char*>(ctx->misc.p);
  float* b = reinterpret_cast<float*>(base + off_b);
  float* c = reinterpret_cast<float*>(base + off_c);
  float* d = reinterpret_cast<float*>(base + off_d);
  unsigned char* kind = reinterpret_cast<unsigned char*>(base + off_kind);
  float* e = reinterpret_cast<float*>(base + off_e);
  float* oi_in = reinterpret_cast<float*>(base + off_oi);
  float* oi_f = reinterpret_cast<float*>(base + off_of);
  (void)off_pose;
  unsigned* d_flags = nullptr;  // [last hit's pose | record counter | records]: persistent, read back with the compaction's counts
  DLIOM_TRY(deskew_flag_buffer(ctx, &d_flags));
  const int threads = 256;
  DLIOM_HIP_TRY(hipMemcpyAsync(base + off_raw, ranges_xyzt, 16 * nn, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(split_xyzt_kernel, dim3(static_cast<unsigned>((n + threads - 1) / threads)), dim3(threads), 0,
                     ctx->stream, reinterpret_cast<const float4*>(base + off_raw), static_cast<int>(n), b, b + nn,
                     b + 2 * nn, b + 3 * nn);
  // hits = VoxelFilter(0.5f * voxel_filter_size).Filter(ranges): the time rides along
  const bool multi = origin_index != nullptr && num_origins > 1;
  DeskewArgs a;
  // the first range always survives the filter, so hits.front() is ranges.front()
  DLIOM_TRY(make_deskew_args(prev_pose, predicted_pose, scan_period, origins, min_range, max_range, ranges_xyzt[3], &a));
  deskew_test_hook(ctx, &a);
  for (int k = 0; k < std::min(num_origins, 4); ++k)
    for (int i = 0; i < 3; ++i) a.origins[k][i] = origins[3 * k + i];
  std::vector<unsigned> host_flags(kDeskewFlagWords);
  int64_t n1 = -1;
  // Round 5: ONE read-back for the stage (single origin).  The filter is only enqueued; the de-skew and the compaction
  // of the returns run over the INPUT's size and take the hits' number from the device word the filter's compaction
  // leaves; that number, "fits the packed table words", the number of returns, the last hit's pose and the de-skew's
  // records come back together.  (Until then: the filter's count read-back, then the compaction's.)
  if (!multi) {
    const unsigned *d_hits = nullptr, *d_unpackable = nullptr, *d_returns = nullptr;
    const int fst = voxel_filter_arrays_enqueue(ctx, Soa{b, b + nn, b + 2 * nn, b + 3 * nn, n}, 0.5f * voxel_filter_size, c, c + nn,
                                                c + 2 * nn, c + 3 * nn, &d_hits, &d_unpackable);
    if (fst == DLIOM_OK) {
      hipLaunchKernelGGL(deskew_kernel, dim3(static_cast<unsigned>((n + threads - 1) / threads)), dim3(threads), 0, ctx->stream, a, c,
                         c + nn, c + 2 * nn, c + 3 * nn, static_cast<const float*>(nullptr), 1, static_cast<int>(n), d, d + nn,
                         d + 2 * nn, 1, kind, d_flags, 0, d_hits);
      DLIOM_HIP_TRY(hipGetLastError());
      DLIOM_TRY(compact_equal_arrays_enqueue(ctx, Soa{d, d + nn, d + 2 * nn, nullptr, n}, kind, 1, e, e + nn, e + 2 * nn, d_hits, &d_returns));
      unsigned* host = static_cast<unsigned*>(ctx->pinned);
      const GatherJob jobs[4] = {{d_hits, 1}, {d_unpackable, 1}, {d_returns, 1}, {d_flags, static_cast<unsigned>(kDeskewFlagWords)}};
      DLIOM_TRY(gather_and_wait(ctx, jobs, 4, host));
      if (host[1] == 0u) {
        n1 = host[0];
        *num_returns = host[2];
        std::memcpy(host_flags.data(), host + 3, 4 * static_cast<size_t>(kDeskewFlagWords));
      } else {
        ++ctx->voxel_unpacked_reruns;  // a range farther than 4095 voxel edges away: the general path below
        // (the de-skew's records of the abandoned launch stay counted: verify_deskew takes the counter's difference)
        ctx->deskew_flag_total = host[3 + 8];
      }
    } else if (fst != DLIOM_ERR_CAPACITY) {
      return fst;
    }
  }
  const bool merged = n1 >= 0;
  if (!merged) {
    DLIOM_TRY(voxel_filter_arrays(ctx, Soa{b, b + nn, b + 2 * nn, b + 3 * nn, n}, 0.5f * voxel_filter_size, c, c + nn,
                                  c + 2 * nn, c + 3 * nn, &n1));
    if (multi) {  // the same filter once more with the origin index as the passenger: same survivors, same order
      DLIOM_HIP_TRY(hipMemcpyAsync(oi_in, origin_index, 4 * nn, hipMemcpyHostToDevice, ctx->stream));
      int64_t n1b = 0;
      DLIOM_TRY(voxel_filter_arrays(ctx, Soa{b, b + nn, b + 2 * nn, oi_in, n}, 0.5f * voxel_filter_size, d, d + nn,
                                    d + 2 * nn, oi_f, &n1b));
      if (n1b != n1) return DLIOM_ERR_INVALID_ARGUMENT;
    }
  }
  const DeskewLaunch launch{c, c + nn, c + 2 * nn, c + 3 * nn, multi ? oi_f : static_cast<const float*>(nullptr), 1, static_cast<int>(n1),
                            d, d + nn, d + 2 * nn, 1, kind};
  if (!merged) {
    hipLaunchKernelGGL(deskew_kernel, dim3(static_cast<unsigned>((n1 + threads - 1) / threads)), dim3(threads), 0,
                       ctx->stream, a, launch.in_x, launch.in_y, launch.in_z, launch.in_t, launch.in_origin, 1, launch.n, d, d + nn,
                       d + 2 * nn, 1, kind, d_flags, 0, static_cast<const unsigned*>(nullptr));
    DLIOM_HIP_TRY(hipGetLastError());
    // returns (kind 1), in hit order; misses (kind 2) are not used by the 3D path.  The compaction's read-back brings the
    // last hit's pose and the de-skew's records along (one round trip, no memcpy)
    DLIOM_TRY(compact_equal_arrays(ctx, Soa{d, d + nn, d + 2 * nn, nullptr, n1}, kind, 1, e, e + nn, e + 2 * nn, num_returns, d_flags,
                                   kDeskewFlagWords, host_flags.data()));
  }
  bool fixed = false;
  DLIOM_TRY(verify_deskew(ctx, a, launch, host_flags.data(), d_flags, &fixed));
  if (fixed)  // hits were redone with the host's quaternion: compact (and read the pose) once more
    DLIOM_TRY(compact_equal_arrays(ctx, Soa{d, d + nn, d + 2 * nn, nullptr, n1}, kind, 1, e, e + nn, e + 2 * nn, num_returns, d_flags,
                                   kDeskewFlagWords, host_flags.data()));
  std::memcpy(current_pose, host_flags.data(), 28);
  *returns = e;
  *stride = nn;
  return DLIOM_OK;
}

// :476-487: VoxelFilter(size) over the accumulated returns (local frame), then into the tracking frame of
// current_pose (TransformRangeData(., current_pose.inverse())), as a device cloud.
static int add_range_data_stage_b(dliom_ctx* ctx, const float* rx, const float* ry, const float* rz, int64_t n2,
                                  float voxel_filter_size, const float current_pose[7], dliom_cloud** returns_in_tracking,
                                  float origin_in_tracking[3]) {
  const size_t nn = static_cast<size_t>(std::max<int64_t>(n2, 1));
  auto al = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  // filtered returns | per-workgroup maxima | transformed returns (misc holds the inputs)
  DLIOM_TRY(ctx->rescore.reserve(6 * al(4 * nn) + 16 * kTransformBlocks));
  float* f = ctx->rescore.as<float>();
  float* d_partial_max = reinterpret_cast<float*>(static_cast<char*>(ctx->rescore.p) + 3 * al(4 * nn));
  float* t = reinterpret_cast<float*>(static_cast<char*>(ctx->rescore.p) + 3 * al(4 * nn) + 16 * kTransformBlocks);
  const size_t fs = al(4 * nn) / 4;
  // current_pose.inverse() in float (rigid_transform.h:167-171)
  const QF qc{current_pose[3], -current_pose[4], -current_pose[5], -current_pose[6]};
  const F3 rt = qrot(qc, F3{current_pose[0], current_pose[1], current_pose[2]});
  const F3 ti{-rt.x, -rt.y, -rt.z};
  const F3 o = add3(qrot(qc, F3{current_pose[0], current_pose[1], current_pose[2]}), ti);  // inverse * origin
  origin_in_tracking[0] = o.x;
  origin_in_tracking[1] = o.y;
  origin_in_tracking[2] = o.z;
  const int threads = 256;
  float* host = static_cast<float*>(ctx->pinned);
  float max_norm = 0.f, abs_max[3] = {0.f, 0.f, 0.f};
  int64_t n3 = -1;
  // Round 5: ONE read-back for the stage.  The filter is only enqueued (its survivor count stays on the device), the
  // transform takes the count from there and writes into scratch arrays of the input's size, and count, "fits the packed
  // table words" and the transform's maxima come back together; the cloud is allocated then and filled by the launch
  // that pads its tail anyway.  (Until then: count read-back, transform, maxima read-back.)
  if (n2 > 0) {
    const unsigned *d_total = nullptr, *d_unpackable = nullptr;
    const int fst = voxel_filter_arrays_enqueue(ctx, Soa{rx, ry, rz, nullptr, n2}, voxel_filter_size, f, f + fs, f + 2 * fs, nullptr,
                                                &d_total, &d_unpackable);
    if (fst == DLIOM_OK) {
      const unsigned blocks = static_cast<unsigned>(std::min<int64_t>((n2 + threads - 1) / threads, kTransformBlocks));
      hipLaunchKernelGGL(transform_kernel, dim3(blocks), dim3(threads), 0, ctx->stream, Quat4{qc.w, qc.x, qc.y, qc.z}, ti.x, ti.y,
                         ti.z, f, f + fs, f + 2 * fs, static_cast<int>(n2), t, t + fs, t + 2 * fs, d_partial_max, d_total);
      const GatherJob jobs[3] = {{d_total, 1}, {d_unpackable, 1}, {d_partial_max, 4 * blocks}};
      DLIOM_TRY(gather_and_wait(ctx, jobs, 3, host));
      unsigned head[2];
      std::memcpy(head, host, 8);
      if (head[1] == 0u) {
        n3 = head[0];
        float sq = 0.f;
        for (unsigned b = 0; b < blocks; ++b) {
          sq = max_nan_wins(sq, host[2 + 4 * b]);
          for (int a = 0; a < 3; ++a) abs_max[a] = max_nan_wins(abs_max[a], host[2 + 4 * b + 1 + a]);
        }
        max_norm = std::sqrt(sq);  // sqrt is monotone and correctly rounded: == the maximum of the norms
      } else {
        ++ctx->voxel_unpacked_reruns;  // a return farther than 4095 voxel edges away: the general filter below
      }
    } else if (fst != DLIOM_ERR_CAPACITY) {
      return fst;
    }
  } else {
    n3 = 0;
  }
  if (n3 >= 0) {
    float *cx, *cy, *cz;  // (filled from the scratch arrays by finish_device_cloud_from)
    DLIOM_TRY(alloc_device_cloud(ctx, n3, returns_in_tracking, &cx, &cy, &cz));
    int st = finish_device_cloud_from(ctx, *returns_in_tracking, max_norm, t, t + fs, t + 2 * fs);
    if (st == DLIOM_OK && n3 > 0)
      for (int a = 0; a < 3; ++a) (*returns_in_tracking)->abs_max[a] = abs_max[a];
    if (st != DLIOM_OK) {
      dliom_cloud_destroy(*returns_in_tracking);
      *returns_in_tracking = nullptr;
    }
    return st;
  }
  // ---- the general path: read-backs after the filter and after the transform
  DLIOM_TRY(voxel_filter_arrays(ctx, Soa{rx, ry, rz, nullptr, n2}, voxel_filter_size, f, f + fs, f + 2 * fs, nullptr, &n3));
  float *ox, *oy, *oz;
  DLIOM_TRY(alloc_device_cloud(ctx, n3, returns_in_tracking, &ox, &oy, &oz));
  int st = DLIOM_OK;
  if (n3 > 0) {
    const unsigned blocks = static_cast<unsigned>(std::min<int64_t>((n3 + threads - 1) / threads, kTransformBlocks));
    hipLaunchKernelGGL(transform_kernel, dim3(blocks), dim3(threads), 0, ctx->stream, Quat4{qc.w, qc.x, qc.y, qc.z}, ti.x, ti.y,
                       ti.z, f, f + fs, f + 2 * fs, static_cast<int>(n3), ox, oy, oz, d_partial_max, static_cast<const unsigned*>(nullptr));
    const GatherJob job{d_partial_max, 4 * blocks};
    st = gather_and_wait(ctx, &job, 1, host);
    if (st == DLIOM_OK) {
      float sq = 0.f;
      for (unsigned b = 0; b < blocks; ++b) {
        sq = max_nan_wins(sq, host[4 * b]);
        for (int a = 0; a < 3; ++a) abs_max[a] = max_nan_wins(abs_max[a], host[4 * b + 1 + a]);
      }
      max_norm = std::sqrt(sq);
    }
  }
  if (st == DLIOM_OK) st = finish_device_cloud(ctx, *returns_in_tracking, max_norm);
  if (st == DLIOM_OK && n3 > 0)
    for (int a = 0; a < 3; ++a) (*returns_in_tracking)->abs_max[a] = abs_max[a];
  if (st != DLIOM_OK) {
    dliom_cloud_destroy(*returns_in_tracking);
    *returns_in_tracking = nullptr;
  }
  return st;
}

}  // namespace dliom

extern "C" int dliom_add_range_data(dliom_ctx* ctx, const double prev_pose[7], const double predicted_pose[7],
                                    double scan_period, const float* ranges_xyzt, int64_t n, const float origin[3],
                                    float min_range, float max_range, float voxel_filter_size,
                                    dliom_cloud** returns_in_tracking, float origin_in_tracking[3],
                                    float current_pose[7]) {
  if (ctx == nullptr || prev_pose == nullptr || predicted_pose == nullptr || origin == nullptr || n < 0 ||
      returns_in_tracking == nullptr || origin_in_tracking == nullptr || current_pose == nullptr ||
      (n > 0 && ranges_xyzt == nullptr) || !(scan_period > 0.) || !(voxel_filter_size > 0.f))
    return DLIOM_ERR_INVALID_ARGUMENT;
  *returns_in_tracking = nullptr;
  if (n == 0) return DLIOM_ERR_EMPTY_CLOUD;  // CHECK(!synchronized_data.ranges.empty()) (:383)
  if (n > (1 << 30)) return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  const float* e = nullptr;
  size_t stride = 0;
  int64_t n2 = 0;
  DLIOM_TRY(add_range_data_stage_a(ctx, prev_pose, predicted_pose, scan_period, ranges_xyzt, nullptr, n, origin, 1,
                                   min_range, max_range, voxel_filter_size, current_pose, &e, &stride, &n2));
  return add_range_data_stage_b(ctx, e, e + stride, e + 2 * stride, n2, voxel_filter_size, current_pose,
                                returns_in_tracking, origin_in_tracking);
}

// ---- num_accumulated_range_data > 1 (:449-476): several AddRangeData calls feed one AddAccumulatedRangeData --------
struct dliom_range_accumulator {
  dliom_ctx* ctx = nullptr;
  dliom::DevBuf points;  // x | y | z, capacity `cap` each
  size_t cap = 0;
  int64_t count = 0;
  int num_accumulated = 0;
  float current_pose[7] = {0, 0, 0, 1, 0, 0, 0};
};

extern "C" int dliom_range_accumulator_create(dliom_ctx* ctx, dliom_range_accumulator** out) {
  if (ctx == nullptr || out == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  dliom_range_accumulator* a = new dliom_range_accumulator;
  a->ctx = ctx;
  *out = a;
  return DLIOM_OK;
}

extern "C" int dliom_range_accumulator_destroy(dliom_range_accumulator* a) {
  if (a == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  a->points.release();
  delete a;
  return DLIOM_OK;
}

extern "C" int dliom_range_accumulator_add(dliom_range_accumulator* a, const double prev_pose[7],
                                           const double predicted_pose[7], double scan_period, const float* ranges_xyzt,
                                           const float* origin_index, int64_t n, const float* origins, int num_origins,
                                           float min_range, float max_range, float voxel_filter_size,
                                           float current_pose[7], int* num_accumulated) {
  if (a == nullptr || prev_pose == nullptr || predicted_pose == nullptr || origins == nullptr || num_origins < 1 ||
      num_origins > 4 || n < 0 || current_pose == nullptr || (n > 0 && ranges_xyzt == nullptr) || !(scan_period > 0.) ||
      !(voxel_filter_size > 0.f))
    return DLIOM_ERR_INVALID_ARGUMENT;
  if (n == 0) return DLIOM_ERR_EMPTY_CLOUD;
  if (n > (1 << 30)) return DLIOM_ERR_INVALID_ARGUMENT;
  dliom_ctx* ctx = a->ctx;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  const float* e = nullptr;
  size_t stride = 0;
  int64_t n2 = 0;
  DLIOM_TRY(add_range_data_stage_a(ctx, prev_pose, predicted_pose, scan_period, ranges_xyzt, origin_index, n, origins,
                                   num_origins, min_range, max_range, voxel_filter_size, current_pose, &e, &stride, &n2));
  const size_t need = static_cast<size_t>(a->count + n2);
  if (need > a->cap) {  // grow, keeping what is there
    const size_t new_cap = std::max<size_t>(need * 2, 4096);
    dliom::DevBuf grown;
    DLIOM_TRY(grown.reserve(new_cap * 12));
    float* g = grown.as<float>();
    const float* old = a->points.as<float>();
    for (int k = 0; k < 3 && a->count > 0; ++k)
      DLIOM_HIP_TRY(hipMemcpyAsync(g + k * new_cap, old + k * a->cap, static_cast<size_t>(a->count) * 4,
                                   hipMemcpyDeviceToDevice, ctx->stream));
    DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    a->points.release();
    a->points = grown;
    a->cap = new_cap;
  }
  float* dst = a->points.as<float>();
  for (int k = 0; k < 3 && n2 > 0; ++k)
    DLIOM_HIP_TRY(hipMemcpyAsync(dst + k * a->cap + a->count, e + k * stride, static_cast<size_t>(n2) * 4,
                                 hipMemcpyDeviceToDevice, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));  // ctx->misc is reused by the next call
  a->count += n2;
  ++a->num_accumulated;
  std::memcpy(a->current_pose, current_pose, sizeof a->current_pose);
  if (num_accumulated != nullptr) *num_accumulated = a->num_accumulated;
  return DLIOM_OK;
}

extern "C" int dliom_range_accumulator_finish(dliom_range_accumulator* a, float voxel_filter_size,
                                              dliom_cloud** returns_in_tracking, float origin_in_tracking[3]) {
  if (a == nullptr || returns_in_tracking == nullptr || origin_in_tracking == nullptr || !(voxel_filter_size > 0.f) ||
      a->num_accumulated == 0)
    return DLIOM_ERR_INVALID_ARGUMENT;
  *returns_in_tracking = nullptr;
  DLIOM_HIP_TRY(hipSetDevice(a->ctx->device));
  const float* p = a->points.as<float>();
  const int st = add_range_data_stage_b(a->ctx, p, p + a->cap, p + 2 * a->cap, a->count, voxel_filter_size, a->current_pose,
                                        returns_in_tracking, origin_in_tracking);
  a->count = 0;  // num_accumulated_ = 0; accumulated_range_data_ reset at the next first call (:449-452)
  a->num_accumulated = 0;
  return st;
}

extern "C" int dliom_deskew(dliom_ctx* ctx, const double prev_pose[7], const double predicted_pose[7],
                            double scan_period, const float* hits_xyzt, int64_t n, const float origin[3],
                            float min_range, float max_range, float* out_xyz, uint8_t* out_kind,
                            float current_pose[7]) {
  if (ctx == nullptr || prev_pose == nullptr || predicted_pose == nullptr || origin == nullptr || n < 0 ||
      current_pose == nullptr || (n > 0 && (hits_xyzt == nullptr || out_xyz == nullptr || out_kind == nullptr)) ||
      !(scan_period > 0.))
    return DLIOM_ERR_INVALID_ARGUMENT;
  if (n == 0) return DLIOM_ERR_EMPTY_CLOUD;  // CHECK(!synchronized_data.ranges.empty()) (:383)
  if (n >= (int64_t{1} << 27)) return DLIOM_ERR_INVALID_ARGUMENT;  // (hit indices share a word with the records' launch tag)
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  DeskewArgs a;
  DLIOM_TRY(make_deskew_args(prev_pose, predicted_pose, scan_period, origin, min_range, max_range, hits_xyzt[3], &a));
  deskew_test_hook(ctx, &a);
  const size_t in_bytes = static_cast<size_t>(n) * 16;
  const size_t xyz_off = (in_bytes + 255) & ~static_cast<size_t>(255);
  const size_t kind_off = xyz_off + ((static_cast<size_t>(n) * 12 + 255) & ~static_cast<size_t>(255));
  const size_t pose_off = kind_off + ((static_cast<size_t>(n) + 255) & ~static_cast<size_t>(255));
  DLIOM_TRY(ctx->misc.reserve(pose_off + 64));
  char* base = static_cast<char*>(ctx->misc.p);
  unsigned* d_flags = nullptr;
  DLIOM_TRY(deskew_flag_buffer(ctx, &d_flags));
  DLIOM_HIP_TRY(hipMemcpyAsync(base, hits_xyzt, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  const float* in = reinterpret_cast<const float*>(base);
  float* out = reinterpret_cast<float*>(base + xyz_off);
  const DeskewLaunch launch{in, in + 1, in + 2, in + 3, nullptr, 4, static_cast<int>(n), out, out + 1, out + 2, 3,
                            reinterpret_cast<unsigned char*>(base + kind_off)};
  hipLaunchKernelGGL(deskew_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, ctx->stream, a, launch.in_x,
                     launch.in_y, launch.in_z, launch.in_t, launch.in_origin, 4, launch.n, launch.out_x, launch.out_y, launch.out_z, 3,
                     launch.out_kind, d_flags, 0, static_cast<const unsigned*>(nullptr));
  DLIOM_HIP_TRY(hipGetLastError());
  std::vector<unsigned> host_flags(kDeskewFlagWords);
  DLIOM_HIP_TRY(hipMemcpyAsync(host_flags.data(), d_flags, 4 * kDeskewFlagWords, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  bool fixed = false;
  DLIOM_TRY(verify_deskew(ctx, a, launch, host_flags.data(), d_flags, &fixed));
  if (fixed) DLIOM_HIP_TRY(hipMemcpyAsync(host_flags.data(), d_flags, 28, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipMemcpyAsync(out_xyz, base + xyz_off, static_cast<size_t>(n) * 12, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipMemcpyAsync(out_kind, base + kind_off, static_cast<size_t>(n), hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  std::memcpy(current_pose, host_flags.data(), 28);
  return DLIOM_OK;
}

extern "C" int dliom_deskew_check_stats(const dliom_ctx* ctx, int64_t* records_checked, int64_t* ring_overflows, int64_t* hits_fixed) {
  if (ctx == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  if (records_checked != nullptr) *records_checked = ctx->deskew_records_checked;
  if (ring_overflows != nullptr) *ring_overflows = ctx->deskew_overflows;
  if (hits_fixed != nullptr) *hits_fixed = ctx->deskew_fixed_hits;
  return DLIOM_OK;
}
