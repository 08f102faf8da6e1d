// Per-scan pre-processing of LocalTrajectoryBuilder3D::AddRangeData
// (mapping/internal/3d/local_trajectory_builder_3d.cc:393-487): the per-hit de-skew -- one
// double-precision slerp + rigid composition per point, the part the reference pays ~1 us per
// point for -- runs on the device; the order-dependent "first point per voxel" filters around it
// stay on the host (front_end.hip) until the device voxel filter lands.
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
  float ox, oy, oz;             // sensor origin in the tracking frame
  float min_range, max_range;
  int use_stamps;               // 0: |t_0| < 1e-3, every hit takes the predicted pose
};

__device__ __forceinline__ void quat_mul_sse_d(const double* a, const double* b, double* r) {
  // Eigen Quaterniond product, SSE2 evaluation order (host_math.h::qmul_d)
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

// One hit: pose_i = (prev * [s t_rel, slerp(I, q_rel, s)]).cast<float>() (:437-445,869-877), then
// hit/origin into the local frame and the range gate (:454-472).
// out_kind: 0 dropped (range < min_range), 1 return, 2 miss (beyond max_range: cropped ray end).
__global__ void deskew_kernel(DeskewArgs a, const float4* __restrict__ hits, int n,
                              float* __restrict__ out_xyz, unsigned char* __restrict__ out_kind,
                              float* __restrict__ last_pose7) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 h = hits[i];
  Quat4 q;
  float tx, ty, tz;
  if (a.use_stamps) {
    const double s = (a.scan_period + static_cast<double>(h.w)) / a.scan_period;
    // Eigen::Quaterniond::Identity().slerp(s, q_rel)
    const double one = 1.0 - 2.220446049250313e-16;
    const double d = a.rel_q[0];
    const double abs_d = fabs(d);
    double scale0, scale1;
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
    double qq[4];
    quat_mul_sse_d(a.prev_q, qi, qq);
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
    tx = static_cast<float>(((ti[0] + u[0] * uvx) + cx) + a.prev_t[0]);
    ty = static_cast<float>(((ti[1] + u[0] * uvy) + cy) + a.prev_t[1]);
    tz = static_cast<float>(((ti[2] + u[0] * uvz) + cz) + a.prev_t[2]);
    q = Quat4{static_cast<float>(qq[0]), static_cast<float>(qq[1]), static_cast<float>(qq[2]),
              static_cast<float>(qq[3])};
  } else {
    q = Quat4{a.cur_q[0], a.cur_q[1], a.cur_q[2], a.cur_q[3]};
    tx = a.cur_t[0];
    ty = a.cur_t[1];
    tz = a.cur_t[2];
  }
  float hx, hy, hz, ox, oy, oz;
  rotate_point(q, h.x, h.y, h.z, hx, hy, hz);
  hx += tx;
  hy += ty;
  hz += tz;
  rotate_point(q, a.ox, a.oy, a.oz, ox, oy, oz);
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
  out_xyz[3 * i] = hx;
  out_xyz[3 * i + 1] = hy;
  out_xyz[3 * i + 2] = hz;
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

}  // namespace dliom

using namespace dliom;

extern "C" int dliom_deskew(dliom_ctx* ctx, const double prev_pose[7], const double predicted_pose[7],
                            double scan_period, const float* hits_xyzt, int64_t n, const float origin[3],
                            float min_range, float max_range, float* out_xyz, uint8_t* out_kind,
                            float current_pose[7]) {
  if (ctx == nullptr || prev_pose == nullptr || predicted_pose == nullptr || origin == nullptr || n < 0 ||
      current_pose == nullptr || (n > 0 && (hits_xyzt == nullptr || out_xyz == nullptr || out_kind == nullptr)) ||
      !(scan_period > 0.))
    return DLIOM_ERR_INVALID_ARGUMENT;
  if (n == 0) return DLIOM_ERR_EMPTY_CLOUD;  // CHECK(!synchronized_data.ranges.empty()) (:383)
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
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
  DeskewArgs a;
  std::memcpy(a.prev_t, prev.t, sizeof(a.prev_t));
  std::memcpy(a.prev_q, prev.q, sizeof(a.prev_q));
  std::memcpy(a.rel_t, rel.t, sizeof(a.rel_t));
  std::memcpy(a.rel_q, rel.q, sizeof(a.rel_q));
  float cf[7];
  pose_to_float7(cur, cf);
  std::memcpy(a.cur_t, cf, 12);
  std::memcpy(a.cur_q, cf + 3, 16);
  a.scan_period = scan_period;
  a.ox = origin[0];
  a.oy = origin[1];
  a.oz = origin[2];
  a.min_range = min_range;
  a.max_range = max_range;
  a.use_stamps = std::abs(hits_xyzt[3]) < 1e-3 ? 0 : 1;  // hits.front().point_time[3] (:429)
  const size_t in_bytes = static_cast<size_t>(n) * 16;
  const size_t xyz_off = (in_bytes + 255) & ~static_cast<size_t>(255);
  const size_t kind_off = xyz_off + ((static_cast<size_t>(n) * 12 + 255) & ~static_cast<size_t>(255));
  const size_t pose_off = kind_off + ((static_cast<size_t>(n) + 255) & ~static_cast<size_t>(255));
  DLIOM_TRY(ctx->misc.reserve(pose_off + 64));
  char* base = static_cast<char*>(ctx->misc.p);
  DLIOM_HIP_TRY(hipMemcpyAsync(base, hits_xyzt, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(deskew_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, ctx->stream, a,
                     reinterpret_cast<const float4*>(base), static_cast<int>(n),
                     reinterpret_cast<float*>(base + xyz_off), reinterpret_cast<unsigned char*>(base + kind_off),
                     reinterpret_cast<float*>(base + pose_off));
  DLIOM_HIP_TRY(hipGetLastError());
  DLIOM_HIP_TRY(hipMemcpyAsync(out_xyz, base + xyz_off, static_cast<size_t>(n) * 12, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipMemcpyAsync(out_kind, base + kind_off, static_cast<size_t>(n), hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipMemcpyAsync(current_pose, base + pose_off, 28, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return DLIOM_OK;
}
