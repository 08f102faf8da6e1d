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
// Inputs / outputs are strided so that packed host layouts (xyzt stride 4, xyz stride 3) and the
// device SoA layout (stride 1) run the same code.
__global__ void deskew_kernel(DeskewArgs a, const float* __restrict__ in_x, const float* __restrict__ in_y,
                              const float* __restrict__ in_z, const float* __restrict__ in_t,
                              const float* __restrict__ in_origin, int in_stride, int n,
                              float* __restrict__ out_x, float* __restrict__ out_y, float* __restrict__ out_z,
                              int out_stride, unsigned char* __restrict__ out_kind,
                              float* __restrict__ last_pose7) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t ii = static_cast<size_t>(i) * in_stride;
  const float4 h = make_float4(in_x[ii], in_y[ii], in_z[ii], in_t[ii]);
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
__global__ __launch_bounds__(256) void transform_kernel(Quat4 q, float tx, float ty, float tz, const float* __restrict__ x,
                                                        const float* __restrict__ y, const float* __restrict__ z, int n,
                                                        float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz,
                                                        float* __restrict__ partial_max) {
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
    m[0] = fmaxf(m[0], rx * rx + (ry * ry + rz * rz));
    m[1] = fmaxf(m[1], fabsf(rx));
    m[2] = fmaxf(m[2], fabsf(ry));
    m[3] = fmaxf(m[3], fabsf(rz));
  }
  __shared__ float part[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m[k] = fmaxf(m[k], __shfl_xor(m[k], off, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][k] = m[k];
  }
  __syncthreads();
  if (threadIdx.x < 4)
    partial_max[4 * blockIdx.x + threadIdx.x] =
        fmaxf(fmaxf(part[0][threadIdx.x], part[1][threadIdx.x]), fmaxf(part[2][threadIdx.x], part[3][threadIdx.x]));
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
  float* d_pose = reinterpret_cast<float*>(base + off_pose);
  const int threads = 256;
  DLIOM_HIP_TRY(hipMemcpyAsync(base + off_raw, ranges_xyzt, 16 * nn, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(split_xyzt_kernel, dim3(static_cast<unsigned>((n + threads - 1) / threads)), dim3(threads), 0,
                     ctx->stream, reinterpret_cast<const float4*>(base + off_raw), static_cast<int>(n), b, b + nn,
                     b + 2 * nn, b + 3 * nn);
  // hits = VoxelFilter(0.5f * voxel_filter_size).Filter(ranges): the time rides along
  int64_t n1 = 0;
  DLIOM_TRY(voxel_filter_arrays(ctx, Soa{b, b + nn, b + 2 * nn, b + 3 * nn, n}, 0.5f * voxel_filter_size, c, c + nn,
                                c + 2 * nn, c + 3 * nn, &n1));
  const bool multi = origin_index != nullptr && num_origins > 1;
  if (multi) {  // the same filter once more with the origin index as the passenger: same survivors, same order
    DLIOM_HIP_TRY(hipMemcpyAsync(oi_in, origin_index, 4 * nn, hipMemcpyHostToDevice, ctx->stream));
    int64_t n1b = 0;
    DLIOM_TRY(voxel_filter_arrays(ctx, Soa{b, b + nn, b + 2 * nn, oi_in, n}, 0.5f * voxel_filter_size, d, d + nn,
                                  d + 2 * nn, oi_f, &n1b));
    if (n1b != n1) return DLIOM_ERR_INVALID_ARGUMENT;
  }
  DeskewArgs a;
  // the first range always survives the filter, so hits.front() is ranges.front()
  DLIOM_TRY(make_deskew_args(prev_pose, predicted_pose, scan_period, origins, min_range, max_range, ranges_xyzt[3], &a));
  for (int k = 0; k < std::min(num_origins, 4); ++k)
    for (int i = 0; i < 3; ++i) a.origins[k][i] = origins[3 * k + i];
  hipLaunchKernelGGL(deskew_kernel, dim3(static_cast<unsigned>((n1 + threads - 1) / threads)), dim3(threads), 0,
                     ctx->stream, a, c, c + nn, c + 2 * nn, c + 3 * nn, multi ? oi_f : static_cast<const float*>(nullptr),
                     1, static_cast<int>(n1), d, d + nn, d + 2 * nn, 1, kind, d_pose);
  DLIOM_HIP_TRY(hipGetLastError());
  // returns (kind 1), in hit order; misses (kind 2) are not used by the 3D path.  The compaction's read-back brings the
  // last hit's pose along (one round trip, no memcpy)
  DLIOM_TRY(compact_equal_arrays(ctx, Soa{d, d + nn, d + 2 * nn, nullptr, n1}, kind, 1, e, e + nn, e + 2 * nn, num_returns, d_pose, 7,
                                 current_pose));
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
  DLIOM_TRY(ctx->rescore.reserve(3 * al(4 * nn) + 16 * kTransformBlocks));  // filtered returns | per-workgroup maxima (misc holds the inputs)
  float* f = ctx->rescore.as<float>();
  float* d_partial_max = reinterpret_cast<float*>(static_cast<char*>(ctx->rescore.p) + 3 * al(4 * nn));
  const size_t fs = al(4 * nn) / 4;
  int64_t n3 = 0;
  DLIOM_TRY(voxel_filter_arrays(ctx, Soa{rx, ry, rz, nullptr, n2}, voxel_filter_size, f, f + fs, f + 2 * fs, nullptr, &n3));
  // current_pose.inverse() in float (rigid_transform.h:167-171)
  const QF qc{current_pose[3], -current_pose[4], -current_pose[5], -current_pose[6]};
  const F3 rt = qrot(qc, F3{current_pose[0], current_pose[1], current_pose[2]});
  const F3 ti{-rt.x, -rt.y, -rt.z};
  const F3 o = add3(qrot(qc, F3{current_pose[0], current_pose[1], current_pose[2]}), ti);  // inverse * origin
  origin_in_tracking[0] = o.x;
  origin_in_tracking[1] = o.y;
  origin_in_tracking[2] = o.z;
  float *ox, *oy, *oz;
  DLIOM_TRY(alloc_device_cloud(ctx, n3, returns_in_tracking, &ox, &oy, &oz));
  float max_norm = 0.f, abs_max[3] = {0.f, 0.f, 0.f};
  int st = DLIOM_OK;
  const int threads = 256;
  if (n3 > 0) {
    const unsigned blocks = static_cast<unsigned>(std::min<int64_t>((n3 + threads - 1) / threads, kTransformBlocks));
    hipLaunchKernelGGL(transform_kernel, dim3(blocks), dim3(threads), 0, ctx->stream, Quat4{qc.w, qc.x, qc.y, qc.z}, ti.x, ti.y,
                       ti.z, f, f + fs, f + 2 * fs, static_cast<int>(n3), ox, oy, oz, d_partial_max);
    float* host = static_cast<float*>(ctx->pinned);
    const GatherJob job{d_partial_max, 4 * blocks};
    st = gather_and_wait(ctx, &job, 1, host);
    if (st == DLIOM_OK) {
      float sq = 0.f;
      for (unsigned b = 0; b < blocks; ++b) {
        sq = std::max(sq, host[4 * b]);
        for (int a = 0; a < 3; ++a) abs_max[a] = std::max(abs_max[a], host[4 * b + 1 + a]);
      }
      max_norm = std::sqrt(sq);  // sqrt is monotone and correctly rounded: == the maximum of the norms
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
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  DeskewArgs a;
  DLIOM_TRY(make_deskew_args(prev_pose, predicted_pose, scan_period, origin, min_range, max_range, hits_xyzt[3], &a));
  const size_t in_bytes = static_cast<size_t>(n) * 16;
  const size_t xyz_off = (in_bytes + 255) & ~static_cast<size_t>(255);
  const size_t kind_off = xyz_off + ((static_cast<size_t>(n) * 12 + 255) & ~static_cast<size_t>(255));
  const size_t pose_off = kind_off + ((static_cast<size_t>(n) + 255) & ~static_cast<size_t>(255));
  DLIOM_TRY(ctx->misc.reserve(pose_off + 64));
  char* base = static_cast<char*>(ctx->misc.p);
  DLIOM_HIP_TRY(hipMemcpyAsync(base, hits_xyzt, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(deskew_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, ctx->stream, a,
                     reinterpret_cast<const float*>(base), reinterpret_cast<const float*>(base) + 1,
                     reinterpret_cast<const float*>(base) + 2, reinterpret_cast<const float*>(base) + 3,
                     static_cast<const float*>(nullptr), 4, static_cast<int>(n), reinterpret_cast<float*>(base + xyz_off),
                     reinterpret_cast<float*>(base + xyz_off) + 1, reinterpret_cast<float*>(base + xyz_off) + 2, 3,
                     reinterpret_cast<unsigned char*>(base + kind_off), reinterpret_cast<float*>(base + pose_off));
  DLIOM_HIP_TRY(hipGetLastError());
  DLIOM_HIP_TRY(hipMemcpyAsync(out_xyz, base + xyz_off, static_cast<size_t>(n) * 12, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipMemcpyAsync(out_kind, base + kind_off, static_cast<size_t>(n), hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipMemcpyAsync(current_pose, base + pose_off, 28, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return DLIOM_OK;
}
