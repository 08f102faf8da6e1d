// ORACLE (test infrastructure, NOT product code).
//
// Per-scan pre-processing of LocalTrajectoryBuilder3D::AddRangeData restated:
//   mapping/internal/3d/local_trajectory_builder_3d.cc:393-395   VoxelFilter(0.5 * voxel_filter_size) on the hits
//   :421-446  per-hit pose: s = (T + t_i)/T, InterpolatePose(s, rel_trans) (:869-877), prev * tmp, cast<float>
//   :454-472  transform hit and origin, range gate [min_range, max_range] (beyond: cropped miss)
//   :476-487  VoxelFilter(voxel_filter_size) on returns / misses, back to the tracking frame by
//             current_pose.inverse()
// Eigen::Quaterniond::slerp (Eigen 3.3 Quaternion.h) is restated in Slerp() below.
#ifndef ORACLE_OM_DESKEW_H_
#define ORACLE_OM_DESKEW_H_

#include <cmath>
#include <limits>
#include <vector>

#include "om_sensor.h"

namespace oracle {

struct TimedPoint {
  float x, y, z, t;  // t <= 0: seconds relative to the end of the scan
};

// Quaterniond::Identity().slerp(t, other)
inline Quatd SlerpFromIdentity(double t, const Quatd& other) {
  const double one = 1.0 - std::numeric_limits<double>::epsilon();
  const double d = other.w;  // dot((0,0,0,1), other) in packet order is exactly w
  const double abs_d = std::fabs(d);
  double scale0, scale1;
  if (abs_d >= one) {
    scale0 = 1.0 - t;
    scale1 = t;
  } else {
    const double theta = std::acos(abs_d);
    const double sin_theta = std::sin(theta);
    scale0 = std::sin((1.0 - t) * theta) / sin_theta;
    scale1 = std::sin(t * theta) / sin_theta;
  }
  if (d < 0.0) scale1 = -scale1;
  return Quatd(scale0 * 1.0 + scale1 * other.w, scale0 * 0.0 + scale1 * other.x,
               scale0 * 0.0 + scale1 * other.y, scale0 * 0.0 + scale1 * other.z);
}

struct DeskewOptions {
  double scan_period;  // scan_period_ (double member initialised from the float option)
  float min_range, max_range;
  float voxel_filter_size;
};

struct DeskewResult {
  Rigid3f current_pose;                       // hits_poses.back()
  std::vector<Vec3f> hits_in_local;           // per input hit (after the 0.5*vfs filter), pre-gate
  std::vector<int> kind;                      // 0 dropped (< min_range), 1 return, 2 miss
  RangeData filtered_in_tracking;             // what AddAccumulatedRangeData receives
};

// hits: the synchronized ranges (single lidar: every origin_index is 0).
inline DeskewResult DeskewAndFilter(const DeskewOptions& o, const Rigid3d& prev_pose, const Rigid3d& cur_pose,
                                    const std::vector<TimedPoint>& ranges, const Vec3f& origin) {
  DeskewResult r;
  // :393-395 VoxelFilter(0.5f * voxel_filter_size) keyed on the xyz part, order preserved
  std::vector<TimedPoint> hits;
  {
    VoxelFilter f(0.5f * o.voxel_filter_size);
    PointCloud xyz;
    xyz.reserve(ranges.size());
    for (const TimedPoint& p : ranges) xyz.emplace_back(p.x, p.y, p.z);
    for (int i : f.FilterIndices(xyz)) hits.push_back(ranges[i]);
  }
  std::vector<Rigid3f> poses;
  poses.reserve(hits.size());
  const Rigid3d rel_trans = prev_pose.inverse() * cur_pose;
  if (std::abs(hits.front().t) < 1e-3) {  // no per-point stamps: "Not discrewing!"
    poses.assign(hits.size(), cur_pose.cast<float>());
  } else {
    for (const TimedPoint& h : hits) {
      const double s = (o.scan_period + h.t) / o.scan_period;
      const Rigid3d tmp(s * rel_trans.translation, SlerpFromIdentity(s, rel_trans.rotation));
      poses.push_back((prev_pose * tmp).cast<float>());
    }
  }
  RangeData accumulated{Vec3f(), {}, {}};
  for (size_t i = 0; i < hits.size(); ++i) {
    const Vec3f hit_in_local = poses[i] * Vec3f(hits[i].x, hits[i].y, hits[i].z);
    const Vec3f origin_in_local = poses[i] * origin;
    const Vec3f delta = hit_in_local - origin_in_local;
    const float range = delta.norm();
    r.hits_in_local.push_back(hit_in_local);
    int kind = 0;
    if (range >= o.min_range) {
      if (range <= o.max_range) {
        accumulated.returns.push_back(hit_in_local);
        kind = 1;
      } else {
        accumulated.misses.push_back(origin_in_local + (o.max_range / range) * delta);
        kind = 2;
      }
    }
    r.kind.push_back(kind);
  }
  r.current_pose = poses.back();
  const RangeData filtered{r.current_pose.translation, VoxelFilter(o.voxel_filter_size).Filter(accumulated.returns),
                           VoxelFilter(o.voxel_filter_size).Filter(accumulated.misses)};
  r.filtered_in_tracking = TransformRangeData(filtered, r.current_pose.inverse());
  return r;
}

// The same per-scan steps with the synchronizer's origin table (every range carries an origin index,
// internal/3d/range_data_synchronizer.cc:92-113) and num_accumulated_range_data > 1 (:449-476): Add() is one
// AddRangeData call up to `++num_accumulated_`, Finish() the block guarded by
// `num_accumulated_ >= options_.num_accumulated_range_data()`.
class RangeDataAccumulator {
 public:
  Rigid3f Add(const DeskewOptions& o, const Rigid3d& prev_pose, const Rigid3d& cur_pose,
              const std::vector<TimedPoint>& ranges, const std::vector<int>& origin_index,
              const std::vector<Vec3f>& origins) {
    std::vector<int> keep;
    {
      VoxelFilter f(0.5f * o.voxel_filter_size);
      PointCloud xyz;
      xyz.reserve(ranges.size());
      for (const TimedPoint& p : ranges) xyz.emplace_back(p.x, p.y, p.z);
      keep = f.FilterIndices(xyz);
    }
    const Rigid3d rel_trans = prev_pose.inverse() * cur_pose;
    const bool stamps = !(std::abs(ranges[keep.front()].t) < 1e-3);
    if (num_ == 0) accumulated_ = RangeData{Vec3f(), {}, {}};
    Rigid3f pose = cur_pose.cast<float>();
    for (int i : keep) {
      const TimedPoint& h = ranges[i];
      if (stamps) {
        const double s = (o.scan_period + h.t) / o.scan_period;
        const Rigid3d tmp(s * rel_trans.translation, SlerpFromIdentity(s, rel_trans.rotation));
        pose = (prev_pose * tmp).cast<float>();
      }
      const Vec3f hit_in_local = pose * Vec3f(h.x, h.y, h.z);
      const Vec3f origin_in_local = pose * origins.at(origin_index.empty() ? 0 : origin_index[i]);
      const Vec3f delta = hit_in_local - origin_in_local;
      const float range = delta.norm();
      if (range >= o.min_range) {
        if (range <= o.max_range) {
          accumulated_.returns.push_back(hit_in_local);
        } else {
          accumulated_.misses.push_back(origin_in_local + (o.max_range / range) * delta);
        }
      }
    }
    current_pose_ = pose;  // hits_poses.back()
    ++num_;
    return current_pose_;
  }
  RangeData Finish(const DeskewOptions& o) {
    num_ = 0;
    const RangeData filtered{current_pose_.translation, VoxelFilter(o.voxel_filter_size).Filter(accumulated_.returns),
                             VoxelFilter(o.voxel_filter_size).Filter(accumulated_.misses)};
    return TransformRangeData(filtered, current_pose_.inverse());
  }
  int num_accumulated() const { return num_; }

 private:
  RangeData accumulated_{Vec3f(), {}, {}};
  Rigid3f current_pose_;
  int num_ = 0;
};

}  // namespace oracle

#endif  // ORACLE_OM_DESKEW_H_
