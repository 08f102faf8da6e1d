// ORACLE (test infrastructure, NOT product code).
//
// CPU restatement of the scan-to-submap orchestration around the matchers:
//   mapping/internal/motion_filter.cc:40-58                     MotionFilter::IsSimilar
//   mapping/3d/submap_3d.cc:196-204,264-326                      Submap3D, ActiveSubmaps3D
//   mapping/internal/3d/local_trajectory_builder_3d.cc:493-572   AddAccumulatedRangeData
//   mapping/internal/3d/local_trajectory_builder_3d.cc:584-622   InsertIntoSubmap
// with the GTSAM window (WindowOptimize, :555-557 -- external dependency, parity unpinned)
// left to the caller: Match() stops at `pose_estimate`, Insert() takes the optimised pose.
// RotationalScanMatcher::ComputeHistogram (:605-610) is not part of this path's outputs.
#ifndef ORACLE_OM_FRONT_END_H_
#define ORACLE_OM_FRONT_END_H_

#include <cstdint>
#include <memory>
#include <vector>

#include "om_csm3d.h"
#include "om_rtcsm3d.h"

namespace oracle {

struct MotionFilterOptions {
  double max_time_seconds, max_distance_meters, max_angle_radians;
};

class MotionFilter {
 public:
  explicit MotionFilter(const MotionFilterOptions& o) : options_(o) {}
  // time in common::Time ticks (100 ns); FromSeconds truncates (common/time.cc:24-27).
  bool IsSimilar(int64_t time, const Rigid3d& pose) {
    ++num_total_;
    if (num_total_ > 1 &&
        time - last_time_ <= static_cast<int64_t>(options_.max_time_seconds * 1e7) &&
        (pose.translation - last_pose_.translation).norm() <= options_.max_distance_meters &&
        GetAngle(pose.inverse() * last_pose_) <= options_.max_angle_radians) {
      return true;
    }
    last_time_ = time;
    last_pose_ = pose;
    return false;
  }

 private:
  const MotionFilterOptions options_;
  int64_t num_total_ = 0;
  int64_t last_time_ = 0;
  Rigid3d last_pose_;
};

struct SubmapsOptions3D {
  double high_resolution, high_resolution_max_range, low_resolution;
  int num_range_data;
  double hit_probability, miss_probability;
  int num_free_space_voxels;
};

class Submap3D {
 public:
  Submap3D(float high_resolution, float low_resolution, const Rigid3d& local_pose)
      : local_pose_(local_pose), hi_(new HybridGrid(high_resolution)), lo_(new HybridGrid(low_resolution)) {}
  const Rigid3d& local_pose() const { return local_pose_; }
  HybridGrid& high_resolution_hybrid_grid() const { return *hi_; }
  HybridGrid& low_resolution_hybrid_grid() const { return *lo_; }
  int num_range_data() const { return num_range_data_; }
  bool finished() const { return finished_; }
  void Finish() { finished_ = true; }
  // submap_3d.cc:264-279 (note: max range arrives as an int)
  void InsertRangeData(const RangeData& range_data, const RangeDataInserter3D& inserter,
                       const int high_resolution_max_range) {
    const RangeData transformed = TransformRangeData(range_data, local_pose_.inverse().cast<float>());
    inserter.Insert(FilterRangeDataByMaxRange(transformed, static_cast<float>(high_resolution_max_range)),
                    hi_.get());
    inserter.Insert(transformed, lo_.get());
    ++num_range_data_;
  }

 private:
  Rigid3d local_pose_;
  std::unique_ptr<HybridGrid> hi_, lo_;
  int num_range_data_ = 0;
  bool finished_ = false;
};

class ActiveSubmaps3D {
 public:
  explicit ActiveSubmaps3D(const SubmapsOptions3D& o)
      : options_(o),
        inserter_(static_cast<float>(o.hit_probability), static_cast<float>(o.miss_probability),
                  o.num_free_space_voxels) {
    AddSubmap(Rigid3d());
  }
  int matching_index() const { return matching_submap_index_; }
  const std::vector<std::shared_ptr<Submap3D>>& submaps() const { return submaps_; }
  // submap_3d.cc:303-314
  bool InsertRangeData(const RangeData& range_data, const Quatd& gravity_alignment) {
    for (auto& submap : submaps_)
      submap->InsertRangeData(range_data, inserter_, static_cast<int>(options_.high_resolution_max_range));
    if (submaps_.back()->num_range_data() == options_.num_range_data) {
      AddSubmap(Rigid3d(range_data.origin.cast<double>(), gravity_alignment));
      return true;
    }
    return false;
  }

 private:
  void AddSubmap(const Rigid3d& local_pose) {  // submap_3d.cc:316-326
    if (submaps_.size() > 1) {
      submaps_.front()->Finish();
      ++matching_submap_index_;
      finished_.push_back(submaps_.front());
      submaps_.erase(submaps_.begin());
    }
    submaps_.emplace_back(new Submap3D(static_cast<float>(options_.high_resolution),
                                       static_cast<float>(options_.low_resolution), local_pose));
  }
  const SubmapsOptions3D options_;
  int matching_submap_index_ = 0;
  std::vector<std::shared_ptr<Submap3D>> submaps_;
  std::vector<std::shared_ptr<Submap3D>> finished_;
  RangeDataInserter3D inserter_;
};

struct FrontEndOptions {
  AdaptiveVoxelFilterOptions high_resolution_adaptive_voxel_filter;
  AdaptiveVoxelFilterOptions low_resolution_adaptive_voxel_filter;
  bool use_online_correlative_scan_matching;
  RealTimeCorrelativeScanMatcherOptions real_time_correlative_scan_matcher;
  CeresScanMatcherOptions3D ceres_scan_matcher;
  MotionFilterOptions motion_filter;
  SubmapsOptions3D submaps;
};

struct MatchResult {
  bool dropped = false;
  Rigid3d pose_estimate, pose_observation_in_submap, initial_ceres_pose;
  float rtcsm_score = 0.f;
  ceres_like::Summary summary;
  int64_t num_high = 0, num_low = 0;
};

class FrontEnd {
 public:
  explicit FrontEnd(const FrontEndOptions& o)
      : options_(o),
        active_submaps_(o.submaps),
        motion_filter_(o.motion_filter),
        rtcsm_(o.real_time_correlative_scan_matcher),
        csm_(o.ceres_scan_matcher) {}

  ActiveSubmaps3D& active_submaps() { return active_submaps_; }

  // local_trajectory_builder_3d.cc:493-553
  MatchResult Match(const Rigid3d& pose_prediction, const RangeData& filtered_range_data_in_tracking) {
    MatchResult r;
    range_data_ = filtered_range_data_in_tracking;
    if (filtered_range_data_in_tracking.returns.empty()) {
      r.dropped = true;
      return r;
    }
    std::shared_ptr<const Submap3D> matching_submap = active_submaps_.submaps().front();
    Rigid3d initial_ceres_pose = matching_submap->local_pose().inverse() * pose_prediction;
    const PointCloud hi = AdaptiveVoxelFilter(options_.high_resolution_adaptive_voxel_filter,
                                              filtered_range_data_in_tracking.returns);
    if (hi.empty()) {
      r.dropped = true;
      return r;
    }
    if (options_.use_online_correlative_scan_matching) {
      const Rigid3d initial_pose = initial_ceres_pose;
      r.rtcsm_score = rtcsm_threads > 1
                          ? rtcsm_.MatchThreaded(initial_pose, hi, matching_submap->high_resolution_hybrid_grid(),
                                                 &initial_ceres_pose, rtcsm_threads)
                          : rtcsm_.Match(initial_pose, hi, matching_submap->high_resolution_hybrid_grid(), &initial_ceres_pose);
    }
    const PointCloud lo = AdaptiveVoxelFilter(options_.low_resolution_adaptive_voxel_filter,
                                              filtered_range_data_in_tracking.returns);
    if (lo.empty()) {
      r.dropped = true;
      return r;
    }
    r.num_high = static_cast<int64_t>(hi.size());
    r.num_low = static_cast<int64_t>(lo.size());
    csm_.Match((matching_submap->local_pose().inverse() * pose_prediction).translation, initial_ceres_pose,
               {{&hi, &matching_submap->high_resolution_hybrid_grid()},
                {&lo, &matching_submap->low_resolution_hybrid_grid()}},
               &r.pose_observation_in_submap, &r.summary);
    r.initial_ceres_pose = initial_ceres_pose;
    r.pose_estimate = matching_submap->local_pose() * r.pose_observation_in_submap;
    return r;
  }

  // local_trajectory_builder_3d.cc:560-565,584-604.  Returns 0 not inserted, 1 inserted,
  // 2 inserted and a new submap was added.
  int Insert(int64_t time, const Rigid3d& pose_estimate, const Quatd& gravity_alignment) {
    if (motion_filter_.IsSimilar(time, pose_estimate)) return 0;
    const RangeData in_local = TransformRangeData(range_data_, pose_estimate.cast<float>());
    return active_submaps_.InsertRangeData(in_local, gravity_alignment) ? 2 : 1;
  }

 private:
  const FrontEndOptions options_;
  ActiveSubmaps3D active_submaps_;
  MotionFilter motion_filter_;
  RealTimeCorrelativeScanMatcher3D rtcsm_;
 public:
  int rtcsm_threads = 1;  // > 1: the candidate loop on this many threads (a CPU baseline variant)
 private:
  CeresScanMatcher3D csm_;
  RangeData range_data_;
};

}  // namespace oracle

#endif  // ORACLE_OM_FRONT_END_H_
