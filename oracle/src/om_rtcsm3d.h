// ORACLE (test infrastructure, NOT product code).
//
// Real-time correlative scan matcher (exhaustive voxel-accurate search) over a
// HybridGrid, restating
//   mapping/internal/3d/scan_matching/real_time_correlative_scan_matcher_3d.cc
//     :34-53   Match            (first strictly greater score wins)
//     :55-95   GenerateExhaustiveSearchTransforms (z,y,x,rz,ry,rx nesting)
//     :97-113  ScoreCandidate   (sequential float sum, double penalty)
// with the reference's mixed float/double arithmetic kept operation by
// operation.
#ifndef ORACLE_OM_RTCSM3D_H_
#define ORACLE_OM_RTCSM3D_H_

#include <cmath>
#include <cstdlib>
#include <thread>
#include <vector>

#include "om_sensor.h"

namespace oracle {

struct RealTimeCorrelativeScanMatcherOptions {
  double linear_search_window;
  double angular_search_window;
  double translation_delta_cost_weight;
  double rotation_delta_cost_weight;
};

struct SearchWindow3D {
  int linear_window_size;
  int angular_window_size;
  float angular_step_size;
  float max_scan_range;
};

class RealTimeCorrelativeScanMatcher3D {
 public:
  explicit RealTimeCorrelativeScanMatcher3D(
      const RealTimeCorrelativeScanMatcherOptions& options)
      : options_(options) {}

  // rtcsm_3d.cc:58-70
  SearchWindow3D ComputeWindow(float resolution, const PointCloud& cloud) const {
    SearchWindow3D w;
    w.linear_window_size = RoundToInt(options_.linear_search_window / resolution);
    float max_scan_range = 3.f * resolution;
    for (const Vec3f& p : cloud) {
      const float range = p.norm();
      max_scan_range = std::max(range, max_scan_range);
    }
    const float kSafetyMargin = 1.f - 1e-3f;
    w.angular_step_size =
        kSafetyMargin * std::acos(1.f - Pow2(resolution) /
                                            (2.f * Pow2(max_scan_range)));
    w.angular_window_size =
        RoundToInt(options_.angular_search_window / w.angular_step_size);
    w.max_scan_range = max_scan_range;
    return w;
  }

  // rtcsm_3d.cc:71-94
  std::vector<Rigid3f> GenerateExhaustiveSearchTransforms(
      float resolution, const PointCloud& cloud) const {
    const SearchWindow3D w = ComputeWindow(resolution, cloud);
    const int L = w.linear_window_size, A = w.angular_window_size;
    std::vector<Rigid3f> result;
    for (int z = -L; z <= L; ++z)
      for (int y = -L; y <= L; ++y)
        for (int x = -L; x <= L; ++x)
          for (int rz = -A; rz <= A; ++rz)
            for (int ry = -A; ry <= A; ++ry)
              for (int rx = -A; rx <= A; ++rx) {
                const Vec3f angle_axis(rx * w.angular_step_size,
                                       ry * w.angular_step_size,
                                       rz * w.angular_step_size);
                result.emplace_back(
                    Vec3f(x * resolution, y * resolution, z * resolution),
                    AngleAxisVectorToRotationQuaternion(angle_axis));
              }
    return result;
  }

  // rtcsm_3d.cc:97-113.  Returns <= 0 only where the reference would CHECK-fail.
  float ScoreCandidate(const HybridGrid& grid, const PointCloud& transformed,
                       const Rigid3f& transform) const {
    float score = 0.f;
    for (const Vec3f& p : transformed) {
      score += grid.GetProbability(grid.GetCellIndex(p));
    }
    score /= static_cast<float>(transformed.size());
    const float angle = GetAngle(transform);
    score *= std::exp(-Pow2(transform.translation.norm() *
                                options_.translation_delta_cost_weight +
                            angle * options_.rotation_delta_cost_weight));
    return score;
  }

  // rtcsm_3d.cc:34-53.  Optionally reports every candidate's score (in
  // generation order) and the index of the winner.
  float Match(const Rigid3d& initial_pose_estimate, const PointCloud& cloud,
              const HybridGrid& grid, Rigid3d* pose_estimate,
              std::vector<float>* all_scores = nullptr,
              int* best_index = nullptr) const {
    float best_score = -1.f;
    int index = 0;
    for (const Rigid3f& transform :
         GenerateExhaustiveSearchTransforms(grid.resolution(), cloud)) {
      const Rigid3f candidate = initial_pose_estimate.cast<float>() * transform;
      const float score =
          ScoreCandidate(grid, TransformPointCloud(cloud, candidate), transform);
      if (!(score > 0.f)) std::abort();  // CHECK_GT(score, 0.f)
      if (all_scores != nullptr) all_scores->push_back(score);
      if (score > best_score) {
        best_score = score;
        *pose_estimate = candidate.cast<double>();
        if (best_index != nullptr) *best_index = index;
      }
      ++index;
    }
    return best_score;
  }

  // BASELINE.md section 2, "8 threads": the same loop with the candidates cut into contiguous ranges, one per thread; a
  // range keeps its first strictly greater score and the ranges are combined in generation order with the same strict
  // `>`, so the winner is the serial loop's.  Not what the reference does (its loop is serial): a CPU baseline variant.
  float MatchThreaded(const Rigid3d& initial_pose_estimate, const PointCloud& cloud, const HybridGrid& grid,
                      Rigid3d* pose_estimate, int num_threads) const {
    const std::vector<Rigid3f> ts = GenerateExhaustiveSearchTransforms(grid.resolution(), cloud);
    const Rigid3f init = initial_pose_estimate.cast<float>();
    const int n = static_cast<int>(ts.size());
    const int parts = std::max(1, std::min(num_threads, n));
    std::vector<float> best(static_cast<size_t>(parts), -1.f);
    std::vector<int> best_c(static_cast<size_t>(parts), -1);
    std::vector<std::thread> pool;
    for (int t = 0; t < parts; ++t)
      pool.emplace_back([&, t] {
        const int lo = static_cast<int>(static_cast<long long>(n) * t / parts);
        const int hi = static_cast<int>(static_cast<long long>(n) * (t + 1) / parts);
        for (int c = lo; c < hi; ++c) {
          const Rigid3f candidate = init * ts[c];
          const float score = ScoreCandidate(grid, TransformPointCloud(cloud, candidate), ts[c]);
          if (score > best[t]) {
            best[t] = score;
            best_c[t] = c;
          }
        }
      });
    for (std::thread& th : pool) th.join();
    float s = -1.f;
    int at = -1;
    for (int t = 0; t < parts; ++t)
      if (best[t] > s) {
        s = best[t];
        at = best_c[t];
      }
    if (at >= 0) *pose_estimate = (init * ts[at]).cast<double>();
    return s;
  }

 private:
  const RealTimeCorrelativeScanMatcherOptions options_;
};

}  // namespace oracle

#endif  // ORACLE_OM_RTCSM3D_H_
