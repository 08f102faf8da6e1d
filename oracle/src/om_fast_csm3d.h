// ORACLE (test infrastructure, NOT product code).
//
// Branch-and-bound loop-closure matcher, restating
//   mapping/internal/3d/scan_matching/precomputation_grid_3d.{h,cc}      (uint8 max-pool pyramid)
//   mapping/internal/3d/scan_matching/fast_correlative_scan_matcher_3d.cc (:57-77 grid stack,
//       :147-262 Match / MatchWith3DofInitial / MatchFullSubmap, :264-304 DiscretizeScan,
//       :306-356 GenerateDiscreteScans, :358-392 lowest-resolution candidates,
//       :394-417 ScoreCandidates, :431-437 GetPoseFromCandidate, :439-492 BranchAndBound)
//   mapping/internal/3d/scan_matching/rotational_scan_matcher.cc          (histograms)
//   mapping/internal/3d/scan_matching/low_resolution_matcher.cc
// The order in which equal-score candidates are visited is whatever std::sort does with
// std::greater<Candidate3D> on the reference's input order; this file calls std::sort the same way.
// Eigen's vectorised reductions over dynamic float vectors (VectorXf::norm / dot, SSE2 packets
// of 4, two accumulators, predux (a0+a2)+(a1+a3), scalar tail) are restated in DynRedux().
#ifndef ORACLE_OM_FAST_CSM3D_H_
#define ORACLE_OM_FAST_CSM3D_H_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <unordered_map>
#include <vector>

#include "om_hybrid_grid.h"
#include "om_sensor.h"

namespace oracle {

using uint8 = std::uint8_t;

// HybridGridBase<uint8>: only value() == 0 for anything never written matters here, so the
// container is a hash map (the reference's tree layout is not observable through this path).
class PrecomputationGrid3D {
 public:
  explicit PrecomputationGrid3D(float resolution) : resolution_(resolution) {}
  float resolution() const { return resolution_; }
  Vec3i GetCellIndex(const Vec3f& point) const {  // hybrid_grid.h:430-435
    return Vec3i(RoundToInt(point.x / resolution_), RoundToInt(point.y / resolution_),
                 RoundToInt(point.z / resolution_));
  }
  uint8 value(const Vec3i& index) const {
    const auto it = cells_.find(Key(index));
    return it == cells_.end() ? 0 : it->second;
  }
  uint8* mutable_value(const Vec3i& index) { return &cells_[Key(index)]; }
  // precomputation_grid_3d.h:31-34
  static float ToProbability(float value) {
    return kMinProbability + value * ((kMaxProbability - kMinProbability) / 255.f);
  }
  template <typename F>
  void ForEachCell(F&& f) const {  // cells not equal to the default value (hybrid_grid.h:97-99)
    for (const auto& kv : cells_) {
      if (kv.second == 0) continue;
      f(Unkey(kv.first), kv.second);
    }
  }
  size_t num_cells() const { return cells_.size(); }

 private:
  static uint64_t Key(const Vec3i& i) {
    return (static_cast<uint64_t>(static_cast<uint32_t>(i.x + (1 << 20))) << 42) |
           (static_cast<uint64_t>(static_cast<uint32_t>(i.y + (1 << 20))) << 21) |
           static_cast<uint64_t>(static_cast<uint32_t>(i.z + (1 << 20)));
  }
  static Vec3i Unkey(uint64_t k) {
    return Vec3i(static_cast<int>((k >> 42) & 0x1FFFFF) - (1 << 20), static_cast<int>((k >> 21) & 0x1FFFFF) - (1 << 20),
                 static_cast<int>(k & 0x1FFFFF) - (1 << 20));
  }
  float resolution_;
  std::unordered_map<uint64_t, uint8> cells_;
};

// precomputation_grid_3d.cc:49-62
inline PrecomputationGrid3D ConvertToPrecomputationGrid(const HybridGrid& hybrid_grid) {
  PrecomputationGrid3D result(hybrid_grid.resolution());
  hybrid_grid.ForEachCell([&](const Vec3i& index, uint16 value) {
    const int cell_value = RoundToInt((ValueToProbability(value) - kMinProbability) *
                                      (255.f / (kMaxProbability - kMinProbability)));
    if (cell_value < 0 || cell_value > 255) std::abort();  // CHECK_GE / CHECK_LE
    *result.mutable_value(index) = static_cast<uint8>(cell_value);
  });
  return result;
}

// precomputation_grid_3d.cc:64-82
inline PrecomputationGrid3D PrecomputeGrid(const PrecomputationGrid3D& grid, bool half_resolution,
                                           const Vec3i& shift) {
  PrecomputationGrid3D result(grid.resolution());
  grid.ForEachCell([&](const Vec3i& index, uint8 value) {
    for (int i = 0; i != 8; ++i) {
      // GetOctant(i): (i & 1, (i >> 1) & 1, (i >> 2) & 1)   hybrid_grid.h:421-426
      const Vec3i cell_index(index.x - shift.x * (i & 1), index.y - shift.y * ((i >> 1) & 1),
                             index.z - shift.z * ((i >> 2) & 1));
      const Vec3i target = half_resolution ? Vec3i(cell_index.x >> 1, cell_index.y >> 1, cell_index.z >> 1)
                                           : cell_index;
      uint8* const cell_value = result.mutable_value(target);
      *cell_value = std::max(value, *cell_value);
    }
  });
  return result;
}

struct FastCorrelativeScanMatcherOptions3D {
  int branch_and_bound_depth;
  int full_resolution_depth;
  double min_rotational_score;
  double min_low_resolution_score;
  double linear_xy_search_window;
  double linear_z_search_window;
  double angular_search_window;
};

// fast_correlative_scan_matcher_3d.cc:57-77
class PrecomputationGridStack3D {
 public:
  PrecomputationGridStack3D(const HybridGrid& hybrid_grid, const FastCorrelativeScanMatcherOptions3D& options) {
    if (options.branch_and_bound_depth < 1 || options.full_resolution_depth < 1) std::abort();
    grids_.push_back(ConvertToPrecomputationGrid(hybrid_grid));
    int last_width = 1;
    for (int depth = 1; depth != options.branch_and_bound_depth; ++depth) {
      const bool half_resolution = depth >= options.full_resolution_depth;
      const int next_width = 1 << depth;
      const int full_voxels_per_high_resolution_voxel = 1 << std::max(0, depth - options.full_resolution_depth);
      const int shift =
          (next_width - last_width + (full_voxels_per_high_resolution_voxel - 1)) / full_voxels_per_high_resolution_voxel;
      grids_.push_back(PrecomputeGrid(grids_.back(), half_resolution, Vec3i(shift, shift, shift)));
      last_width = next_width;
    }
  }
  const PrecomputationGrid3D& Get(int depth) const { return grids_.at(depth); }
  int max_depth() const { return static_cast<int>(grids_.size()) - 1; }

 private:
  std::vector<PrecomputationGrid3D> grids_;
};

// ---- rotational scan matcher ------------------------------------------------------------
using Histogram = std::vector<float>;

// Eigen 3.3 redux_impl<Func, Derived, LinearVectorizedTraversal, NoUnrolling> on an aligned
// dynamic float vector with SSE2 packets; term(i) is the i-th coefficient of the reduced expression.
template <typename Term>
float DynRedux(int size, Term term) {
  const int packet = 4;
  const int aligned_size2 = (size / (2 * packet)) * (2 * packet);
  const int aligned_size = (size / packet) * packet;
  float res;
  if (aligned_size) {
    float p0[4], p1[4];
    for (int l = 0; l < 4; ++l) p0[l] = term(l);
    if (aligned_size > packet) {
      for (int l = 0; l < 4; ++l) p1[l] = term(packet + l);
      for (int index = 2 * packet; index < aligned_size2; index += 2 * packet)
        for (int l = 0; l < 4; ++l) {
          p0[l] = p0[l] + term(index + l);
          p1[l] = p1[l] + term(index + packet + l);
        }
      for (int l = 0; l < 4; ++l) p0[l] = p0[l] + p1[l];
      if (aligned_size > aligned_size2)
        for (int l = 0; l < 4; ++l) p0[l] = p0[l] + term(aligned_size2 + l);
    }
    res = (p0[0] + p0[2]) + (p0[1] + p0[3]);  // predux<Packet4f>, SSE2
    for (int index = aligned_size; index < size; ++index) res = res + term(index);
  } else {
    res = term(0);
    for (int index = 1; index < size; ++index) res = res + term(index);
  }
  return res;
}
inline float HistNorm(const Histogram& h) {
  return std::sqrt(DynRedux(static_cast<int>(h.size()), [&](int i) { return h[i] * h[i]; }));
}
inline float HistDot(const Histogram& a, const Histogram& b) {
  return DynRedux(static_cast<int>(a.size()), [&](int i) { return a[i] * b[i]; });
}

// rotational_scan_matcher.cc:125-144
inline Histogram RotateHistogram(const Histogram& histogram, float angle) {
  const int n = static_cast<int>(histogram.size());
  const float rotate_by_buckets = static_cast<float>(-angle * static_cast<float>(n) / M_PI);
  int full_buckets = RoundToInt(rotate_by_buckets - 0.5f);
  const float fraction = rotate_by_buckets - static_cast<float>(full_buckets);
  while (full_buckets < 0) full_buckets += n;
  Histogram out(n);
  for (int i = 0; i != n; ++i) {
    const float h0 = histogram[(i + full_buckets) % n];
    const float h1 = histogram[(i + 1 + full_buckets) % n];
    out[i] = fraction * h1 + (1.f - fraction) * h0;
  }
  return out;
}

// rotational_scan_matcher.cc:146-157
inline float MatchHistograms(const Histogram& submap_histogram, const Histogram& scan_histogram) {
  const float scan_histogram_norm = HistNorm(scan_histogram);
  const float submap_histogram_norm = HistNorm(submap_histogram);
  const float normalization = scan_histogram_norm * submap_histogram_norm;
  if (normalization < 1e-3f) return 1.f;
  return HistDot(submap_histogram, scan_histogram) / normalization;
}

namespace rsm_detail {
constexpr float kMinDistance = 0.2f;
constexpr float kMaxDistance = 0.9f;
constexpr float kSliceHeight = 0.2f;
inline float Norm2(float x, float y) { return std::sqrt(x * x + y * y); }

// Test instrumentation (not in the reference): when set, every (bucket, value) AddValueToHistogram adds is appended here
// in the order of the additions -- the histogram alone cannot tell whether a contribution of 1e-5 went into the right
// bucket once that bucket's sum is in the hundreds.
inline std::vector<std::pair<int, float>>*& ContributionTrace() {
  static thread_local std::vector<std::pair<int, float>>* trace = nullptr;
  return trace;
}
// rotational_scan_matcher.cc:35-50
inline void AddValueToHistogram(float angle, float value, Histogram* histogram) {
  while (angle > static_cast<float>(M_PI)) angle -= static_cast<float>(M_PI);
  while (angle < 0.f) angle += static_cast<float>(M_PI);
  const float zero_to_one = angle / static_cast<float>(M_PI);
  const int size = static_cast<int>(histogram->size());
  int bucket = RoundToInt(static_cast<float>(size) * zero_to_one - 0.5f);
  bucket = std::min(std::max(bucket, 0), size - 1);
  if (ContributionTrace() != nullptr) ContributionTrace()->emplace_back(bucket, value);
  (*histogram)[bucket] += value;
}
// :52-59
inline Vec3f ComputeCentroid(const PointCloud& slice) {
  Vec3f sum(0.f, 0.f, 0.f);
  for (const Vec3f& p : slice) sum = sum + p;
  const float n = static_cast<float>(slice.size());
  return Vec3f(sum.x / n, sum.y / n, sum.z / n);
}
// :61-92
inline void AddPointCloudSliceToHistogram(const PointCloud& slice, Histogram* histogram) {
  if (slice.empty()) return;
  const Vec3f centroid = ComputeCentroid(slice);
  Vec3f last_point = slice.front();
  for (const Vec3f& point : slice) {
    const float dx = point.x - last_point.x, dy = point.y - last_point.y;
    const float ex = point.x - centroid.x, ey = point.y - centroid.y;
    const float distance = Norm2(dx, dy);
    const float direction_norm = Norm2(ex, ey);
    if (distance < kMinDistance || direction_norm < kMinDistance) continue;
    if (distance > kMaxDistance) {
      last_point = point;
      continue;
    }
    const float angle = std::atan2(dy, dx);
    // normalized(): v / sqrt(squaredNorm) when squaredNorm > 0
    const float ndx = dx / distance, ndy = dy / distance;
    const float nex = ex / direction_norm, ney = ey / direction_norm;
    const float value = std::max(0.f, 1.f - std::abs(ndx * nex + ndy * ney));
    AddValueToHistogram(angle, value, histogram);
  }
}
// :97-121
inline PointCloud SortSlice(const PointCloud& slice) {
  struct Pair {
    bool operator<(const Pair& rhs) const { return angle < rhs.angle; }
    float angle;
    Vec3f point;
  };
  const Vec3f centroid = ComputeCentroid(slice);
  std::vector<Pair> by_angle;
  by_angle.reserve(slice.size());
  for (const Vec3f& point : slice) {
    const float dx = point.x - centroid.x, dy = point.y - centroid.y;
    if (Norm2(dx, dy) < kMinDistance) continue;
    by_angle.push_back(Pair{std::atan2(dy, dx), point});
  }
  std::sort(by_angle.begin(), by_angle.end());
  PointCloud result;
  for (const Pair& p : by_angle) result.push_back(p.point);
  return result;
}
}  // namespace rsm_detail

// rotational_scan_matcher.cc:161-172
inline Histogram ComputeHistogram(const PointCloud& point_cloud, int histogram_size) {
  Histogram histogram(histogram_size, 0.f);
  std::map<int, PointCloud> slices;
  for (const Vec3f& point : point_cloud) slices[RoundToInt(point.z / rsm_detail::kSliceHeight)].push_back(point);
  for (const auto& slice : slices) rsm_detail::AddPointCloudSliceToHistogram(rsm_detail::SortSlice(slice.second), &histogram);
  return histogram;
}

class RotationalScanMatcher {
 public:
  // rotational_scan_matcher.cc:174-182
  explicit RotationalScanMatcher(const std::vector<std::pair<Histogram, float>>& histograms_at_angles)
      : histogram_(histograms_at_angles.at(0).first.size(), 0.f) {
    for (const auto& ha : histograms_at_angles) {
      const Histogram r = RotateHistogram(ha.first, ha.second);
      for (size_t i = 0; i < histogram_.size(); ++i) histogram_[i] += r[i];
    }
  }
  // :184-194
  std::vector<float> Match(const Histogram& histogram, float initial_angle, const std::vector<float>& angles) const {
    std::vector<float> result;
    result.reserve(angles.size());
    for (const float angle : angles) {
      const Histogram scan_histogram = RotateHistogram(histogram, initial_angle + angle);
      result.push_back(MatchHistograms(histogram_, scan_histogram));
    }
    return result;
  }
  const Histogram& histogram() const { return histogram_; }

 private:
  Histogram histogram_;
};

// transform/transform.h:41-46 GetYaw(Quaternion)
template <typename T>
T GetYaw(const Quat<T>& rotation) {
  const Vec3<T> direction = rotation * Vec3<T>(T(1), T(0), T(0));
  return std::atan2(direction.y, direction.x);
}

// Eigen QuaternionBase::inverse(): conjugate().coeffs() / squaredNorm() when > 0, else zero.
template <typename T>
Quat<T> QuatInverse(const Quat<T>& q) {
  const T n2 = q.squaredNorm();
  if (n2 > T(0)) return Quat<T>(q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2);
  return Quat<T>(T(0), T(0), T(0), T(0));
}

// ---- the matcher -------------------------------------------------------------------------
struct DiscreteScan3D {
  Rigid3f pose;
  std::vector<std::vector<Vec3i>> cell_indices_per_depth;
  float rotational_score;
};

struct Candidate3D {
  Candidate3D(int scan_index_, const Vec3i& offset_) : scan_index(scan_index_), offset(offset_) {}
  static Candidate3D Unsuccessful() { return Candidate3D(0, Vec3i(0, 0, 0)); }
  int scan_index;
  Vec3i offset;
  float score = -std::numeric_limits<float>::infinity();
  float low_resolution_score = 0.f;
  bool operator<(const Candidate3D& other) const { return score < other.score; }
  bool operator>(const Candidate3D& other) const { return score > other.score; }
};

struct NodeData {  // the fields of TrajectoryNode::Data the matcher reads
  Quatd gravity_alignment;
  PointCloud high_resolution_point_cloud;
  PointCloud low_resolution_point_cloud;
  Histogram rotational_scan_matcher_histogram;
};

struct FastMatchResult {
  bool found = false;
  float score = 0.f;
  Rigid3d pose_estimate;
  float rotational_score = 0.f;
  float low_resolution_score = 0.f;
  // instrumentation for the parity tests / benchmarks
  long long num_scored_candidates = 0;
  int num_discrete_scans = 0;
};

class FastCorrelativeScanMatcher3D {
 public:
  FastCorrelativeScanMatcher3D(const HybridGrid& hybrid_grid, const HybridGrid* low_resolution_hybrid_grid,
                               const std::vector<std::pair<Histogram, float>>& histograms_at_angles,
                               const FastCorrelativeScanMatcherOptions3D& options)
      : options_(options),
        resolution_(hybrid_grid.resolution()),
        width_in_voxels_(hybrid_grid.grid_size()),
        stack_(hybrid_grid, options),
        low_resolution_hybrid_grid_(low_resolution_hybrid_grid),
        rotational_scan_matcher_(histograms_at_angles) {}

  const PrecomputationGridStack3D& stack() const { return stack_; }

  // fast_correlative_scan_matcher_3d.cc:147-165
  FastMatchResult Match(const Rigid3d& global_node_pose, const Rigid3d& global_submap_pose, const NodeData& data,
                        float min_score) const {
    const SearchParameters sp{RoundToInt(options_.linear_xy_search_window / resolution_),
                              RoundToInt(options_.linear_z_search_window / resolution_),
                              options_.angular_search_window, &data.low_resolution_point_cloud};
    return MatchWithSearchParameters(sp, global_node_pose.cast<float>(), global_submap_pose.cast<float>(), data,
                                     min_score);
  }

  // :204-232
  FastMatchResult MatchFullSubmap(const Quatd& global_node_rotation, const Quatd& global_submap_rotation,
                                  const NodeData& data, float min_score) const {
    float max_point_distance = 0.f;
    for (const Vec3f& p : data.high_resolution_point_cloud) max_point_distance = std::max(max_point_distance, p.norm());
    const int linear_window_size = (width_in_voxels_ + 1) / 2 + RoundToInt(max_point_distance / resolution_ + 0.5f);
    const SearchParameters sp{linear_window_size, linear_window_size, M_PI, &data.low_resolution_point_cloud};
    return MatchWithSearchParameters(sp, Rigid3f::Rotation(global_node_rotation.cast<float>()),
                                     Rigid3f::Rotation(global_submap_rotation.cast<float>()), data, min_score);
  }

  // :168-201 (fork addition)
  FastMatchResult MatchWith3DofInitial(const Rigid3d& pose_in_submap_guess, const NodeData& data, float min_score) const {
    const SearchParameters sp{RoundToInt(options_.linear_xy_search_window / resolution_),
                              RoundToInt(options_.linear_z_search_window / resolution_),
                              options_.angular_search_window, &data.low_resolution_point_cloud};
    std::vector<DiscreteScan3D> discrete_scans;
    // options_.min_rotational_score() + 0.01: double, narrowed to the float parameter
    discrete_scans.push_back(DiscretizeScan(sp, data.high_resolution_point_cloud, pose_in_submap_guess.cast<float>(),
                                            static_cast<float>(options_.min_rotational_score + 0.01)));
    return Finish(sp, discrete_scans, min_score);
  }

 private:
  struct SearchParameters {
    int linear_xy_window_size;
    int linear_z_window_size;
    double angular_search_window;
    const PointCloud* low_resolution_points;
  };

  // low_resolution_matcher.cc:23-36
  float LowResolutionScore(const SearchParameters& sp, const Rigid3f& pose) const {
    float score = 0.f;
    for (const Vec3f& p : *sp.low_resolution_points) {
      const Vec3f q = pose * p;
      score += low_resolution_hybrid_grid_->GetProbability(low_resolution_hybrid_grid_->GetCellIndex(q));
    }
    return score / static_cast<float>(sp.low_resolution_points->size());
  }

  FastMatchResult MatchWithSearchParameters(const SearchParameters& sp, const Rigid3f& global_node_pose,
                                            const Rigid3f& global_submap_pose, const NodeData& data,
                                            float min_score) const {
    const std::vector<DiscreteScan3D> discrete_scans = GenerateDiscreteScans(sp, data, global_node_pose, global_submap_pose);
    return Finish(sp, discrete_scans, min_score);
  }

  FastMatchResult Finish(const SearchParameters& sp, const std::vector<DiscreteScan3D>& discrete_scans,
                         float min_score) const {
    scored_ = 0;
    const std::vector<Candidate3D> lowest = ComputeLowestResolutionCandidates(sp, discrete_scans);
    const Candidate3D best = BranchAndBound(sp, discrete_scans, lowest, stack_.max_depth(), min_score);
    FastMatchResult r;
    r.num_scored_candidates = scored_;
    r.num_discrete_scans = static_cast<int>(discrete_scans.size());
    if (best.score > min_score) {
      r.found = true;
      r.score = best.score;
      r.pose_estimate = GetPoseFromCandidate(discrete_scans, best).cast<double>();
      r.rotational_score = discrete_scans[best.scan_index].rotational_score;
      r.low_resolution_score = best.low_resolution_score;
    }
    return r;
  }

  // :264-304
  DiscreteScan3D DiscretizeScan(const SearchParameters& sp, const PointCloud& point_cloud, const Rigid3f& pose,
                                float rotational_score) const {
    std::vector<std::vector<Vec3i>> per_depth;
    const PrecomputationGrid3D& original = stack_.Get(0);
    std::vector<Vec3i> full;
    for (const Vec3f& p : point_cloud) full.push_back(original.GetCellIndex(pose * p));
    const int full_resolution_depth = std::min(options_.full_resolution_depth, options_.branch_and_bound_depth);
    for (int i = 0; i != full_resolution_depth; ++i) per_depth.push_back(full);
    const int low_resolution_depth = options_.branch_and_bound_depth - full_resolution_depth;
    const Vec3i start(-sp.linear_xy_window_size, -sp.linear_xy_window_size, -sp.linear_z_window_size);
    for (int i = 0; i != low_resolution_depth; ++i) {
      const int e = i + 1;
      const Vec3i low_start(start.x >> e, start.y >> e, start.z >> e);
      per_depth.emplace_back();
      for (const Vec3i& c : full) {
        const Vec3i at_start = c + start;
        per_depth.back().push_back(Vec3i(at_start.x >> e, at_start.y >> e, at_start.z >> e) - low_start);
      }
    }
    return DiscreteScan3D{pose, per_depth, rotational_score};
  }

  // :306-356
  std::vector<DiscreteScan3D> GenerateDiscreteScans(const SearchParameters& sp, const NodeData& data,
                                                    const Rigid3f& global_node_pose,
                                                    const Rigid3f& global_submap_pose) const {
    std::vector<DiscreteScan3D> result;
    float max_scan_range = 3.f * resolution_;
    for (const Vec3f& p : data.high_resolution_point_cloud) max_scan_range = std::max(p.norm(), max_scan_range);
    const float kSafetyMargin = 1.f - 1e-2f;
    const float angular_step_size =
        kSafetyMargin * std::acos(1.f - Pow2(resolution_) / (2.f * Pow2(max_scan_range)));
    const int angular_window_size = RoundToInt(sp.angular_search_window / angular_step_size);
    std::vector<float> angles;
    for (int rz = -angular_window_size; rz <= angular_window_size; ++rz) angles.push_back(rz * angular_step_size);
    const Rigid3f node_to_submap = global_submap_pose.inverse() * global_node_pose;
    const Quatf gravity_inverse = QuatInverse(data.gravity_alignment).cast<float>();  // inverse in double, then cast
    const std::vector<float> scores = rotational_scan_matcher_.Match(
        data.rotational_scan_matcher_histogram, GetYaw(node_to_submap.rotation * gravity_inverse), angles);
    for (size_t i = 0; i != angles.size(); ++i) {
      if (scores[i] < options_.min_rotational_score) continue;
      const Vec3f angle_axis(0.f, 0.f, angles[i]);
      const Rigid3f pose(node_to_submap.translation,
                         QuatInverse(global_submap_pose.rotation) * AngleAxisVectorToRotationQuaternion(angle_axis) *
                             global_node_pose.rotation);
      result.push_back(DiscretizeScan(sp, data.high_resolution_point_cloud, pose, scores[i]));
    }
    return result;
  }

  // :358-392
  std::vector<Candidate3D> GenerateLowestResolutionCandidates(const SearchParameters& sp, int num_discrete_scans) const {
    const int step = 1 << stack_.max_depth();
    std::vector<Candidate3D> candidates;
    for (int scan_index = 0; scan_index != num_discrete_scans; ++scan_index)
      for (int z = -sp.linear_z_window_size; z <= sp.linear_z_window_size; z += step)
        for (int y = -sp.linear_xy_window_size; y <= sp.linear_xy_window_size; y += step)
          for (int x = -sp.linear_xy_window_size; x <= sp.linear_xy_window_size; x += step)
            candidates.emplace_back(scan_index, Vec3i(x, y, z));
    return candidates;
  }

  // :394-417
  void ScoreCandidates(int depth, const std::vector<DiscreteScan3D>& discrete_scans,
                       std::vector<Candidate3D>* candidates) const {
    const int e = std::max(0, depth - options_.full_resolution_depth + 1);
    for (Candidate3D& c : *candidates) {
      int sum = 0;
      const DiscreteScan3D& scan = discrete_scans[c.scan_index];
      const Vec3i offset(c.offset.x >> e, c.offset.y >> e, c.offset.z >> e);
      for (const Vec3i& cell : scan.cell_indices_per_depth[depth]) sum += stack_.Get(depth).value(cell + offset);
      c.score = PrecomputationGrid3D::ToProbability(sum / static_cast<float>(scan.cell_indices_per_depth[depth].size()));
    }
    scored_ += static_cast<long long>(candidates->size());
    std::sort(candidates->begin(), candidates->end(), std::greater<Candidate3D>());
  }

  std::vector<Candidate3D> ComputeLowestResolutionCandidates(const SearchParameters& sp,
                                                             const std::vector<DiscreteScan3D>& scans) const {
    std::vector<Candidate3D> lowest = GenerateLowestResolutionCandidates(sp, static_cast<int>(scans.size()));
    ScoreCandidates(stack_.max_depth(), scans, &lowest);
    return lowest;
  }

  // :431-437
  Rigid3f GetPoseFromCandidate(const std::vector<DiscreteScan3D>& scans, const Candidate3D& c) const {
    const Vec3f t(resolution_ * static_cast<float>(c.offset.x), resolution_ * static_cast<float>(c.offset.y),
                  resolution_ * static_cast<float>(c.offset.z));
    return Rigid3f::Translation(t) * scans[c.scan_index].pose;
  }

  // :439-492
  Candidate3D BranchAndBound(const SearchParameters& sp, const std::vector<DiscreteScan3D>& scans,
                             const std::vector<Candidate3D>& candidates, int candidate_depth, float min_score) const {
    if (candidate_depth == 0) {
      for (const Candidate3D& c : candidates) {
        if (c.score <= min_score) return Candidate3D::Unsuccessful();
        const float low = LowResolutionScore(sp, GetPoseFromCandidate(scans, c));
        if (low >= options_.min_low_resolution_score) {
          Candidate3D best = c;
          best.low_resolution_score = low;
          return best;
        }
      }
      return Candidate3D::Unsuccessful();
    }
    Candidate3D best = Candidate3D::Unsuccessful();
    best.score = min_score;
    for (const Candidate3D& c : candidates) {
      if (c.score <= min_score) break;
      std::vector<Candidate3D> higher;
      const int half_width = 1 << (candidate_depth - 1);
      for (int z : {0, half_width}) {
        if (c.offset.z + z > sp.linear_z_window_size) break;
        for (int y : {0, half_width}) {
          if (c.offset.y + y > sp.linear_xy_window_size) break;
          for (int x : {0, half_width}) {
            if (c.offset.x + x > sp.linear_xy_window_size) break;
            higher.emplace_back(c.scan_index, c.offset + Vec3i(x, y, z));
          }
        }
      }
      ScoreCandidates(candidate_depth - 1, scans, &higher);
      best = std::max(best, BranchAndBound(sp, scans, higher, candidate_depth - 1, best.score));
    }
    return best;
  }

  const FastCorrelativeScanMatcherOptions3D options_;
  const float resolution_;
  const int width_in_voxels_;
  PrecomputationGridStack3D stack_;
  const HybridGrid* const low_resolution_hybrid_grid_;
  RotationalScanMatcher rotational_scan_matcher_;
  mutable long long scored_ = 0;
};

}  // namespace oracle

#endif  // ORACLE_OM_FAST_CSM3D_H_
