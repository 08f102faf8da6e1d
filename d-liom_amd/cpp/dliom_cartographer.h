// Header-only C++ adapters over the C ABI (include/dliom.h) that keep the reference's operator
// surface: same class names, method names, argument meaning and error behaviour (a failed
// CHECK in the reference == std::abort() after printing the status here).
//
//   dliom::mapping::HybridGrid                                mapping/3d/hybrid_grid.h:470-547
//   dliom::mapping::RangeDataInserter3D                       mapping/3d/range_data_inserter_3d.h:35-47
//   dliom::mapping::scan_matching::RealTimeCorrelativeScanMatcher3D
//                                  .../scan_matching/real_time_correlative_scan_matcher_3d.h:34-66
//   dliom::mapping::scan_matching::CeresScanMatcher3D         .../scan_matching/ceres_scan_matcher_3d.h:37-63
//   dliom::mapping::scan_matching::FastCorrelativeScanMatcher3D
//                                  .../scan_matching/fast_correlative_scan_matcher_3d.h:100-132
//   dliom::mapping::ActiveSubmaps3D / Submap3D                mapping/3d/submap_3d.h:43-130
//   dliom::mapping::RangeDataSynchronizer                     mapping/internal/3d/range_data_synchronizer.h
//   dliom::mapping::LocalTrajectoryBuilder3D                  mapping/internal/3d/local_trajectory_builder_3d.h:83-111
//
// The value types below are layout-compatible stand-ins for Eigen::Vector3f / transform::Rigid3d
// so that this header builds without Eigen; inside cartographer the same adapters are
// instantiated on the real types: everything is funnelled through
// ToArray()/FromArray() on [tx,ty,tz,qw,qx,qy,qz] and packed float xyz.
#ifndef DLIOM_CPP_DLIOM_CARTOGRAPHER_H_
#define DLIOM_CPP_DLIOM_CARTOGRAPHER_H_

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <deque>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "../../include/dliom.h"

// -DDLIOM_ADAPTER_STAGE_TIMES (tools/wref_cpp.py --stages): microseconds between the marks of AddRangeData's single-sensor
// path, summed per mark.  Nothing of it exists in a normal build.
#ifdef DLIOM_ADAPTER_STAGE_TIMES
namespace dliom {
namespace stage_times {
inline double* table() {
  static double t[16] = {0};
  return t;
}
inline std::chrono::steady_clock::time_point& last() {
  static std::chrono::steady_clock::time_point p = std::chrono::steady_clock::now();
  return p;
}
inline void mark(int i) {
  const auto now = std::chrono::steady_clock::now();
  table()[i] += std::chrono::duration<double, std::micro>(now - last()).count();
  last() = now;
}
}  // namespace stage_times
}  // namespace dliom
#define DLIOM_ADAPTER_STAGE(i) ::dliom::stage_times::mark(i)
#else
#define DLIOM_ADAPTER_STAGE(i)
#endif

namespace dliom {

inline void Check(int status, const char* what) {
  if (status != DLIOM_OK) {
    std::fprintf(stderr, "Check failed: %s: %s %s\n", what, dliom_status_string(status),
                 dliom_last_error());
    std::abort();  // glog CHECK semantics of the reference
  }
}

// cartographer/metrics/{gauge,histogram,family_factory}.h: the interfaces LocalTrajectoryBuilder3D::RegisterMetrics
// (local_trajectory_builder_3d.h:113, .cc:624-649) is written against -- stand-ins like the value types below; inside
// cartographer the real headers take their place (same names, same virtuals).
namespace metrics {
class Gauge {
 public:
  static Gauge* Null() {
    struct NullGauge : Gauge {
      void Increment() override {}
      void Increment(double) override {}
      void Decrement() override {}
      void Decrement(double) override {}
      void Set(double) override {}
    };
    static NullGauge null_gauge;
    return &null_gauge;
  }
  virtual ~Gauge() = default;
  virtual void Increment() = 0;
  virtual void Increment(double by_value) = 0;
  virtual void Decrement() = 0;
  virtual void Decrement(double by_value) = 0;
  virtual void Set(double value) = 0;
};
class Histogram {
 public:
  using BucketBoundaries = std::vector<double>;
  static Histogram* Null() {
    struct NullHistogram : Histogram {
      void Observe(double) override {}
    };
    static NullHistogram null_histogram;
    return &null_histogram;
  }
  static BucketBoundaries FixedWidth(double width, int num_finite_buckets) {  // metrics/histogram.cc:37-46
    BucketBoundaries result;
    for (int i = 1; i <= num_finite_buckets; ++i) result.push_back(width * i);
    return result;
  }
  static BucketBoundaries ScaledPowersOf(double base, double scale_factor, double max_value) {  // :48-60
    BucketBoundaries result;
    if (!(base > 1) || !(scale_factor > 0)) Check(DLIOM_ERR_INVALID_ARGUMENT, "Histogram::ScaledPowersOf");
    for (double boundary = scale_factor; boundary < max_value; boundary *= base) result.push_back(boundary);
    return result;
  }
  virtual ~Histogram() = default;
  virtual void Observe(double value) = 0;
};
template <typename MetricType>
class Family {
 public:
  virtual ~Family() = default;
  virtual MetricType* Add(const std::map<std::string, std::string>& labels) = 0;
};
class FamilyFactory {  // (NewCounterFamily is not used on this path)
 public:
  virtual ~FamilyFactory() = default;
  virtual Family<Gauge>* NewGaugeFamily(const std::string& name, const std::string& description) = 0;
  virtual Family<Histogram>* NewHistogramFamily(const std::string& name, const std::string& description,
                                                const Histogram::BucketBoundaries& boundaries) = 0;
};
}  // namespace metrics

namespace transform {
struct Vector3d {
  double v[3];
  double x() const { return v[0]; }
  double y() const { return v[1]; }
  double z() const { return v[2]; }
};
struct Quaterniond {
  double wxyz[4];
  double w() const { return wxyz[0]; }
  double x() const { return wxyz[1]; }
  double y() const { return wxyz[2]; }
  double z() const { return wxyz[3]; }
};
class Rigid3d {
 public:
  Rigid3d() : t_{{0, 0, 0}}, q_{{1, 0, 0, 0}} {}
  Rigid3d(const Vector3d& t, const Quaterniond& q) : t_(t), q_(q) {}
  static Rigid3d Translation(const Vector3d& t) { return Rigid3d(t, Quaterniond{{1, 0, 0, 0}}); }
  const Vector3d& translation() const { return t_; }
  const Quaterniond& rotation() const { return q_; }
  std::array<double, 7> ToArray() const {
    return {{t_.v[0], t_.v[1], t_.v[2], q_.wxyz[0], q_.wxyz[1], q_.wxyz[2], q_.wxyz[3]}};
  }
  static Rigid3d FromArray(const double* a) {
    return Rigid3d(Vector3d{{a[0], a[1], a[2]}}, Quaterniond{{a[3], a[4], a[5], a[6]}});
  }

 private:
  Vector3d t_;
  Quaterniond q_;
};
}  // namespace transform

namespace sensor {
struct Vector3f {
  float x, y, z;
  Vector3f() {}  // uninitialised like Eigen::Vector3f's: resizing a cloud that a download fills does not zero it first
  Vector3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
static_assert(sizeof(Vector3f) == 12, "packed xyz like Eigen::Vector3f");
using PointCloud = std::vector<Vector3f>;
struct RangeData {
  Vector3f origin;
  PointCloud returns;
  PointCloud misses;
};
}  // namespace sensor

// One per calling thread (the reference's matchers are re-entered from pool threads:
// constraint_builder_3d.cc:320).
class Context {
 public:
  explicit Context(int device_id = 0) { Check(dliom_ctx_create(device_id, &ctx_), "dliom_ctx_create"); }
  ~Context() { dliom_ctx_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  dliom_ctx* get() const { return ctx_; }
  // The calling thread's own context on `device_id` (created on first use, destroyed with the thread): what the
  // const, concurrently called members of the reference (FastCorrelativeScanMatcher3D::Match from the
  // ConstraintBuilder3D pool threads, constraint_builder_3d.cc:270-275) run on.
  static Context* ForThisThread(int device_id = 0) {
    struct Slot {
      int device = -1;
      Context* ctx = nullptr;
      ~Slot() { delete ctx; }
    };
    static thread_local Slot slots[8];
    for (Slot& s : slots) {
      if (s.ctx != nullptr && s.device == device_id) return s.ctx;
      if (s.ctx == nullptr) {
        s.device = device_id;
        s.ctx = new Context(device_id);
        return s.ctx;
      }
    }
    Check(DLIOM_ERR_INVALID_ARGUMENT, "Context::ForThisThread: more than 8 devices per thread");
    return nullptr;
  }

 private:
  dliom_ctx* ctx_ = nullptr;
};

namespace mapping {

class HybridGrid {
 public:
  HybridGrid(Context* context, float resolution) : resolution_(resolution) {
    Check(dliom_grid_create(context->get(), resolution, &grid_), "dliom_grid_create");
  }
  ~HybridGrid() { dliom_grid_destroy(grid_); }
  HybridGrid(const HybridGrid&) = delete;
  HybridGrid& operator=(const HybridGrid&) = delete;
  float resolution() const { return resolution_; }
  dliom_grid* get() const { return grid_; }
  // hybrid_grid.h:489-491
  void SetProbability(const std::array<int32_t, 3>& index, float probability) {
    const uint16_t v = dliom_probability_to_value(probability);
    Check(dliom_grid_set_values(grid_, index.data(), &v, 1), "dliom_grid_set_values");
  }
  // hybrid_grid.h:430-435 (host arithmetic: true float division + lround)
  std::array<int32_t, 3> GetCellIndex(const sensor::Vector3f& p) const {
    return {{static_cast<int32_t>(std::lround(p.x / resolution_)),
             static_cast<int32_t>(std::lround(p.y / resolution_)),
             static_cast<int32_t>(std::lround(p.z / resolution_))}};
  }
  // HybridGrid::value for a batch of cell indices (packed int xyz).
  std::vector<uint16_t> values(const std::vector<std::array<int32_t, 3>>& cells) const {
    std::vector<uint16_t> out(cells.size());
    Check(dliom_grid_get_values(grid_, cells.empty() ? nullptr : cells[0].data(),
                                static_cast<int64_t>(cells.size()), out.data()),
          "dliom_grid_get_values");
    return out;
  }

 private:
  float resolution_;
  dliom_grid* grid_ = nullptr;
};

struct RangeDataInserterOptions3D {  // proto/3d/range_data_inserter_options_3d.proto
  double hit_probability;
  double miss_probability;
  int num_free_space_voxels;
};

class RangeDataInserter3D {
 public:
  RangeDataInserter3D(Context* context, const RangeDataInserterOptions3D& options) {
    Check(dliom_inserter_create(context->get(), options.hit_probability, options.miss_probability,
                                options.num_free_space_voxels, &inserter_),
          "dliom_inserter_create (CHECK_GT(hit, 0.5), CHECK_LT(miss, 0.5))");
  }
  ~RangeDataInserter3D() { dliom_inserter_destroy(inserter_); }
  RangeDataInserter3D(const RangeDataInserter3D&) = delete;
  RangeDataInserter3D& operator=(const RangeDataInserter3D&) = delete;
  // range_data_inserter_3d.cc:78-92
  void Insert(const sensor::RangeData& range_data, HybridGrid* hybrid_grid) const {
    if (hybrid_grid == nullptr) Check(DLIOM_ERR_INVALID_ARGUMENT, "CHECK_NOTNULL(hybrid_grid)");
    const float origin[3] = {range_data.origin.x, range_data.origin.y, range_data.origin.z};
    Check(dliom_inserter_insert(inserter_, hybrid_grid->get(), origin,
                                range_data.returns.empty() ? nullptr : &range_data.returns[0].x,
                                static_cast<int64_t>(range_data.returns.size())),
          "dliom_inserter_insert");
  }

 private:
  dliom_inserter* inserter_ = nullptr;
};

namespace scan_matching {

using RealTimeCorrelativeScanMatcherOptions = dliom_rtcsm_options;

class RealTimeCorrelativeScanMatcher3D {
 public:
  RealTimeCorrelativeScanMatcher3D(Context* context,
                                   const RealTimeCorrelativeScanMatcherOptions& options)
      : context_(context), options_(options) {}
  // real_time_correlative_scan_matcher_3d.h:47-50
  float Match(const transform::Rigid3d& initial_pose_estimate, const sensor::PointCloud& point_cloud,
              const HybridGrid& hybrid_grid, transform::Rigid3d* pose_estimate) const {
    if (pose_estimate == nullptr) Check(DLIOM_ERR_INVALID_ARGUMENT, "CHECK_NOTNULL(pose_estimate)");
    const std::array<double, 7> init = initial_pose_estimate.ToArray();
    double out[7];
    float score = 0.f;
    Check(dliom_rtcsm3d_match(context_->get(), &options_, init.data(),
                              point_cloud.empty() ? nullptr : &point_cloud[0].x,
                              static_cast<int64_t>(point_cloud.size()), hybrid_grid.get(), out, &score),
          "RealTimeCorrelativeScanMatcher3D::Match");
    *pose_estimate = transform::Rigid3d::FromArray(out);
    return score;
  }

 private:
  Context* context_;
  const RealTimeCorrelativeScanMatcherOptions options_;
};

struct CeresScanMatcherOptions3D {  // proto/scan_matching/ceres_scan_matcher_options_3d.proto
  std::vector<double> occupied_space_weight;
  double translation_weight = 0;
  double rotation_weight = 0;
  bool only_optimize_yaw = false;
  bool use_nonmonotonic_steps = false;  // ceres_solver_options
  int max_num_iterations = 50;
  int num_threads = 1;
};

using Summary = dliom_csm_summary;  // the fields of ceres::Solver::Summary the path reads

class CeresScanMatcher3D {
 public:
  using PointCloudAndHybridGridPointers = std::pair<const sensor::PointCloud*, const HybridGrid*>;

  CeresScanMatcher3D(Context* context, const CeresScanMatcherOptions3D& options) : context_(context) {
    options_.num_occupied_space_weights = static_cast<int>(options.occupied_space_weight.size());
    for (size_t i = 0; i < options.occupied_space_weight.size() && i < DLIOM_MAX_CLOUDS; ++i)
      options_.occupied_space_weight[i] = options.occupied_space_weight[i];
    options_.translation_weight = options.translation_weight;
    options_.rotation_weight = options.rotation_weight;
    options_.only_optimize_yaw = options.only_optimize_yaw;
    options_.use_nonmonotonic_steps = options.use_nonmonotonic_steps;
    options_.max_num_iterations = options.max_num_iterations;
    options_.num_threads = options.num_threads;
  }
  // ceres_scan_matcher_3d.h:51-56
  void Match(const transform::Vector3d& target_translation,
             const transform::Rigid3d& initial_pose_estimate,
             const std::vector<PointCloudAndHybridGridPointers>& point_clouds_and_hybrid_grids,
             transform::Rigid3d* pose_estimate, Summary* summary) const {
    const int k = static_cast<int>(point_clouds_and_hybrid_grids.size());
    std::vector<const float*> pts(k);
    std::vector<int64_t> n(k);
    std::vector<const dliom_grid*> grids(k);
    for (int i = 0; i < k; ++i) {
      const sensor::PointCloud& c = *point_clouds_and_hybrid_grids[i].first;
      pts[i] = c.empty() ? nullptr : &c[0].x;
      n[i] = static_cast<int64_t>(c.size());
      grids[i] = point_clouds_and_hybrid_grids[i].second->get();
    }
    const std::array<double, 7> init = initial_pose_estimate.ToArray();
    double out[7];
    Check(dliom_csm3d_match(context_->get(), &options_, target_translation.v, init.data(), k, pts.data(),
                            n.data(), grids.data(), out, summary),
          "CeresScanMatcher3D::Match");
    *pose_estimate = transform::Rigid3d::FromArray(out);
  }

 private:
  Context* context_;
  dliom_csm_options options_ = {};
};

// proto/scan_matching/fast_correlative_scan_matcher_options_3d.proto
using FastCorrelativeScanMatcherOptions3D = dliom_fast_csm_options;

// The fields of mapping::TrajectoryNode::Data the loop-closure matcher reads
// (mapping/trajectory_node.h:45-69).
struct TrajectoryNodeData {
  transform::Quaterniond gravity_alignment{{1, 0, 0, 0}};
  sensor::PointCloud high_resolution_point_cloud;
  sensor::PointCloud low_resolution_point_cloud;
  std::vector<float> rotational_scan_matcher_histogram;
};

// fast_correlative_scan_matcher_3d.h:100-132.  `nodes` are passed as the (histogram, yaw) pairs the
// reference's HistogramsAtAnglesFromNodes (fast_correlative_scan_matcher_3d.cc:114-127) extracts.
class FastCorrelativeScanMatcher3D {
 public:
  struct Result {
    float score;
    transform::Rigid3d pose_estimate;
    float rotational_score;
    float low_resolution_score;
  };

  FastCorrelativeScanMatcher3D(Context* context, const HybridGrid& hybrid_grid,
                               const HybridGrid* low_resolution_hybrid_grid,
                               const std::vector<std::pair<std::vector<float>, float>>& histograms_at_angles,
                               const FastCorrelativeScanMatcherOptions3D& options) {
    if (low_resolution_hybrid_grid == nullptr || histograms_at_angles.empty())
      Check(DLIOM_ERR_INVALID_ARGUMENT, "FastCorrelativeScanMatcher3D: nodes.at(0) / low resolution grid");
    histogram_size_ = static_cast<int>(histograms_at_angles[0].first.size());
    std::vector<float> h, a;
    for (const auto& ha : histograms_at_angles) {
      h.insert(h.end(), ha.first.begin(), ha.first.end());
      a.push_back(ha.second);
    }
    device_ = dliom_ctx_device(context->get());
    Check(dliom_fast_csm_create(context->get(), hybrid_grid.get(), low_resolution_hybrid_grid->get(), h.data(), a.data(),
                                static_cast<int>(a.size()), histogram_size_, &options, &matcher_),
          "dliom_fast_csm_create (CHECK_GE(branch_and_bound_depth, 1), CHECK_GE(full_resolution_depth, 1))");
  }
  ~FastCorrelativeScanMatcher3D() { dliom_fast_csm_destroy(matcher_); }
  FastCorrelativeScanMatcher3D(const FastCorrelativeScanMatcher3D&) = delete;
  FastCorrelativeScanMatcher3D& operator=(const FastCorrelativeScanMatcher3D&) = delete;

  // Returns false where the reference returns nullptr.  const and re-entrant like the reference's: every calling
  // thread works on its own context; the matcher (pyramid, histogram) is only read.
  bool Match(const transform::Rigid3d& global_node_pose, const transform::Rigid3d& global_submap_pose,
             const TrajectoryNodeData& constant_data, float min_score, Result* result) const {
    const dliom_fast_csm_node_data d = Data(constant_data);
    dliom_fast_csm_result r;
    Check(dliom_fast_csm_match(Context::ForThisThread(device_)->get(), matcher_, global_node_pose.ToArray().data(), global_submap_pose.ToArray().data(), &d,
                               min_score, &r),
          "FastCorrelativeScanMatcher3D::Match");
    return Store(r, result);
  }
  bool MatchFullSubmap(const transform::Quaterniond& global_node_rotation,
                       const transform::Quaterniond& global_submap_rotation, const TrajectoryNodeData& constant_data,
                       float min_score, Result* result) const {
    const dliom_fast_csm_node_data d = Data(constant_data);
    dliom_fast_csm_result r;
    Check(dliom_fast_csm_match_full_submap(Context::ForThisThread(device_)->get(), matcher_, global_node_rotation.wxyz, global_submap_rotation.wxyz, &d,
                                           min_score, &r),
          "FastCorrelativeScanMatcher3D::MatchFullSubmap");
    return Store(r, result);
  }
  bool MatchWith3DofInitial(const transform::Rigid3d& pose_in_submap_guess, const TrajectoryNodeData& constant_data,
                            float min_score, Result* result) const {
    const dliom_fast_csm_node_data d = Data(constant_data);
    dliom_fast_csm_result r;
    Check(dliom_fast_csm_match_with_3dof_initial(Context::ForThisThread(device_)->get(), matcher_, pose_in_submap_guess.ToArray().data(), &d, min_score, &r),
          "FastCorrelativeScanMatcher3D::MatchWith3DofInitial");
    return Store(r, result);
  }

 private:
  dliom_fast_csm_node_data Data(const TrajectoryNodeData& c) const {
    if (static_cast<int>(c.rotational_scan_matcher_histogram.size()) != histogram_size_)
      Check(DLIOM_ERR_INVALID_ARGUMENT, "rotational_scan_matcher_histogram size");
    dliom_fast_csm_node_data d;
    for (int i = 0; i < 4; ++i) d.gravity_alignment[i] = c.gravity_alignment.wxyz[i];
    d.high_resolution_points = c.high_resolution_point_cloud.empty() ? nullptr : &c.high_resolution_point_cloud[0].x;
    d.num_high_resolution_points = static_cast<int64_t>(c.high_resolution_point_cloud.size());
    d.low_resolution_points = c.low_resolution_point_cloud.empty() ? nullptr : &c.low_resolution_point_cloud[0].x;
    d.num_low_resolution_points = static_cast<int64_t>(c.low_resolution_point_cloud.size());
    d.rotational_scan_matcher_histogram = c.rotational_scan_matcher_histogram.data();
    return d;
  }
  static bool Store(const dliom_fast_csm_result& r, Result* result) {
    if (!r.found) return false;
    result->score = r.score;
    result->pose_estimate = transform::Rigid3d::FromArray(r.pose_estimate);
    result->rotational_score = r.rotational_score;
    result->low_resolution_score = r.low_resolution_score;
    return true;
  }
  dliom_fast_csm* matcher_ = nullptr;
  int histogram_size_ = 0;
  int device_ = 0;
};

}  // namespace scan_matching

// ---- sensor data of the LocalTrajectoryBuilder3D surface (sensor/timed_point_cloud_data.h, sensor/imu_data.h) ----
}  // namespace mapping
namespace sensor {
struct TimedPoint {
  float x, y, z, t;  // Eigen::Vector4f: position and time relative to TimedPointCloudData::time (<= 0)
};
using TimedPointCloud = std::vector<TimedPoint>;
struct TimedPointCloudData {
  int64_t time;  // common::Time ticks (100 ns): when the last point was acquired
  Vector3f origin;
  TimedPointCloud ranges;
};
struct ImuData {
  int64_t time;
  double linear_acceleration[3];
  double angular_velocity[3];
};
struct OdometryData {
  int64_t time;
  transform::Rigid3d pose;
};
// TimedPointCloudOriginData (sensor/timed_point_cloud_data.h:37-46)
struct TimedPointCloudOriginData {
  struct RangeMeasurement {
    TimedPoint point_time;
    size_t origin_index;
  };
  int64_t time = 0;
  std::vector<Vector3f> origins;
  std::vector<RangeMeasurement> ranges;
};
}  // namespace sensor
namespace mapping {

// Host logic restated from mapping/internal/3d/range_data_synchronizer.cc:29-130: the prior lidar's cloud is passed
// on, with the part of a secondary lidar's cloud that overlaps it in time merged in (second origin, times re-based,
// ranges sorted by time).  `descrew` stamps the ranges linearly over the scan period (StampRangeData, :115-130).
class RangeDataSynchronizer {
 public:
  explicit RangeDataSynchronizer(const std::vector<std::string>& expected_range_sensor_ids)
      : expected_sensor_ids_(expected_range_sensor_ids.begin(), expected_range_sensor_ids.end()),
        prior_sensor_id_(expected_range_sensor_ids.empty() ? std::string() : expected_range_sensor_ids.front()) {}

  // true: `sensor_id` is the only expected range sensor (no secondary cloud can be pending)
  bool SingleSensor(const std::string& sensor_id) const {
    return expected_sensor_ids_.size() == 1 && sensor_id == prior_sensor_id_ && secondary_cloud_.empty();
  }
  sensor::TimedPointCloudOriginData AddRangeData(const std::string& sensor_id, const sensor::TimedPointCloudData& data,
                                                 bool descrew) {
    if (expected_sensor_ids_.count(sensor_id) == 0)
      Check(DLIOM_ERR_INVALID_ARGUMENT, "CHECK_NE(expected_sensor_ids_.count(sensor_id), 0)");
    sensor::TimedPointCloudOriginData result;
    sensor::TimedPointCloudData cloud = data;
    if (descrew) StampRangeData(&cloud, 0.1);
    if (sensor_id != prior_sensor_id_) {
      secondary_cloud_.push_back(cloud);
      return result;
    }
    const double current_end = Seconds(cloud.time);
    const double current_start = cloud.ranges.empty() ? current_end : current_end + cloud.ranges.front().t;
    while (!secondary_cloud_.empty() && Seconds(secondary_cloud_.front().time) < current_start) secondary_cloud_.pop_front();
    if (secondary_cloud_.empty() || secondary_cloud_.front().ranges.empty() ||
        Seconds(secondary_cloud_.front().time) + secondary_cloud_.front().ranges.front().t > current_end) {
      ToOriginData(cloud, &result);  // no secondary cloud, or "the secondary lidar may be too fast"
      return result;
    }
    const sensor::TimedPointCloudData& sec = secondary_cloud_.front();
    const double sec_time = Seconds(sec.time);
    int i_start = -1, i_end = -1;
    for (int i = 0; i < static_cast<int>(sec.ranges.size()); ++i) {
      const double t = sec_time + sec.ranges[i].t;
      if (t >= current_start && t <= current_end && i_start == -1) i_start = i;
      if (i_start != -1 && t > current_end) {
        i_end = i - 1;
        break;
      }
    }
    if (i_start == -1) Check(DLIOM_ERR_INVALID_ARGUMENT, "CHECK(i_start != -1) range_data_synchronizer.cc:84");
    if (i_end == -1) i_end = static_cast<int>(sec.ranges.size()) - 1;
    result.time = cloud.time;
    result.origins.push_back(cloud.origin);
    for (const sensor::TimedPoint& p : data.ranges) result.ranges.push_back({p, 0});  // the UNSTAMPED input, as :97
    result.origins.push_back(sec.origin);
    for (int i = i_start; i <= i_end; ++i) {
      sensor::TimedPoint p = sec.ranges[i];
      p.t = static_cast<float>(static_cast<double>(sec.ranges[i].t) + sec_time - current_end);
      result.ranges.push_back({p, 1});
    }
    std::sort(result.ranges.begin(), result.ranges.end(),
              [](const sensor::TimedPointCloudOriginData::RangeMeasurement& a,
                 const sensor::TimedPointCloudOriginData::RangeMeasurement& b) { return a.point_time.t < b.point_time.t; });
    return result;
  }

 private:
  // common::ToSecondsStamp (common/time.cc:48-56), operation by operation: universal-time ticks (100 ns since 0001-01-01)
  // minus the Unix epoch, in nanoseconds as an integer, times 1e-9.  (ticks * 1e-7 on raw ~6.3e17 ticks would round
  // to 12.8 us; this resolves ~0.25 us like the reference and picks the same overlap indices.)
  static double Seconds(int64_t ticks) {
    constexpr int64_t kUtsEpochOffsetFromUnixEpochInSeconds = 719162ll * 24ll * 60ll * 60ll;  // common/time.h:29-30
    const int64_t ns_since_unix_epoch = (ticks - kUtsEpochOffsetFromUnixEpochInSeconds * 10000000ll) * 100ll;
    return static_cast<double>(ns_since_unix_epoch) * 1e-9;
  }
  static void ToOriginData(const sensor::TimedPointCloudData& c, sensor::TimedPointCloudOriginData* out) {
    out->time = c.time;
    out->origins.assign(1, c.origin);
    out->ranges.clear();
    for (const sensor::TimedPoint& p : c.ranges) out->ranges.push_back({p, 0});
  }
  static void StampRangeData(sensor::TimedPointCloudData* cloud, double scan_period) {
    const int n = static_cast<int>(cloud->ranges.size());
    if (n < 2) return;
    const double duration = scan_period / (n - 1);
    for (int i = 0; i < n; ++i) cloud->ranges[i].t = static_cast<float>(-scan_period + i * duration);
    cloud->ranges.back().t = 0.f;
  }
  std::set<std::string> expected_sensor_ids_;
  std::string prior_sensor_id_;
  std::deque<sensor::TimedPointCloudData> secondary_cloud_;
};

// proto::LocalTrajectoryBuilderOptions3D: the front end's options plus the AddRangeData / IMU fields
struct LocalTrajectoryBuilderOptions3D {
  // The IMU options start from the library's defaults (trajectory_builder_3d.lua's imu block): a caller overrides fields,
  // it never has to know every field -- a struct filled by hand would leave fields added later (round 4:
  // imu.tangent_preintegration, which selects the integrator) indeterminate, and dliom_imu_window_create refuses those.
  LocalTrajectoryBuilderOptions3D() : front_end() {
    dliom_imu_window_default_options(&imu);
    imu.graph_reset_every = -1;  // like the reference: reset at submaps.num_range_data, every key kept until then
  }
  dliom_front_end_options front_end;     // adaptive filters, matchers, motion filter, submaps (the caller fills it: no defaults)
  dliom_imu_window_options imu;          // imu block + WindowOptimize; imu.graph_reset_every < 0: follow
                                         // front_end.num_range_data like the reference (.cc:750), 0: never reset
  bool keep_imu_window_size = false;     // false: graph reset on => imu.window_size = 0 (the reference's rule: every key
                                         // until the reset); true: the fixed-lag smoother of imu.window_size states
  float min_range = 1.f, max_range = 100.f;
  int num_accumulated_range_data = 1;
  float voxel_filter_size = 0.15f;
  double scan_period = 0.1;
  bool enable_manual_descrew = false;    // eable_mannually_discrew_
  int rotational_histogram_size = 120;   // trajectory_builder_3d.lua: rotational_histogram_size
};

// A submap of the active pair; the grids stay owned by the front end (borrowed handles).
class Submap3D {
 public:
  Submap3D(const transform::Rigid3d& local_pose, int num_range_data, bool finished, dliom_grid* hi, dliom_grid* lo)
      : local_pose_(local_pose), num_range_data_(num_range_data), finished_(finished), hi_(hi), lo_(lo) {}
  const transform::Rigid3d& local_pose() const { return local_pose_; }
  int num_range_data() const { return num_range_data_; }
  bool finished() const { return finished_; }
  dliom_grid* high_resolution_hybrid_grid() const { return hi_; }
  dliom_grid* low_resolution_hybrid_grid() const { return lo_; }
  // Submap3D::ToProto(proto::Submap*, include_probability_grid_data) (mapping/3d/submap_3d.cc:217-230) as the
  // serialized bytes of proto::Submap{submap_3d}: proto.ParseFromString(submap.ToProtoBytes(true)) on the caller's side.
  std::string ToProtoBytes(bool include_probability_grid_data) const {
    std::string grids[2];
    if (include_probability_grid_data) {
      dliom_grid* const g[2] = {hi_, lo_};
      for (int k = 0; k < 2; ++k) {
        int64_t n = 0;
        Check(dliom_grid_to_proto(g[k], nullptr, 0, &n), "dliom_grid_to_proto");
        grids[k].resize(static_cast<size_t>(n));
        Check(dliom_grid_to_proto(g[k], reinterpret_cast<uint8_t*>(&grids[k][0]), n, &n), "dliom_grid_to_proto");
      }
    }
    const uint8_t dummy = 0;
    auto ptr = [&](const std::string& b) {
      return include_probability_grid_data ? (b.empty() ? &dummy : reinterpret_cast<const uint8_t*>(b.data())) : nullptr;
    };
    const std::array<double, 7> pose = local_pose_.ToArray();
    int64_t n = 0;
    Check(dliom_submap3d_to_proto(pose.data(), num_range_data_, finished_ ? 1 : 0, ptr(grids[0]),
                                  static_cast<int64_t>(grids[0].size()), ptr(grids[1]), static_cast<int64_t>(grids[1].size()), 1,
                                  nullptr, 0, &n), "dliom_submap3d_to_proto");
    std::string out(static_cast<size_t>(n), '\0');
    Check(dliom_submap3d_to_proto(pose.data(), num_range_data_, finished_ ? 1 : 0, ptr(grids[0]),
                                  static_cast<int64_t>(grids[0].size()), ptr(grids[1]), static_cast<int64_t>(grids[1].size()), 1,
                                  reinterpret_cast<uint8_t*>(&out[0]), n, &n), "dliom_submap3d_to_proto");
    return out;
  }

 private:
  transform::Rigid3d local_pose_;
  int num_range_data_;
  bool finished_;
  dliom_grid* hi_;
  dliom_grid* lo_;
};

// ActiveSubmaps3D (mapping/3d/submap_3d.h:95-122) over the front end's submap pair.
class ActiveSubmaps3D {
 public:
  ActiveSubmaps3D(Context* context, const dliom_front_end_options& options) : context_(context) {
    Check(dliom_front_end_create(context->get(), &options, &fe_), "dliom_front_end_create");
  }
  ~ActiveSubmaps3D() { dliom_front_end_destroy(fe_); }
  ActiveSubmaps3D(const ActiveSubmaps3D&) = delete;
  ActiveSubmaps3D& operator=(const ActiveSubmaps3D&) = delete;

  int matching_index() const {
    int i = 0;
    Check(dliom_front_end_matching_index(fe_, &i), "ActiveSubmaps3D::matching_index");
    return i;
  }
  void InsertRangeData(const sensor::RangeData& range_data, const transform::Quaterniond& gravity_alignment) {
    dliom_cloud* cloud = nullptr;
    Check(dliom_cloud_create(context_->get(), range_data.returns.empty() ? nullptr : &range_data.returns[0].x,
                             static_cast<int64_t>(range_data.returns.size()), &cloud),
          "ActiveSubmaps3D::InsertRangeData (upload)");
    dliom_insertion_result r;
    Check(dliom_front_end_insert_range_data(fe_, &range_data.origin.x, cloud, gravity_alignment.wxyz, &r),
          "ActiveSubmaps3D::InsertRangeData");
    dliom_cloud_destroy(cloud);
  }
  std::vector<std::shared_ptr<Submap3D>> submaps() const {
    int n = 0;
    Check(dliom_front_end_num_active_submaps(fe_, &n), "ActiveSubmaps3D::submaps");
    std::vector<std::shared_ptr<Submap3D>> out;
    for (int i = 0; i < n; ++i) {
      double pose[7];
      int num = 0, fin = 0;
      dliom_grid *hi = nullptr, *lo = nullptr;
      Check(dliom_front_end_active_submap(fe_, i, pose, &num, &fin, &hi, &lo), "ActiveSubmaps3D::submaps");
      out.push_back(std::make_shared<Submap3D>(transform::Rigid3d::FromArray(pose), num, fin != 0, hi, lo));
    }
    return out;
  }
  dliom_front_end* get() const { return fe_; }

 private:
  Context* context_;
  dliom_front_end* fe_ = nullptr;
};

// LocalTrajectoryBuilder3D (mapping/internal/3d/local_trajectory_builder_3d.h:83-111), steady state: the state after
// the reference's IMU-lidar initialisation (InitializeStatic / InitilizeByNDT: PCL + VINS alignment, start-up only,
// out of scope -- SURVEY 8c) is supplied through SetInitialState().  AddOdometryData is accepted and ignored like in
// the reference's D-LIOM path, whose extrapolator is never created (.cc:324-333).
class LocalTrajectoryBuilder3D {
 public:
  struct InsertionResult {
    int64_t time;
    transform::Quaterniond gravity_alignment;
    transform::Rigid3d local_pose;
    std::vector<int> insertion_submap_indices;  // trajectory-wide indices of the submaps inserted into
    bool submap_finished;                        // take it with dliom_front_end_take_finished_submap
    // TrajectoryNode::Data::rotational_scan_matcher_histogram (.cc:605-610): what the loop-closure matcher's
    // RotationalScanMatcher is built from (FastCorrelativeScanMatcher3D's `nodes`)
    std::vector<float> rotational_scan_matcher_histogram;
    // TrajectoryNode::Data::high_resolution_point_cloud / low_resolution_point_cloud (.cc:613-619): the adaptively
    // filtered clouds in the tracking frame, what ConstraintBuilder3D matches against finished submaps
    sensor::PointCloud high_resolution_point_cloud, low_resolution_point_cloud;
  };
  struct MatchingResult {
    int64_t time;
    transform::Rigid3d local_pose;
    sensor::RangeData range_data_in_local;
    std::unique_ptr<const InsertionResult> insertion_result;  // nullptr if dropped by the motion filter
  };

  LocalTrajectoryBuilder3D(Context* context, const LocalTrajectoryBuilderOptions3D& options,
                           const std::vector<std::string>& expected_range_sensor_ids)
      : context_(context), options_(options), active_submaps_(context, options.front_end),
        synchronizer_(expected_range_sensor_ids) {
    dliom_imu_window_options imu = options.imu;
    if (imu.graph_reset_every < 0) imu.graph_reset_every = options.front_end.num_range_data >= 2 ? options.front_end.num_range_data : 0;
    // with the graph reset on, WindowOptimize follows the reference's rule: every key stays in the problem until the reset
    // (window_size 0; .cc:749-797), linearisation points move by ISAM2's threshold (imu.relinearize_threshold, 0.1).  A
    // caller who wants the fixed-lag smoother between resets sets imu.window_size and keep_imu_window_size.
    if (imu.graph_reset_every >= 2 && !options.keep_imu_window_size) imu.window_size = 0;
    Check(dliom_imu_window_create(&imu, &window_), "dliom_imu_window_create");
    Check(dliom_range_accumulator_create(context->get(), &accumulator_), "dliom_range_accumulator_create");
  }
  ~LocalTrajectoryBuilder3D() {
    dliom_range_accumulator_destroy(accumulator_);
    dliom_imu_window_destroy(window_);
  }
  LocalTrajectoryBuilder3D(const LocalTrajectoryBuilder3D&) = delete;
  LocalTrajectoryBuilder3D& operator=(const LocalTrajectoryBuilder3D&) = delete;

  // prev_state_ / prev_bias_ as InitializeIMU() leaves them (.cc:322-345)
  // start_graph: the reference starts its factor graph at the FIRST WindowOptimize call after InitializeIMU (.cc:712-745),
  // i.e. the first scan after the initialisation only starts the graph and is reported (and inserted) at the initial pose --
  // harmless on the platform at rest D-LIOM's static initialisation assumes, and what this adapter does by default.  true
  // starts the graph here instead: for a caller whose initial state IS the state at the time of the call (a replay that
  // begins in motion), so that the first scan is fused like every later one.
  void SetInitialState(const transform::Rigid3d& pose, const transform::Vector3d& velocity, const double bias6[6],
                       bool start_graph = false) {
    Check(dliom_imu_window_initialize(window_, pose.ToArray().data(), velocity.v, bias6), "dliom_imu_window_initialize");
    if (start_graph) {
      double p[7], v[3], b[6];
      Check(dliom_imu_window_window_optimize(window_, pose.ToArray().data(), 0, p, v, b), "WindowOptimize (graph start)");
    }
    last_imu_time_ = -1;
    imu_initialized_ = true;
    have_prediction_ = false;
  }
  void AddImuData(const sensor::ImuData& imu) {
    if (!imu_initialized_) return;  // the reference buffers it for its initialisation
    const double dt = last_imu_time_ < 0 ? 1.0 / 500.0 : static_cast<double>(imu.time - last_imu_time_) * 1e-7;  // .cc:183-185
    last_imu_time_ = imu.time;
    if (!(dt > 0)) return;
    Check(dliom_imu_window_add_imu(window_, imu.linear_acceleration, imu.angular_velocity, dt), "AddImuData");
    have_prediction_ = true;
  }
  void AddOdometryData(const sensor::OdometryData&) {}

  std::unique_ptr<MatchingResult> AddRangeData(const std::string& sensor_id, const sensor::TimedPointCloudData& unsynchronized) {
    // One range sensor, its own time stamps, one scan per result -- D-LIOM's configurations: the synchronizer would hand
    // the scan back as it came (range_data_synchronizer.cc: no secondary cloud -> ToOriginData) and the accumulation holds
    // one scan.  The scan goes to the device as it lies in the caller's vector (TimedPoint = packed x, y, z, t) and
    // AddRangeData is ONE library call; the copies the general path makes (the synchronizer's cloud, its origin-tagged
    // ranges, the packed staging vector: ~3 MB of host traffic and two reallocating push_back loops per 64 x 1024 scan)
    // were a quarter of the adapter's time per scan (round 6, tools/wref_cpp.cc).
    if (synchronizer_.SingleSensor(sensor_id) && !options_.enable_manual_descrew && options_.num_accumulated_range_data == 1) {
      if (unsynchronized.ranges.empty() || !imu_initialized_ || !have_prediction_) return nullptr;
      if (unsynchronized.ranges.back().t > 0.1f) Check(DLIOM_ERR_INVALID_ARGUMENT, "CHECK_LE(ranges.back().point_time[3], 0.1f)");
      static_assert(sizeof(sensor::TimedPoint) == 16, "packed x, y, z, t");
      DLIOM_ADAPTER_STAGE(0);  // since the last result: the caller, AddImuData
      double prev[7], vel[3], bias[6], predicted[7], pvel[3];
      Check(dliom_imu_window_state(window_, 0, prev, vel, bias), "dliom_imu_window_state");
      Check(dliom_imu_window_predict(window_, predicted, pvel), "dliom_imu_window_predict");
      accumulation_started_ = std::chrono::steady_clock::now();
      const float origin[3] = {unsynchronized.origin.x, unsynchronized.origin.y, unsynchronized.origin.z};
      dliom_cloud* cloud = nullptr;
      float origin_in_tracking[3], current_pose[7];
      Check(dliom_add_range_data(context_->get(), prev, predicted, options_.scan_period, &unsynchronized.ranges[0].x,
                                 static_cast<int64_t>(unsynchronized.ranges.size()), origin, options_.min_range, options_.max_range,
                                 options_.voxel_filter_size, &cloud, origin_in_tracking, current_pose),
            "AddRangeData (de-skew, filters, tracking frame)");
      DLIOM_ADAPTER_STAGE(1);
      return AddAccumulatedRangeData(unsynchronized.time, current_pose, origin_in_tracking, cloud);
    }
    const sensor::TimedPointCloudOriginData sync =
        synchronizer_.AddRangeData(sensor_id, unsynchronized, options_.enable_manual_descrew);
    if (sync.ranges.empty() || !imu_initialized_ || !have_prediction_) return nullptr;
    if (sync.ranges.back().point_time.t > 0.1f) Check(DLIOM_ERR_INVALID_ARGUMENT, "CHECK_LE(ranges.back().point_time[3], 0.1f)");
    // prev_state_ and predicted_states_.back() (.cc:424-427)
    double prev[7], vel[3], bias[6], predicted[7], pvel[3];
    Check(dliom_imu_window_state(window_, 0, prev, vel, bias), "dliom_imu_window_state");
    Check(dliom_imu_window_predict(window_, predicted, pvel), "dliom_imu_window_predict");
    std::vector<float> xyzt(4 * sync.ranges.size()), index(sync.ranges.size()), origins(3 * sync.origins.size());
    for (size_t i = 0; i < sync.ranges.size(); ++i) {
      xyzt[4 * i] = sync.ranges[i].point_time.x;
      xyzt[4 * i + 1] = sync.ranges[i].point_time.y;
      xyzt[4 * i + 2] = sync.ranges[i].point_time.z;
      xyzt[4 * i + 3] = sync.ranges[i].point_time.t;
      index[i] = static_cast<float>(sync.ranges[i].origin_index);
    }
    for (size_t k = 0; k < sync.origins.size(); ++k) {
      origins[3 * k] = sync.origins[k].x;
      origins[3 * k + 1] = sync.origins[k].y;
      origins[3 * k + 2] = sync.origins[k].z;
    }
    float current_pose[7];
    int accumulated = 0;
    if (!accumulating_) {  // num_accumulated_ == 0 (.cc:389-391)
      accumulation_started_ = std::chrono::steady_clock::now();
      accumulating_ = true;
    }
    Check(dliom_range_accumulator_add(accumulator_, prev, predicted, options_.scan_period, xyzt.data(),
                                      sync.origins.size() > 1 ? index.data() : nullptr,
                                      static_cast<int64_t>(sync.ranges.size()), origins.data(),
                                      static_cast<int>(sync.origins.size()), options_.min_range, options_.max_range,
                                      options_.voxel_filter_size, current_pose, &accumulated),
          "AddRangeData (de-skew + accumulate)");
    if (accumulated < options_.num_accumulated_range_data) return nullptr;
    accumulating_ = false;
    dliom_cloud* cloud = nullptr;
    float origin_in_tracking[3];
    Check(dliom_range_accumulator_finish(accumulator_, options_.voxel_filter_size, &cloud, origin_in_tracking),
          "AddRangeData (voxel filter + tracking frame)");
    return AddAccumulatedRangeData(sync.time, current_pose, origin_in_tracking, cloud);
  }

  const ActiveSubmaps3D& active_submaps() const { return active_submaps_; }
  // local_trajectory_builder_3d.h:113, .cc:624-649: the same five metrics under the same family names, labels and bucket
  // boundaries; they are process-wide like the reference's file-scope statics and observe nothing until this is called
  static void RegisterMetrics(metrics::FamilyFactory* family_factory) {
    Metrics& m = metrics_();
    m.latency = family_factory->NewGaugeFamily("mapping_internal_3d_local_trajectory_builder_latency",
                                               "Duration from first incoming point cloud in accumulation to local slam result")
                    ->Add({});
    auto* scores = family_factory->NewHistogramFamily("mapping_internal_3d_local_trajectory_builder_scores",
                                                      "Local scan matcher scores", metrics::Histogram::FixedWidth(0.05, 20));
    m.rtcsm_score = scores->Add({{"scan_matcher", "real_time_correlative"}});
    auto* costs = family_factory->NewHistogramFamily("mapping_internal_3d_local_trajectory_builder_costs",
                                                     "Local scan matcher costs", metrics::Histogram::ScaledPowersOf(2, 0.01, 100));
    m.ceres_cost = costs->Add({{"scan_matcher", "ceres"}});
    auto* residuals = family_factory->NewHistogramFamily("mapping_internal_3d_local_trajectory_builder_residuals",
                                                         "Local scan matcher residuals", metrics::Histogram::ScaledPowersOf(2, 0.01, 10));
    m.residual_distance = residuals->Add({{"component", "distance"}});
    m.residual_angle = residuals->Add({{"component", "angle"}});
  }
  // g_vec_est_G_ of the last EstimateGravity() (.cc:1106-1154), whether it passed the gates, gravity factors added so far
  // (options.imu.enable_gravity_factor: WindowOptimize adds the Pose3GravityFactor itself, .cc:819-831)
  bool GravityEstimate(transform::Vector3d* gravity_in_global, int64_t* factors_added = nullptr) const {
    int valid = 0;
    Check(dliom_imu_window_gravity_estimate(window_, gravity_in_global->v, &valid, factors_added), "EstimateGravity");
    return valid != 0;
  }

 private:
  struct Metrics {  // kLocalSlamLatencyMetric ... kScanMatcherResidualAngleMetric (.cc:36-41)
    metrics::Gauge* latency = metrics::Gauge::Null();
    metrics::Histogram* rtcsm_score = metrics::Histogram::Null();
    metrics::Histogram* ceres_cost = metrics::Histogram::Null();
    metrics::Histogram* residual_distance = metrics::Histogram::Null();
    metrics::Histogram* residual_angle = metrics::Histogram::Null();
  };
  static Metrics& metrics_() {
    static Metrics m;
    return m;
  }
  // .cc:493-572
  std::unique_ptr<MatchingResult> AddAccumulatedRangeData(int64_t time, const float current_pose[7], const float origin[3],
                                                          dliom_cloud* cloud) {
    struct Owner {
      dliom_cloud* c;
      ~Owner() { dliom_cloud_destroy(c); }
    } owner{cloud};
    int64_t n = 0;
    Check(dliom_cloud_size(cloud, &n), "dliom_cloud_size");
    if (n == 0) return nullptr;  // "Dropped empty range data."
    double prediction[7];
    for (int i = 0; i < 7; ++i) prediction[i] = static_cast<double>(current_pose[i]);  // current_pose.cast<double>()
    dliom_match_result m;
    Check(dliom_front_end_match_cloud(active_submaps_.get(), prediction, origin, cloud, &m), "AddAccumulatedRangeData (match)");
    DLIOM_ADAPTER_STAGE(2);
    if (m.dropped) return nullptr;
    if (options_.front_end.use_online_correlative_scan_matching) metrics_().rtcsm_score->Observe(m.rtcsm_score);  // :520
    metrics_().ceres_cost->Observe(m.summary.final_cost);                                                         // :543
    metrics_().residual_distance->Observe(m.residual_distance);                                                   // :547
    metrics_().residual_angle->Observe(m.residual_angle);                                                         // :551
    // WindowOptimize(pose_estimate, false) and opt_pose = PoseFromGtsamNavState(prev_state_)
    double opt[7], vel[3], bias[6];
    // (its first call after SetInitialState only starts the graph and returns the initial state, like the reference's)
    const int ws = dliom_imu_window_window_optimize(window_, m.pose_estimate, 0, opt, vel, bias);
    // FailureDetection (.cc:856-859): ResetParams() and on with the scan -- opt_pose is what the diverged solve left in
    // prev_state_, and the next WindowOptimize starts a new graph there (the library does; failure_detections() counts)
    if (ws == DLIOM_ERR_DIVERGED)
      ++failure_detections_;
    else
      Check(ws, "WindowOptimize");
    DLIOM_ADAPTER_STAGE(3);
    have_prediction_ = false;
    std::unique_ptr<MatchingResult> result(new MatchingResult);
    result->time = time;
    result->local_pose = transform::Rigid3d::FromArray(opt);
    // filtered_range_data_in_local = TransformRangeData(in_tracking, opt_pose.cast<float>())
    // (on the device, where the cloud is: one kernel writes the moved points into pinned memory -- a download followed by
    // a host loop over ~30 000 returns was a third of the adapter's time per scan, round 6)
    float pf[7];
    for (int i = 0; i < 7; ++i) pf[i] = static_cast<float>(opt[i]);
    result->range_data_in_local.origin = TransformPoint(pf, origin[0], origin[1], origin[2]);
    result->range_data_in_local.returns.resize(static_cast<size_t>(n));
    static_assert(sizeof(sensor::Vector3f) == 12, "packed xyz");
    DLIOM_ADAPTER_STAGE(4);
    // ComputeHistogram (.cc:605-610) reads the same filtered cloud as the insertion and writes nothing the insertion
    // reads: its kernels are started first, on the context's auxiliary stream, and run beside the insertion's
    const float rot_wxyz[4] = {pf[3], pf[4], pf[5], pf[6]};
    const bool histogram_on_device = options_.rotational_histogram_size > 0 && options_.rotational_histogram_size <= 255;
    bool histogram_pending = false;
    if (histogram_on_device) {
      const int hb = dliom_cloud_rotational_histogram_begin(context_->get(), cloud, rot_wxyz, options_.rotational_histogram_size);
      if (hb != DLIOM_ERR_CAPACITY) Check(hb, "RotationalScanMatcher::ComputeHistogram (begin)");
      histogram_pending = hb == DLIOM_OK;
    }
    struct PendingHistogram {  // never leave one pending on the context, whatever path leaves this function
      dliom_ctx* c;
      bool* pending;
      ~PendingHistogram() {
        float discard[256];
        if (*pending) (void)dliom_cloud_rotational_histogram_finish(c, discard);
      }
    } pending_guard{context_->get(), &histogram_pending};
    DLIOM_ADAPTER_STAGE(5);
    // (the returns come down while the histogram's kernels run on their stream: in front of them this wait was exposed)
    Check(dliom_cloud_download_transformed(cloud, pf, &result->range_data_in_local.returns[0].x), "TransformRangeData");
    DLIOM_ADAPTER_STAGE(6);
    // InsertIntoSubmap (.cc:584-622): gravity_alignment = opt_pose.rotation()
    dliom_insertion_result ins;
    Check(dliom_front_end_insert(active_submaps_.get(), time, opt, opt + 3, &ins), "InsertIntoSubmap");
    DLIOM_ADAPTER_STAGE(7);
    if (ins.inserted) {
      std::unique_ptr<InsertionResult> ir(new InsertionResult);
      ir->time = time;
      ir->gravity_alignment = transform::Quaterniond{{opt[3], opt[4], opt[5], opt[6]}};
      ir->local_pose = result->local_pose;
      for (int i = 0; i < ins.num_insertion_submaps; ++i) ir->insertion_submap_indices.push_back(ins.insertion_submap_index[i]);
      ir->submap_finished = ins.submap_finished != 0;
      // (the two adaptively filtered clouds first: their downloads overlap what is left of the histogram)
      const dliom_cloud* filtered[2] = {nullptr, nullptr};
      Check(dliom_front_end_matched_clouds(active_submaps_.get(), &filtered[0], &filtered[1]), "matched clouds");
      sensor::PointCloud* const dst[2] = {&ir->high_resolution_point_cloud, &ir->low_resolution_point_cloud};
      for (int k = 0; k < 2; ++k) {
        int64_t m_points = 0;
        if (filtered[k] == nullptr) continue;
        Check(dliom_cloud_size(filtered[k], &m_points), "dliom_cloud_size");
        dst[k]->resize(static_cast<size_t>(m_points));
        if (m_points > 0) Check(dliom_cloud_download(filtered[k], &(*dst[k])[0].x), "dliom_cloud_download");
      }
      DLIOM_ADAPTER_STAGE(8);
      // ComputeHistogram(TransformPointCloud(filtered_range_data_in_tracking.returns, Rotation(gravity_alignment.cast<float>())), size)
      // on the device, where the filtered cloud already is (rotation fused; slices of any size -- the floor of a real scan
      // puts 15 000 returns into one 0.2 m slice); the host version only for what the device one refuses (|z| beyond
      // 409 m, non-finite coordinates, more than 63 slices above 4096 points): histogram_host_fallbacks() counts them
      if (options_.rotational_histogram_size > 0) {
        ir->rotational_scan_matcher_histogram.resize(static_cast<size_t>(options_.rotational_histogram_size));
        int hs = DLIOM_ERR_CAPACITY;
        if (histogram_pending) {
          histogram_pending = false;
          hs = dliom_cloud_rotational_histogram_finish(context_->get(), ir->rotational_scan_matcher_histogram.data());
        }
        if (hs == DLIOM_ERR_CAPACITY) {
          ++histogram_host_fallbacks_;
          const float rot[7] = {0.f, 0.f, 0.f, pf[3], pf[4], pf[5], pf[6]};
          std::vector<float> aligned(3 * static_cast<size_t>(n));  // (rare path: the rotation on the device all the same)
          Check(dliom_cloud_download_transformed(cloud, rot, aligned.data()), "TransformPointCloud (gravity alignment)");
          hs = dliom_rotational_histogram(aligned.data(), n, options_.rotational_histogram_size,
                                          ir->rotational_scan_matcher_histogram.data());
        }
        Check(hs, "RotationalScanMatcher::ComputeHistogram");
      }
      DLIOM_ADAPTER_STAGE(9);
      result->insertion_result = std::move(ir);
    }
    DLIOM_ADAPTER_STAGE(10);
    // .cc:566-568 (whole seconds, as the reference casts it)
    metrics_().latency->Set(static_cast<double>(
        std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - accumulation_started_).count()));
    return result;
  }
  // Rigid3f * Vector3f with Eigen's operation order (rotation * p + translation)
  static sensor::Vector3f TransformPoint(const float p7[7], float x, float y, float z) {
    const float w = p7[3], qx = p7[4], qy = p7[5], qz = p7[6];
    float uvx = qy * z - qz * y, uvy = qz * x - qx * z, uvz = qx * y - qy * x;
    uvx += uvx;
    uvy += uvy;
    uvz += uvz;
    const float cx = qy * uvz - qz * uvy, cy = qz * uvx - qx * uvz, cz = qx * uvy - qy * uvx;
    return sensor::Vector3f{((x + w * uvx) + cx) + p7[0], ((y + w * uvy) + cy) + p7[1], ((z + w * uvz) + cz) + p7[2]};
  }

  Context* context_;
  LocalTrajectoryBuilderOptions3D options_;
  ActiveSubmaps3D active_submaps_;
  RangeDataSynchronizer synchronizer_;
  dliom_imu_window* window_ = nullptr;
  dliom_range_accumulator* accumulator_ = nullptr;
  int64_t last_imu_time_ = -1;
  bool imu_initialized_ = false;
  bool have_prediction_ = false;
  int64_t failure_detections_ = 0;
  bool accumulating_ = false;  // num_accumulated_ > 0
  std::chrono::steady_clock::time_point accumulation_started_ = std::chrono::steady_clock::now();
  int64_t histogram_host_fallbacks_ = 0;

 public:
  // ComputeHistogram calls that the device entry point refused and the host one served
  int64_t histogram_host_fallbacks() const { return histogram_host_fallbacks_; }
  // scans after which FailureDetection fired (large velocity / bias: "reset IMU-preintegration!")
  int64_t failure_detections() const { return failure_detections_; }
};

}  // namespace mapping
}  // namespace dliom

#endif  // DLIOM_CPP_DLIOM_CARTOGRAPHER_H_
