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
//
// The value types below are layout-compatible stand-ins for Eigen::Vector3f / transform::Rigid3d
// so that this header builds without Eigen; inside cartographer the same adapters are
// instantiated on the real types: everything is funnelled through
// ToArray()/FromArray() on [tx,ty,tz,qw,qx,qy,qz] and packed float xyz.
#ifndef DLIOM_CPP_DLIOM_CARTOGRAPHER_H_
#define DLIOM_CPP_DLIOM_CARTOGRAPHER_H_

#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "../../include/dliom.h"

namespace dliom {

inline void Check(int status, const char* what) {
  if (status != DLIOM_OK) {
    std::fprintf(stderr, "Check failed: %s: %s %s\n", what, dliom_status_string(status),
                 dliom_last_error());
    std::abort();  // glog CHECK semantics of the reference
  }
}

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
}  // namespace mapping
}  // namespace dliom

#endif  // DLIOM_CPP_DLIOM_CARTOGRAPHER_H_
