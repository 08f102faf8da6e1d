// Host-side orchestration of the scan-to-submap front end over the device matchers and grids.
//
//   sensor/internal/voxel_filter.cc:28-150           VoxelFilter, AdaptiveVoxelFilter (host)
//   mapping/internal/motion_filter.cc:40-58          MotionFilter::IsSimilar
//   mapping/3d/submap_3d.cc:264-326                  Submap3D::InsertRangeData, ActiveSubmaps3D
//   mapping/internal/3d/local_trajectory_builder_3d.cc:493-572,584-622
//                                                    AddAccumulatedRangeData, InsertIntoSubmap
// The GTSAM window (WindowOptimize, :555-557) stays with the caller: match() returns the Ceres
// pose estimate, insert() takes the pose the caller optimised.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <unordered_set>
#include <vector>

#include "host_math.h"
#include "internal.h"

namespace dliom {

// ---- voxel filters (host) --------------------------------------------------------------------
struct VoxelKey {
  int x, y, z;
  bool operator==(const VoxelKey& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct VoxelKeyHash {
  size_t operator()(const VoxelKey& k) const {
    uint64_t h = 1469598103934665603ull;
    for (uint32_t v : {static_cast<uint32_t>(k.x), static_cast<uint32_t>(k.y), static_cast<uint32_t>(k.z)}) {
      h ^= v;
      h *= 1099511628211ull;
    }
    return static_cast<size_t>(h);
  }
};

// First point per voxel of edge `size` (voxel_filter.cc:81-90,126-131): lround(p / size).
static std::vector<F3> voxel_filter(float size, const std::vector<F3>& in) {
  std::unordered_set<VoxelKey, VoxelKeyHash> seen;
  seen.reserve(in.size());
  std::vector<F3> out;
  for (const F3& p : in) {
    const VoxelKey k{static_cast<int>(std::lround(p.x / size)), static_cast<int>(std::lround(p.y / size)),
                     static_cast<int>(std::lround(p.z / size))};
    if (seen.insert(k).second) out.push_back(p);
  }
  return out;
}

// voxel_filter.cc:28-37,39-77,147-150
static std::vector<F3> adaptive_voxel_filter(const dliom_adaptive_voxel_filter_options& o,
                                             const std::vector<F3>& cloud) {
  std::vector<F3> in_range;
  for (const F3& p : cloud)
    if (norm3(p) <= o.max_range) in_range.push_back(p);
  if (in_range.size() <= o.min_num_points) return in_range;
  std::vector<F3> result = voxel_filter(o.max_length, in_range);
  if (result.size() >= o.min_num_points) return result;
  for (float high = o.max_length; high > 1e-2f * o.max_length; high /= 2.f) {
    float low = high / 2.f;
    result = voxel_filter(low, in_range);
    if (result.size() >= o.min_num_points) {
      while ((high - low) / low > 1e-1f) {
        const float mid = (low + high) / 2.f;
        std::vector<F3> candidate = voxel_filter(mid, in_range);
        if (candidate.size() >= o.min_num_points) {
          low = mid;
          result.swap(candidate);
        } else {
          high = mid;
        }
      }
      return result;
    }
  }
  return result;
}

static std::vector<F3> to_f3(const float* p, int64_t n) {
  std::vector<F3> v(static_cast<size_t>(n));
  for (int64_t i = 0; i < n; ++i) v[i] = F3{p[3 * i], p[3 * i + 1], p[3 * i + 2]};
  return v;
}

}  // namespace dliom

using namespace dliom;

struct dliom_front_end {
  dliom_ctx* ctx = nullptr;
  dliom_front_end_options options;
  dliom_inserter* inserter = nullptr;
  struct Submap {  // Submap3D
    PoseD local_pose;
    dliom_grid* hi = nullptr;
    dliom_grid* lo = nullptr;
    int num_range_data = 0;
    bool finished = false;
  };
  std::vector<std::unique_ptr<Submap>> submaps;  // ActiveSubmaps3D::submaps_ (<= 2)
  std::vector<std::unique_ptr<Submap>> retired;  // finished submaps, oldest first, until the caller takes them
  int matching_submap_index = 0;
  // MotionFilter state
  int64_t num_total = 0;
  int64_t last_time = 0;
  PoseD last_pose;
  // range data of the last match (tracking frame)
  float origin[3] = {0, 0, 0};
  dliom_cloud* returns_cloud = nullptr;
  bool owns_returns_cloud = false;
  // the adaptively filtered clouds of the last match (TrajectoryNode::Data::high / low_resolution_point_cloud),
  // kept until the next match replaces them
  dliom_cloud* matched_hi = nullptr;
  dliom_cloud* matched_lo = nullptr;
  void drop_matched() {
    if (matched_hi != nullptr) dliom_cloud_destroy(matched_hi);
    if (matched_lo != nullptr) dliom_cloud_destroy(matched_lo);
    matched_hi = matched_lo = nullptr;
  }

  int add_submap(const PoseD& local_pose, int* finished_flag) {
    // submap_3d.cc:316-326
    if (submaps.size() > 1) {
      submaps.front()->finished = true;
      // a finished submap is never matched by the online matcher again: give its dense mirror back
      // (272 MB at 10 cm / +-25.6 m); the leaf pool stays for the back end
      if (submaps.front()->hi != nullptr) submaps.front()->hi->drop_dense();
      // ... and the slack of its leaf pools (sized for the worst case while the submap was active)
      // best effort: the scan IS inserted and the counters have moved; a failed reallocation only means the finished
      // submap keeps its slack (an error here would leave the caller with a rolled-back MotionFilter and a submap that
      // never rolls over)
      if (submaps.front()->hi != nullptr) (void)submaps.front()->hi->shrink_to_fit();
      if (submaps.front()->lo != nullptr) (void)submaps.front()->lo->shrink_to_fit();
      ++matching_submap_index;
      retired.push_back(std::move(submaps.front()));
      submaps.erase(submaps.begin());
      if (finished_flag != nullptr) *finished_flag = 1;
    }
    std::unique_ptr<Submap> s(new Submap);
    s->local_pose = local_pose;
    DLIOM_TRY(dliom_grid_create(ctx, static_cast<float>(options.high_resolution), &s->hi));
    DLIOM_TRY(dliom_grid_create(ctx, static_cast<float>(options.low_resolution), &s->lo));
    submaps.push_back(std::move(s));
    return DLIOM_OK;
  }
};

static int front_end_match_cloud(dliom_front_end* fe, const double pose_prediction7[7], const float origin[3],
                                 dliom_cloud* cloud, bool take_ownership, dliom_match_result* r);

static PoseD pose_from(const double* a) {
  PoseD p;
  for (int i = 0; i < 3; ++i) p.t[i] = a[i];
  for (int i = 0; i < 4; ++i) p.q[i] = a[3 + i];
  return p;
}
static void pose_to(const PoseD& p, double* a) {
  for (int i = 0; i < 3; ++i) a[i] = p.t[i];
  for (int i = 0; i < 4; ++i) a[3 + i] = p.q[i];
}

extern "C" {

int dliom_voxel_filter(float size, const float* points_xyz, int64_t n, float* out_xyz, int64_t* num_out) {
  if (n < 0 || num_out == nullptr || (n > 0 && (points_xyz == nullptr || out_xyz == nullptr)) || !(size > 0.f))
    return DLIOM_ERR_INVALID_ARGUMENT;
  const std::vector<F3> r = voxel_filter(size, to_f3(points_xyz, n));
  for (size_t i = 0; i < r.size(); ++i) {
    out_xyz[3 * i] = r[i].x;
    out_xyz[3 * i + 1] = r[i].y;
    out_xyz[3 * i + 2] = r[i].z;
  }
  *num_out = static_cast<int64_t>(r.size());
  return DLIOM_OK;
}

int dliom_adaptive_voxel_filter(const dliom_adaptive_voxel_filter_options* o, const float* points_xyz,
                                int64_t n, float* out_xyz, int64_t* num_out) {
  if (o == nullptr || n < 0 || num_out == nullptr || (n > 0 && (points_xyz == nullptr || out_xyz == nullptr)))
    return DLIOM_ERR_INVALID_ARGUMENT;
  const std::vector<F3> r = adaptive_voxel_filter(*o, to_f3(points_xyz, n));
  for (size_t i = 0; i < r.size(); ++i) {
    out_xyz[3 * i] = r[i].x;
    out_xyz[3 * i + 1] = r[i].y;
    out_xyz[3 * i + 2] = r[i].z;
  }
  *num_out = static_cast<int64_t>(r.size());
  return DLIOM_OK;
}

int dliom_front_end_create(dliom_ctx* ctx, const dliom_front_end_options* o, dliom_front_end** out) {
  if (ctx == nullptr || o == nullptr || out == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!(o->num_range_data > 0)) return DLIOM_ERR_INVALID_ARGUMENT;  // CHECK_GT(num_range_data, 0)
  std::unique_ptr<dliom_front_end> fe(new dliom_front_end);
  fe->ctx = ctx;
  fe->options = *o;
  DLIOM_TRY(dliom_inserter_create(ctx, o->hit_probability, o->miss_probability, o->num_free_space_voxels,
                                  &fe->inserter));
  PoseD identity;
  identity.t[0] = identity.t[1] = identity.t[2] = 0.0;
  identity.q[0] = 1.0;
  identity.q[1] = identity.q[2] = identity.q[3] = 0.0;
  fe->last_pose = identity;
  const int s = fe->add_submap(identity, nullptr);  // submap_3d.cc:286-295
  if (s != DLIOM_OK) {
    dliom_front_end_destroy(fe.release());
    return s;
  }
  *out = fe.release();
  return DLIOM_OK;
}

int dliom_front_end_destroy(dliom_front_end* fe) {
  if (fe == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  for (auto* list : {&fe->submaps, &fe->retired})
    for (auto& s : *list) {
      if (s->hi) dliom_grid_destroy(s->hi);
      if (s->lo) dliom_grid_destroy(s->lo);
    }
  if (fe->returns_cloud && fe->owns_returns_cloud) dliom_cloud_destroy(fe->returns_cloud);
  fe->drop_matched();
  if (fe->inserter) dliom_inserter_destroy(fe->inserter);
  delete fe;
  return DLIOM_OK;
}

int dliom_front_end_match(dliom_front_end* fe, const double pose_prediction7[7], const float origin[3],
                          const float* returns_xyz, int64_t n, dliom_match_result* r) {
  if (fe == nullptr || pose_prediction7 == nullptr || origin == nullptr || r == nullptr || n < 0 ||
      (n > 0 && returns_xyz == nullptr))
    return DLIOM_ERR_INVALID_ARGUMENT;
  // the range data moves to the device once; filters, matchers and insertion all read it there
  dliom_cloud* cloud = nullptr;
  DLIOM_TRY(dliom_cloud_create(fe->ctx, returns_xyz, n, &cloud));
  const int s = front_end_match_cloud(fe, pose_prediction7, origin, cloud, true, r);
  if (s != DLIOM_OK && fe->returns_cloud != cloud) dliom_cloud_destroy(cloud);
  return s;
}

int dliom_front_end_matched_clouds(const dliom_front_end* fe, const dliom_cloud** high_resolution,
                                   const dliom_cloud** low_resolution) {
  if (fe == nullptr || high_resolution == nullptr || low_resolution == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  *high_resolution = fe->matched_hi;
  *low_resolution = fe->matched_lo;
  return DLIOM_OK;
}

int dliom_front_end_match_cloud(dliom_front_end* fe, const double pose_prediction7[7], const float origin[3],
                                const dliom_cloud* returns, dliom_match_result* r) {
  if (fe == nullptr || pose_prediction7 == nullptr || origin == nullptr || r == nullptr || returns == nullptr)
    return DLIOM_ERR_INVALID_ARGUMENT;
  return front_end_match_cloud(fe, pose_prediction7, origin, const_cast<dliom_cloud*>(returns), false, r);
}

}  // extern "C"

// AddAccumulatedRangeData (local_trajectory_builder_3d.cc:493-553) on a device-resident cloud.
// take_ownership: the front end destroys `cloud` when the next match replaces it.
static int front_end_match_cloud(dliom_front_end* fe, const double pose_prediction7[7], const float origin[3],
                                 dliom_cloud* cloud, bool take_ownership, dliom_match_result* r) {
  std::memset(r, 0, sizeof(*r));
  const dliom_front_end_options& o = fe->options;
  std::memcpy(fe->origin, origin, sizeof(fe->origin));
  if (fe->returns_cloud != nullptr && fe->owns_returns_cloud) dliom_cloud_destroy(fe->returns_cloud);
  fe->returns_cloud = cloud;
  fe->owns_returns_cloud = take_ownership;
  fe->drop_matched();
  if (cloud->n == 0) {  // "Dropped empty range data." (:497-500)
    r->dropped = 1;
    return DLIOM_OK;
  }
  struct CloudGuard {  // the filtered clouds live for this call only
    dliom_cloud* c = nullptr;
    ~CloudGuard() {
      if (c != nullptr) dliom_cloud_destroy(c);
    }
  } hi, lo;

  const dliom_front_end::Submap& matching = *fe->submaps.front();
  const PoseD pose_prediction = pose_from(pose_prediction7);
  const PoseD prediction_in_submap = pose_mul(pose_inverse(matching.local_pose), pose_prediction);  // :504-505
  PoseD initial_ceres_pose = prediction_in_submap;
  // both adaptive filters (:507-512 and :523-533) in one joint search: they are functions of `cloud` alone, and
  // together they cost the launch / readback round trips of one
  DLIOM_TRY(dliom_cloud_adaptive_voxel_filter_pair(fe->ctx, cloud, &o.high_resolution_adaptive_voxel_filter,
                                                   &o.low_resolution_adaptive_voxel_filter, &hi.c, &lo.c));
  if (hi.c->n == 0) {
    r->dropped = 1;
    return DLIOM_OK;
  }
  r->num_high_resolution_points = hi.c->n;
  r->matching_submap_index = fe->matching_submap_index;
  double init7[7];
  pose_to(initial_ceres_pose, init7);
  if (o.use_online_correlative_scan_matching) {  // :514-521
    double out7[7];
    DLIOM_TRY(dliom_rtcsm3d_match_cloud(fe->ctx, &o.real_time_correlative_scan_matcher, init7, hi.c, matching.hi,
                                        out7, &r->rtcsm_score));
    initial_ceres_pose = pose_from(out7);
    pose_to(initial_ceres_pose, init7);
  }
  if (lo.c->n == 0) {
    r->dropped = 1;
    return DLIOM_OK;
  }
  r->num_low_resolution_points = lo.c->n;
  const dliom_cloud* clouds[2] = {hi.c, lo.c};
  const dliom_grid* grids[2] = {matching.hi, matching.lo};
  double obs7[7];
  DLIOM_TRY(dliom_csm3d_match_cloud(fe->ctx, &o.ceres_scan_matcher, prediction_in_submap.t, init7, 2, clouds, grids,
                                    obs7, &r->summary));  // :535-542
  const PoseD observation = pose_from(obs7);
  std::memcpy(r->initial_ceres_pose, init7, sizeof(init7));
  std::memcpy(r->pose_observation_in_submap, obs7, sizeof(obs7));
  // :544-551
  {
    const double dx = observation.t[0] - initial_ceres_pose.t[0], dy = observation.t[1] - initial_ceres_pose.t[1],
                 dz = observation.t[2] - initial_ceres_pose.t[2];
    r->residual_distance = std::sqrt(dx * dx + (dy * dy + dz * dz));
  }
  {
    // Eigen angularDistance: d = a * conj(b); 2 atan2(|d.vec|, |d.w|)
    const double cb[4] = {initial_ceres_pose.q[0], -initial_ceres_pose.q[1], -initial_ceres_pose.q[2],
                          -initial_ceres_pose.q[3]};
    double d[4];
    qmul_d(observation.q, cb, d);
    r->residual_angle = 2.0 * std::atan2(std::sqrt(d[1] * d[1] + d[2] * d[2] + d[3] * d[3]), std::fabs(d[0]));
  }
  pose_to(pose_mul(matching.local_pose, observation), r->pose_estimate);  // :552-553
  fe->matched_hi = hi.c;  // the guards let go: dliom_front_end_matched_clouds hands them to the caller's InsertionResult
  fe->matched_lo = lo.c;
  hi.c = lo.c = nullptr;
  return DLIOM_OK;
}

extern "C" {

}  // extern "C"

// ActiveSubmaps3D::InsertRangeData (submap_3d.cc:296-314) for range data given in the frame `pose` maps into the
// local frame: fused insertion into every active grid, counters, submap roll-over.  No MotionFilter here.
static int insert_into_active_submaps(dliom_front_end* fe, const PoseD& pose, const float origin[3], const dliom_cloud* cloud,
                                      const double gravity_alignment[4], dliom_insertion_result* r) {
  const dliom_front_end_options& o = fe->options;
  // filtered_range_data_in_local = TransformRangeData(in_tracking, opt_pose.cast<float>()) (:560-561)
  // then per submap TransformRangeData(., local_pose().inverse().cast<float>()) (submap_3d.cc:270-271)
  float poses[14];
  pose_to_float7(pose, poses);
  r->inserted = 1;
  r->num_insertion_submaps = static_cast<int>(fe->submaps.size());
  float origin_local[3] = {0, 0, 0};
  // all active grids (hi + lo of each submap) in one fused insertion: up to 4 targets
  dliom_grid* targets[4];
  float target_poses[4 * 14];
  int target_num_poses[4];
  float target_max_range[4];
  int nt = 0;
  // Submap3D::InsertRangeData takes high_resolution_max_range as an int (submap_3d.h:79-81)
  const float hi_max_range = static_cast<float>(static_cast<int>(o.high_resolution_max_range));
  for (size_t i = 0; i < fe->submaps.size(); ++i) {
    dliom_front_end::Submap& s = *fe->submaps[i];
    r->insertion_submap_index[i] = fe->matching_submap_index + static_cast<int>(i);
    pose_to_float7(pose_inverse(s.local_pose), poses + 7);
    for (int hl = 0; hl < 2; ++hl) {
      targets[nt] = hl == 0 ? s.hi : s.lo;
      std::memcpy(target_poses + 14 * nt, poses, sizeof(poses));
      target_num_poses[nt] = 2;
      target_max_range[nt] = hl == 0 ? hi_max_range : 0.f;
      ++nt;
    }
  }
  {
    const int status = dliom_inserter_insert_cloud_multi(fe->inserter, nt, targets, target_poses, target_num_poses, origin,
                                                         cloud, target_max_range);
    if (status != DLIOM_OK) {
      std::memset(r, 0, sizeof(*r));
      return status;
    }
  }
  for (auto& sm : fe->submaps) ++sm->num_range_data;
  if (fe->submaps.back()->num_range_data == o.num_range_data) {  // submap_3d.cc:310-313
    // new submap at (range_data.origin in the local frame, gravity_alignment)
    const QF qf{poses[3], poses[4], poses[5], poses[6]};
    const F3 ol = add3(qrot(qf, F3{origin[0], origin[1], origin[2]}), F3{poses[0], poses[1], poses[2]});
    origin_local[0] = ol.x;
    origin_local[1] = ol.y;
    origin_local[2] = ol.z;
    PoseD p;
    for (int i = 0; i < 3; ++i) p.t[i] = static_cast<double>(origin_local[i]);
    for (int i = 0; i < 4; ++i) p.q[i] = gravity_alignment[i];
    int finished = 0;
    DLIOM_TRY(fe->add_submap(p, &finished));
    r->submap_added = 1;
    r->submap_finished = finished;
  }
  return DLIOM_OK;
}


extern "C" {

int dliom_front_end_insert(dliom_front_end* fe, int64_t time_ticks, const double pose_estimate7[7],
                           const double gravity_alignment[4], dliom_insertion_result* r) {
  if (fe == nullptr || pose_estimate7 == nullptr || gravity_alignment == nullptr || r == nullptr)
    return DLIOM_ERR_INVALID_ARGUMENT;
  std::memset(r, 0, sizeof(*r));
  const dliom_front_end_options& o = fe->options;
  const PoseD pose = pose_from(pose_estimate7);
  // MotionFilter::IsSimilar (motion_filter.cc:40-58)
  ++fe->num_total;
  if (fe->num_total > 1) {
    const int64_t max_ticks = static_cast<int64_t>(o.motion_filter_max_time_seconds * 1e7);  // FromSeconds
    const double dx = pose.t[0] - fe->last_pose.t[0], dy = pose.t[1] - fe->last_pose.t[1],
                 dz = pose.t[2] - fe->last_pose.t[2];
    const double d2 = dx * dx + (dy * dy + dz * dz);  // Eigen Vector3d::norm() reduction order
    const PoseD rel = pose_mul(pose_inverse(pose), fe->last_pose);
    const double angle =
        2.0 * std::atan2(std::sqrt(rel.q[1] * rel.q[1] + (rel.q[2] * rel.q[2] + rel.q[3] * rel.q[3])),
                         std::fabs(rel.q[0]));
    if (time_ticks - fe->last_time <= max_ticks && std::sqrt(d2) <= o.motion_filter_max_distance_meters &&
        angle <= o.motion_filter_max_angle_radians) {
      return DLIOM_OK;  // similar: nothing inserted
    }
  }
  if (fe->returns_cloud == nullptr) {
    --fe->num_total;  // nothing happened: the filter state must not advance on an error
    return DLIOM_ERR_EMPTY_CLOUD;
  }
  const int status = insert_into_active_submaps(fe, pose, fe->origin, fe->returns_cloud, gravity_alignment, r);
  if (status != DLIOM_OK) {
    --fe->num_total;  // a failed insertion leaves MotionFilter where it was
    return status;
  }
  fe->last_time = time_ticks;  // MotionFilter::IsSimilar's state update (motion_filter.cc:54-56), now that it is final
  fe->last_pose = pose;
  return DLIOM_OK;
}

int dliom_front_end_insert_range_data(dliom_front_end* fe, const float origin_in_local[3], const dliom_cloud* returns_in_local,
                                      const double gravity_alignment[4], dliom_insertion_result* r) {
  if (fe == nullptr || origin_in_local == nullptr || returns_in_local == nullptr || gravity_alignment == nullptr || r == nullptr)
    return DLIOM_ERR_INVALID_ARGUMENT;
  std::memset(r, 0, sizeof(*r));
  PoseD identity;
  identity.t[0] = identity.t[1] = identity.t[2] = 0.0;
  identity.q[0] = 1.0;
  identity.q[1] = identity.q[2] = identity.q[3] = 0.0;
  return insert_into_active_submaps(fe, identity, origin_in_local, returns_in_local, gravity_alignment, r);
}


int dliom_front_end_num_active_submaps(const dliom_front_end* fe, int* n) {
  if (fe == nullptr || n == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  *n = static_cast<int>(fe->submaps.size());
  return DLIOM_OK;
}

int dliom_front_end_matching_index(const dliom_front_end* fe, int* index) {
  if (fe == nullptr || index == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  *index = fe->matching_submap_index;
  return DLIOM_OK;
}

int dliom_front_end_num_finished_submaps(const dliom_front_end* fe, int* n) {
  if (fe == nullptr || n == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  *n = static_cast<int>(fe->retired.size());
  return DLIOM_OK;
}

int dliom_front_end_take_finished_submap(dliom_front_end* fe, double local_pose[7], int* num_range_data, dliom_grid** hi,
                                         dliom_grid** lo) {
  if (fe == nullptr || hi == nullptr || lo == nullptr || fe->retired.empty()) return DLIOM_ERR_INVALID_ARGUMENT;
  std::unique_ptr<dliom_front_end::Submap> s = std::move(fe->retired.front());
  fe->retired.erase(fe->retired.begin());
  if (local_pose != nullptr) pose_to(s->local_pose, local_pose);
  if (num_range_data != nullptr) *num_range_data = s->num_range_data;
  *hi = s->hi;  // ownership moves to the caller: dliom_grid_destroy() when the back end is done with them
  *lo = s->lo;
  return DLIOM_OK;
}

int dliom_front_end_active_submap(const dliom_front_end* fe, int i, double local_pose[7], int* num_range_data,
                                  int* finished, dliom_grid** hi, dliom_grid** lo) {
  if (fe == nullptr || i < 0 || i >= static_cast<int>(fe->submaps.size())) return DLIOM_ERR_INVALID_ARGUMENT;
  const dliom_front_end::Submap& s = *fe->submaps[i];
  if (local_pose != nullptr) pose_to(s.local_pose, local_pose);
  if (num_range_data != nullptr) *num_range_data = s.num_range_data;
  if (finished != nullptr) *finished = s.finished ? 1 : 0;
  if (hi != nullptr) *hi = s.hi;
  if (lo != nullptr) *lo = s.lo;
  return DLIOM_OK;
}

}  // extern "C"
