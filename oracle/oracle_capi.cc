// ORACLE (test infrastructure, NOT product code).
//
// C entry points over the CPU restatement in oracle/src, loaded with ctypes by
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg ONLY.  The
// product library (d-liom_amd/csrc) never links or loads this file.
//
// Pose arrays are [tx,ty,tz,qw,qx,qy,qz] (the order of CeresPose::Data,
// optimization/ceres_pose.h:47-51); points are packed float xyz.
#include <chrono>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

#include "src/om_csm3d.h"
#include "src/om_deskew.h"
#include "src/om_fast_csm3d.h"
#include "src/om_front_end.h"
#include "src/om_grid2d.h"
#include "src/om_imu.h"
#include "src/om_rtcsm3d.h"

using namespace oracle;

namespace {

PointCloud ToCloud(const float* pts, int n) {
  PointCloud c;
  c.reserve(n);
  for (int i = 0; i < n; ++i) c.emplace_back(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
  return c;
}
Rigid3d ToRigid(const double* p) {
  return Rigid3d(Vec3d(p[0], p[1], p[2]), Quatd(p[3], p[4], p[5], p[6]));
}
void FromRigid(const Rigid3d& r, double* p) {
  p[0] = r.translation.x; p[1] = r.translation.y; p[2] = r.translation.z;
  p[3] = r.rotation.w; p[4] = r.rotation.x; p[5] = r.rotation.y; p[6] = r.rotation.z;
}
void FromRigidF(const Rigid3f& r, float* p) {
  p[0] = r.translation.x; p[1] = r.translation.y; p[2] = r.translation.z;
  p[3] = r.rotation.w; p[4] = r.rotation.x; p[5] = r.rotation.y; p[6] = r.rotation.z;
}
HybridGrid* G(void* g) { return static_cast<HybridGrid*>(g); }

}  // namespace

namespace {
template <typename T>
void ToMatrix(const Rigid3<T>& r, T m[16]) {  // Translation * Quaternion -> Transform::matrix()
  const T w = r.rotation.w, x = r.rotation.x, y = r.rotation.y, z = r.rotation.z;
  const T tx = T(2) * x, ty = T(2) * y, tz = T(2) * z;  // Eigen QuaternionBase::toRotationMatrix()
  const T twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y,
          tzz = tz * z;
  const T rot[9] = {T(1) - (tyy + tzz), txy - twz, txz + twy, txy + twz, T(1) - (txx + tzz), tyz - twx,
                    txz - twy,          tyz + twx, T(1) - (txx + tyy)};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) m[4 * i + j] = rot[3 * i + j];
    m[12 + i] = T(0);
  }
  m[3] = r.translation.x;
  m[7] = r.translation.y;
  m[11] = r.translation.z;
  m[15] = T(1);
}
template <typename T>
double ApproxRatio(const Rigid3<T>& a, const Rigid3<T>& b) {
  T ma[16], mb[16];
  ToMatrix(a, ma);
  ToMatrix(b, mb);
  double d2 = 0., na = 0., nb = 0.;
  for (int i = 0; i < 16; ++i) {
    d2 += static_cast<double>(ma[i] - mb[i]) * static_cast<double>(ma[i] - mb[i]);
    na += static_cast<double>(ma[i]) * ma[i];
    nb += static_cast<double>(mb[i]) * mb[i];
  }
  return std::sqrt(d2) / (static_cast<double>(std::numeric_limits<T>::epsilon()) * std::sqrt(std::min(na, nb)));
}
template <typename T>
double RigidTransformKat() {
  double worst = 0.;
  for (int test = 0; test < 2; ++test) {  // every TYPED_TEST has a fresh fixture: prng_(42)
    std::mt19937 prng(42);
    std::uniform_real_distribution<T> distribution(-1., 1.);
    const T x = T(0.7) * distribution(prng), y = T(0.7) * distribution(prng), z = T(0.7) * distribution(prng);
    const T ax = T(0.7) * distribution(prng), ay = T(0.7) * distribution(prng), az = T(0.7) * distribution(prng);
    const Rigid3<T> pose(Vec3<T>(x, y, z), AngleAxisVectorToRotationQuaternion(Vec3<T>(ax, ay, az)));
    const Rigid3<T> identity;
    if (test == 0) {
      worst = std::max(worst, ApproxRatio(pose * identity, pose));
      worst = std::max(worst, ApproxRatio(identity * pose, pose));
    } else {
      worst = std::max(worst, ApproxRatio(pose.inverse() * pose, identity));
      worst = std::max(worst, ApproxRatio(pose * pose.inverse(), identity));
    }
  }
  return worst;
}
}  // namespace

extern "C" {

// ---------------------------------------------------------------- probability values
void orc_value_to_probability_table(float* out65536) {
  const std::vector<float>& t = ValueToProbabilityTable();
  std::memcpy(out65536, t.data(), 65536 * sizeof(float));
}
void orc_value_to_correspondence_cost_table(float* out65536) {
  const std::vector<float>& t = ValueToCorrespondenceCostTable();
  std::memcpy(out65536, t.data(), 65536 * sizeof(float));
}
uint16_t orc_probability_to_value(float p) { return ProbabilityToValue(p); }
uint16_t orc_correspondence_cost_to_value(float c) { return CorrespondenceCostToValue(c); }
float orc_odds(float p) { return Odds(p); }
float orc_probability_from_odds(float o) { return ProbabilityFromOdds(o); }
uint16_t orc_probability_value_to_correspondence_cost_value(uint16_t v) {
  return ProbabilityValueToCorrespondenceCostValue(v);
}
uint16_t orc_correspondence_cost_value_to_probability_value(uint16_t v) {
  return CorrespondenceCostValueToProbabilityValue(v);
}
void orc_lookup_table_to_apply_odds(float odds, uint16_t* out32768) {
  const std::vector<uint16> t = ComputeLookupTableToApplyOdds(odds);
  std::memcpy(out32768, t.data(), 32768 * sizeof(uint16_t));
}
void orc_lookup_table_to_apply_correspondence_cost_odds(float odds, uint16_t* out32768) {
  const std::vector<uint16> t = ComputeLookupTableToApplyCorrespondenceCostOdds(odds);
  std::memcpy(out32768, t.data(), 32768 * sizeof(uint16_t));
}

// ---------------------------------------------------------------- HybridGrid
void* orc_grid_new(float resolution) { return new HybridGrid(resolution); }
void orc_grid_free(void* g) { delete G(g); }
float orc_grid_resolution(void* g) { return G(g)->resolution(); }
int orc_grid_bits(void* g) { return G(g)->bits(); }
void orc_cell_indices(float resolution, const float* pts, int n, int* out) {
  HybridGrid tmp(resolution);
  for (int i = 0; i < n; ++i) {
    const Vec3i c = tmp.GetCellIndex(Vec3f(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
    out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
  }
}
void orc_grid_center_of_cell(void* g, int x, int y, int z, float* out3) {
  const Vec3f c = G(g)->GetCenterOfCell(Vec3i(x, y, z));
  out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
}
void orc_grid_set_probability(void* g, int x, int y, int z, float p) {
  G(g)->SetProbability(Vec3i(x, y, z), p);
}
void orc_grid_set_value(void* g, int x, int y, int z, uint16_t v) {
  *G(g)->mutable_value(Vec3i(x, y, z)) = v;
}
void orc_grid_set_values(void* g, const int* xyz, const uint16_t* v, int n) {
  for (int i = 0; i < n; ++i)
    *G(g)->mutable_value(Vec3i(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2])) = v[i];
}
uint16_t orc_grid_value(void* g, int x, int y, int z) { return G(g)->value(Vec3i(x, y, z)); }
void orc_grid_values(void* g, const int* xyz, int n, uint16_t* out) {
  for (int i = 0; i < n; ++i)
    out[i] = G(g)->value(Vec3i(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
}
float orc_grid_probability(void* g, int x, int y, int z) {
  return G(g)->GetProbability(Vec3i(x, y, z));
}
int orc_grid_is_known(void* g, int x, int y, int z) { return G(g)->IsKnown(Vec3i(x, y, z)); }
int orc_grid_apply_lookup_table(void* g, int x, int y, int z, const uint16_t* table32768) {
  const std::vector<uint16> t(table32768, table32768 + 32768);
  return G(g)->ApplyLookupTable(Vec3i(x, y, z), t) ? 1 : 0;
}
void orc_grid_finish_update(void* g) { G(g)->FinishUpdate(); }
int64_t orc_grid_num_cells(void* g) {
  int64_t n = 0;
  G(g)->ForEachCell([&](const Vec3i&, uint16) { ++n; });
  return n;
}
// Cells in the reference iterator's order.
void orc_grid_export_cells(void* g, int* xyz, uint16_t* values) {
  int64_t i = 0;
  G(g)->ForEachCell([&](const Vec3i& c, uint16 v) {
    xyz[3 * i] = c.x; xyz[3 * i + 1] = c.y; xyz[3 * i + 2] = c.z;
    values[i] = v;
    ++i;
  });
}
int64_t orc_grid_num_leaves(void* g) {
  int64_t n = 0;
  G(g)->ForEachLeaf([&](const Vec3i&, const uint16*) { ++n; });
  return n;
}
// Leaves as (origin voxel index of the 8^3 block, 512 z-major values).
void orc_grid_export_leaves(void* g, int* origin_xyz, uint16_t* values512) {
  int64_t i = 0;
  G(g)->ForEachLeaf([&](const Vec3i& o, const uint16* cells) {
    origin_xyz[3 * i] = o.x; origin_xyz[3 * i + 1] = o.y; origin_xyz[3 * i + 2] = o.z;
    std::memcpy(values512 + 512 * i, cells, 512 * sizeof(uint16_t));
    ++i;
  });
}

// ---------------------------------------------------------------- insertion / filters
void orc_insert_range_data(void* g, const float* origin3, const float* returns, int n,
                           double hit_probability, double miss_probability,
                           int num_free_space_voxels) {
  // RangeDataInserter3D ctor: Odds(float(options.hit_probability()))
  const RangeDataInserter3D inserter(static_cast<float>(hit_probability),
                                     static_cast<float>(miss_probability),
                                     num_free_space_voxels);
  RangeData rd{Vec3f(origin3[0], origin3[1], origin3[2]), ToCloud(returns, n), {}};
  inserter.Insert(rd, G(g));
}
// Same, but with prebuilt tables (amortises the 2 x 32768-entry table build).
void orc_insert_range_data_tables(void* g, const float* origin3, const float* returns, int n,
                                  const uint16_t* hit_table, const uint16_t* miss_table,
                                  int num_free_space_voxels) {
  const std::vector<uint16> hit(hit_table, hit_table + 32768);
  const std::vector<uint16> miss(miss_table, miss_table + 32768);
  HybridGrid* grid = G(g);
  const Vec3f origin(origin3[0], origin3[1], origin3[2]);
  for (int i = 0; i < n; ++i)
    grid->ApplyLookupTable(
        grid->GetCellIndex(Vec3f(returns[3 * i], returns[3 * i + 1], returns[3 * i + 2])), hit);
  const Vec3i origin_cell = grid->GetCellIndex(origin);
  for (int i = 0; i < n; ++i) {
    const Vec3i hit_cell =
        grid->GetCellIndex(Vec3f(returns[3 * i], returns[3 * i + 1], returns[3 * i + 2]));
    const Vec3i delta = hit_cell - origin_cell;
    const int num_samples =
        std::max(std::abs(delta.x), std::max(std::abs(delta.y), std::abs(delta.z)));
    for (int position = std::max(0, num_samples - num_free_space_voxels);
         position < num_samples; ++position) {
      grid->ApplyLookupTable(Vec3i(origin_cell.x + delta.x * position / num_samples,
                                   origin_cell.y + delta.y * position / num_samples,
                                   origin_cell.z + delta.z * position / num_samples),
                             miss);
    }
  }
  grid->FinishUpdate();
}
// Returns the number of surviving points and their indices.
int orc_voxel_filter(float size, const float* pts, int n, int* keep_idx) {
  VoxelFilter f(size);
  const std::vector<int> kept = f.FilterIndices(ToCloud(pts, n));
  std::memcpy(keep_idx, kept.data(), kept.size() * sizeof(int));
  return static_cast<int>(kept.size());
}
int orc_adaptive_voxel_filter(float max_length, float min_num_points, float max_range,
                              const float* pts, int n, float* out_pts) {
  const PointCloud r = AdaptiveVoxelFilter(
      AdaptiveVoxelFilterOptions{max_length, min_num_points, max_range}, ToCloud(pts, n));
  for (size_t i = 0; i < r.size(); ++i) {
    out_pts[3 * i] = r[i].x; out_pts[3 * i + 1] = r[i].y; out_pts[3 * i + 2] = r[i].z;
  }
  return static_cast<int>(r.size());
}
// sensor::TransformPointCloud with a float pose.
void orc_transform_points(const float* pose7, const float* pts, int n, float* out) {
  const Rigid3f t(Vec3f(pose7[0], pose7[1], pose7[2]),
                  Quatf(pose7[3], pose7[4], pose7[5], pose7[6]));
  for (int i = 0; i < n; ++i) {
    const Vec3f p = t * Vec3f(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    out[3 * i] = p.x; out[3 * i + 1] = p.y; out[3 * i + 2] = p.z;
  }
}
// Rigid3d composition / inverse / cast, for building test poses.
void orc_rigid3d_multiply(const double* a7, const double* b7, double* out7) {
  FromRigid(ToRigid(a7) * ToRigid(b7), out7);
}
void orc_rigid3d_inverse(const double* a7, double* out7) { FromRigid(ToRigid(a7).inverse(), out7); }

// ---------------------------------------------------------------- RTCSM3D
// opts = [linear_search_window, angular_search_window, translation_delta_cost_weight,
//         rotation_delta_cost_weight]
void orc_rtcsm3d_window(const double* opts, float resolution, const float* pts, int n,
                        int* linear_window, int* angular_window, float* angular_step,
                        float* max_scan_range) {
  const RealTimeCorrelativeScanMatcher3D m(
      RealTimeCorrelativeScanMatcherOptions{opts[0], opts[1], opts[2], opts[3]});
  const SearchWindow3D w = m.ComputeWindow(resolution, ToCloud(pts, n));
  *linear_window = w.linear_window_size;
  *angular_window = w.angular_window_size;
  *angular_step = w.angular_step_size;
  *max_scan_range = w.max_scan_range;
}
// Writes, per candidate in generation order, transform (7 floats) and
// candidate = float(init) * transform (7 floats).  Returns the count; pass
// null outputs to query it.
int64_t orc_rtcsm3d_candidates(const double* opts, float resolution, const float* pts, int n,
                               const double* init7, float* transforms7, float* candidates7) {
  const RealTimeCorrelativeScanMatcher3D m(
      RealTimeCorrelativeScanMatcherOptions{opts[0], opts[1], opts[2], opts[3]});
  const std::vector<Rigid3f> ts = m.GenerateExhaustiveSearchTransforms(resolution, ToCloud(pts, n));
  if (transforms7 != nullptr || candidates7 != nullptr) {
    const Rigid3f init = ToRigid(init7).cast<float>();
    for (size_t i = 0; i < ts.size(); ++i) {
      if (transforms7 != nullptr) FromRigidF(ts[i], transforms7 + 7 * i);
      if (candidates7 != nullptr) FromRigidF(init * ts[i], candidates7 + 7 * i);
    }
  }
  return static_cast<int64_t>(ts.size());
}
// Full match.  scores (may be null) receives every candidate's score.
float orc_rtcsm3d_match(const double* opts, const double* init7, const float* pts, int n,
                        void* grid, double* out7, float* scores, int* best_index) {
  const RealTimeCorrelativeScanMatcher3D m(
      RealTimeCorrelativeScanMatcherOptions{opts[0], opts[1], opts[2], opts[3]});
  Rigid3d pose;
  std::vector<float> all;
  int best = -1;
  const float s = m.Match(ToRigid(init7), ToCloud(pts, n), *G(grid), &pose,
                          scores != nullptr ? &all : nullptr, &best);
  FromRigid(pose, out7);
  if (scores != nullptr) std::memcpy(scores, all.data(), all.size() * sizeof(float));
  if (best_index != nullptr) *best_index = best;
  return s;
}
// The reference's Match loop (rtcsm_3d.cc:40-51) restricted to candidates
// [first, first+count): per candidate a fresh TransformPointCloud allocation
// and the sequential float score, exactly as the reference pays for them.
// Used by bench.py's cpu_baseline leg to time a bounded sample.  Returns the
// best score of the range and its index.
float orc_rtcsm3d_match_range(const double* opts, const double* init7, const float* pts, int n,
                              void* grid, int64_t first, int64_t count, int64_t* best_index) {
  const RealTimeCorrelativeScanMatcher3D m(
      RealTimeCorrelativeScanMatcherOptions{opts[0], opts[1], opts[2], opts[3]});
  const PointCloud cloud = ToCloud(pts, n);
  const HybridGrid& g = *G(grid);
  const std::vector<Rigid3f> ts = m.GenerateExhaustiveSearchTransforms(g.resolution(), cloud);
  const Rigid3f init = ToRigid(init7).cast<float>();
  const int64_t end = std::min<int64_t>(first + count, static_cast<int64_t>(ts.size()));
  float best = -1.f;
  int64_t best_c = -1;
  for (int64_t c = first; c < end; ++c) {
    const Rigid3f candidate = init * ts[c];
    const float score = m.ScoreCandidate(g, TransformPointCloud(cloud, candidate), ts[c]);
    if (score > best) {
      best = score;
      best_c = c;
    }
  }
  if (best_index != nullptr) *best_index = best_c;
  return best;
}
// BASELINE.md section 2, variant (ii) "fair-CPU": the same arithmetic as orc_rtcsm3d_match_range -- Rigid3f * point, true
// division + lround for the cell, the LUT probability, the sequential float sum, first strictly greater score -- but a
// FLAT leaf table instead of the two pointer levels (orc_flat_grid_new, built once per grid like an index a CPU
// implementation would maintain beside the tree) and no TransformPointCloud allocation per candidate.
struct FlatGridView {
  float resolution;
  int half, leaves;  // voxel shift, leaves per axis
  std::vector<const uint16*> table;
};
void* orc_flat_grid_new(void* grid) {
  const HybridGrid& g = *G(grid);
  FlatGridView* f = new FlatGridView;
  f->resolution = g.resolution();
  f->half = g.grid_size() >> 1;
  f->leaves = g.grid_size() >> 3;
  f->table.assign(static_cast<size_t>(f->leaves) * f->leaves * f->leaves, nullptr);
  g.ForEachLeaf([&](const Vec3i& origin, const uint16* cells) {
    const int lx = (origin.x + f->half) >> 3, ly = (origin.y + f->half) >> 3, lz = (origin.z + f->half) >> 3;
    f->table[(static_cast<size_t>(lz) * f->leaves + ly) * f->leaves + lx] = cells;
  });
  return f;
}
void orc_flat_grid_free(void* flat) { delete static_cast<FlatGridView*>(flat); }
float orc_rtcsm3d_match_range_fair(const double* opts, const double* init7, const float* pts, int n, void* grid, void* flat,
                                   int64_t first, int64_t count, int64_t* best_index) {
  const RealTimeCorrelativeScanMatcherOptions o{opts[0], opts[1], opts[2], opts[3]};
  const RealTimeCorrelativeScanMatcher3D m(o);
  const PointCloud cloud = ToCloud(pts, n);
  const HybridGrid& g = *G(grid);
  const FlatGridView& f = *static_cast<FlatGridView*>(flat);
  const std::vector<Rigid3f> ts = m.GenerateExhaustiveSearchTransforms(g.resolution(), cloud);
  const Rigid3f init = ToRigid(init7).cast<float>();
  const int64_t end = std::min<int64_t>(first + count, static_cast<int64_t>(ts.size()));
  const unsigned gsize = static_cast<unsigned>(2 * f.half);
  float best = -1.f;
  int64_t best_c = -1;
  for (int64_t c = first; c < end; ++c) {
    const Rigid3f candidate = init * ts[c];
    float score = 0.f;
    for (const Vec3f& p : cloud) {
      const Vec3i i = g.GetCellIndex(candidate * p);
      const unsigned sx = static_cast<unsigned>(i.x + f.half), sy = static_cast<unsigned>(i.y + f.half), sz = static_cast<unsigned>(i.z + f.half);
      uint16 v = 0;
      if (sx < gsize && sy < gsize && sz < gsize) {
        const uint16* leaf = f.table[(static_cast<size_t>(sz >> 3) * f.leaves + (sy >> 3)) * f.leaves + (sx >> 3)];
        if (leaf != nullptr) v = leaf[((sz & 7u) << 6) | ((sy & 7u) << 3) | (sx & 7u)];
      }
      score += ValueToProbability(v);
    }
    score /= static_cast<float>(cloud.size());
    const float angle = GetAngle(ts[c]);
    score *= std::exp(-Pow2(ts[c].translation.norm() * o.translation_delta_cost_weight + angle * o.rotation_delta_cost_weight));
    if (score > best) {
      best = score;
      best_c = c;
    }
  }
  if (best_index != nullptr) *best_index = best_c;
  return best;
}
// The whole of RealTimeCorrelativeScanMatcher3D::Match's loop body (real_time_correlative_scan_matcher_3d.cc:40-51 with
// ScoreCandidate :97-113) for candidates [first, first + count), in the fair-CPU layout, keeping EVERY candidate's
// reference score and, from the same lookups, its integer value sum (what the HIP score volume holds): the full-size
// config-5 parity run (tools/config5_full_parity.py) compares all C of both with the device.
void orc_rtcsm3d_range_fair_volume(const double* opts, const double* init7, const float* pts, int n, void* grid, void* flat,
                                   int64_t first, int64_t count, uint64_t* value_sums, float* scores) {
  const RealTimeCorrelativeScanMatcherOptions o{opts[0], opts[1], opts[2], opts[3]};
  const RealTimeCorrelativeScanMatcher3D m(o);
  const PointCloud cloud = ToCloud(pts, n);
  const HybridGrid& g = *G(grid);
  const FlatGridView& f = *static_cast<FlatGridView*>(flat);
  const std::vector<Rigid3f> ts = m.GenerateExhaustiveSearchTransforms(g.resolution(), cloud);
  const Rigid3f init = ToRigid(init7).cast<float>();
  const int64_t end = std::min<int64_t>(first + count, static_cast<int64_t>(ts.size()));
  const unsigned gsize = static_cast<unsigned>(2 * f.half);
  for (int64_t c = first; c < end; ++c) {
    const Rigid3f candidate = init * ts[c];
    float score = 0.f;
    uint64_t sum = 0;
    for (const Vec3f& p : cloud) {
      const Vec3i i = g.GetCellIndex(candidate * p);
      const unsigned sx = static_cast<unsigned>(i.x + f.half), sy = static_cast<unsigned>(i.y + f.half), sz = static_cast<unsigned>(i.z + f.half);
      uint16 v = 0;
      if (sx < gsize && sy < gsize && sz < gsize) {
        const uint16* leaf = f.table[(static_cast<size_t>(sz >> 3) * f.leaves + (sy >> 3)) * f.leaves + (sx >> 3)];
        if (leaf != nullptr) v = leaf[((sz & 7u) << 6) | ((sy & 7u) << 3) | (sx & 7u)];
      }
      score += ValueToProbability(v);
      const uint16 w = v & 0x7fff;
      sum += w == 0 ? 1 : w;
    }
    score /= static_cast<float>(cloud.size());
    const float angle = GetAngle(ts[c]);
    score *= std::exp(-Pow2(ts[c].translation.norm() * o.translation_delta_cost_weight + angle * o.rotation_delta_cost_weight));
    value_sums[c - first] = sum;
    scores[c - first] = score;
  }
}
// Per candidate: sum over points of max(value & 0x7fff, 1) (exact integers),
// the order-independent quantity the HIP score-volume kernel accumulates.
// Candidates [first, first+count) only (count<0: all).
void orc_rtcsm3d_value_sums(const double* opts, const double* init7, const float* pts, int n,
                            void* grid, int64_t first, int64_t count, uint64_t* sums) {
  const RealTimeCorrelativeScanMatcher3D m(
      RealTimeCorrelativeScanMatcherOptions{opts[0], opts[1], opts[2], opts[3]});
  const PointCloud cloud = ToCloud(pts, n);
  const HybridGrid& g = *G(grid);
  const std::vector<Rigid3f> ts = m.GenerateExhaustiveSearchTransforms(g.resolution(), cloud);
  const Rigid3f init = ToRigid(init7).cast<float>();
  const int64_t end = count < 0 ? static_cast<int64_t>(ts.size()) : first + count;
  for (int64_t c = first; c < end; ++c) {
    const Rigid3f cand = init * ts[c];
    uint64_t s = 0;
    for (const Vec3f& p : cloud) {
      const uint16 v = g.value(g.GetCellIndex(cand * p)) & 0x7fff;
      s += v == 0 ? 1 : v;
    }
    sums[c - first] = s;
  }
}
// The sequential float sum of probabilities (rtcsm_3d.cc:101-104, before `score /= N`) for k
// candidates given by index.
void orc_rtcsm3d_float_sums(const double* opts, const double* init7, const float* pts, int n, void* grid,
                            const int64_t* indices, int64_t k, float* sums) {
  const RealTimeCorrelativeScanMatcher3D m(
      RealTimeCorrelativeScanMatcherOptions{opts[0], opts[1], opts[2], opts[3]});
  const PointCloud cloud = ToCloud(pts, n);
  const HybridGrid& g = *G(grid);
  const std::vector<Rigid3f> ts = m.GenerateExhaustiveSearchTransforms(g.resolution(), cloud);
  const Rigid3f init = ToRigid(init7).cast<float>();
  for (int64_t i = 0; i < k; ++i) {
    const Rigid3f cand = init * ts[indices[i]];
    float score = 0.f;
    for (const Vec3f& p : TransformPointCloud(cloud, cand)) score += g.GetProbability(g.GetCellIndex(p));
    sums[i] = score;
  }
}
// Candidates given by index (huge windows where the full loop would take hours): the integer value sum and the
// reference's score (ScoreCandidate, rtcsm_3d.cc:97-113) of each.  The candidate list is generated once per call.
void orc_rtcsm3d_at(const double* opts, const double* init7, const float* pts, int n, void* grid,
                    const int64_t* indices, int64_t k, uint64_t* value_sums, float* scores) {
  const RealTimeCorrelativeScanMatcher3D m(
      RealTimeCorrelativeScanMatcherOptions{opts[0], opts[1], opts[2], opts[3]});
  const PointCloud cloud = ToCloud(pts, n);
  const HybridGrid& g = *G(grid);
  const std::vector<Rigid3f> ts = m.GenerateExhaustiveSearchTransforms(g.resolution(), cloud);
  const Rigid3f init = ToRigid(init7).cast<float>();
  for (int64_t i = 0; i < k; ++i) {
    const Rigid3f cand = init * ts[indices[i]];
    const PointCloud moved = TransformPointCloud(cloud, cand);
    if (value_sums != nullptr) {
      uint64_t s = 0;
      for (const Vec3f& p : moved) {
        const uint16 v = g.value(g.GetCellIndex(p)) & 0x7fff;
        s += v == 0 ? 1 : v;
      }
      value_sums[i] = s;
    }
    if (scores != nullptr) scores[i] = m.ScoreCandidate(g, moved, ts[indices[i]]);
  }
}
// Cell indices of a cloud under a float pose (the bit-exactness probe).
void orc_transform_cell_indices(const float* pose7, const float* pts, int n, float resolution,
                                int* out) {
  const Rigid3f t(Vec3f(pose7[0], pose7[1], pose7[2]),
                  Quatf(pose7[3], pose7[4], pose7[5], pose7[6]));
  HybridGrid tmp(resolution);
  for (int i = 0; i < n; ++i) {
    const Vec3i c = tmp.GetCellIndex(t * Vec3f(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
    out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
  }
}

// ---------------------------------------------------------------- CSM3D
double orc_interpolated_probability(void* grid, double x, double y, double z) {
  const InterpolatedGrid ig(*G(grid));
  return ig.GetProbability(x, y, z);
}
// One OccupiedSpaceCostFunction3D block: residuals[n], jac_t[n*3], jac_q[n*4]
// (ambient, as ceres::AutoDiffCostFunction fills them); jac pointers may be null.
void orc_occupied_space_evaluate(void* grid, const float* pts, int n, double scaling,
                                 const double* t3, const double* q4, double* residuals,
                                 double* jac_t, double* jac_q) {
  const PointCloud cloud = ToCloud(pts, n);
  auto f = std::make_shared<OccupiedSpaceCostFunction3D>(scaling, cloud, *G(grid));
  ceres_like::ResidualBlock b = ceres_like::MakeAutoDiffBlock34(f, n);
  b.evaluate(t3, q4, residuals, jac_t, jac_q);
}
double orc_rotation_delta_squared_cost(const double* q4, double scaling, const double* target4) {
  const RotationDeltaCostFunctor3D f(scaling, Quatd(target4[0], target4[1], target4[2], target4[3]));
  double r[3];
  f(q4, r);
  return r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
}
// summary_out[10] = initial_cost, final_cost, num_successful, num_unsuccessful,
//   num_iterations, num_residual_evals, num_jacobian_evals, termination_type, 0, 0
void orc_csm3d_match(const double* occupied_space_weights, int k, double translation_weight,
                     double rotation_weight, int only_optimize_yaw, int use_nonmonotonic_steps,
                     int max_num_iterations, const double* target3, const double* init7,
                     const float* const* pts, const int* n, void* const* grids, double* out7,
                     double* summary_out) {
  CeresScanMatcherOptions3D o;
  o.occupied_space_weight.assign(occupied_space_weights, occupied_space_weights + k);
  o.translation_weight = translation_weight;
  o.rotation_weight = rotation_weight;
  o.only_optimize_yaw = only_optimize_yaw != 0;
  o.use_nonmonotonic_steps = use_nonmonotonic_steps != 0;
  o.max_num_iterations = max_num_iterations;
  std::vector<PointCloud> clouds;
  for (int i = 0; i < k; ++i) clouds.push_back(ToCloud(pts[i], n[i]));
  std::vector<CeresScanMatcher3D::PointCloudAndHybridGridPointers> pairs;
  for (int i = 0; i < k; ++i) pairs.emplace_back(&clouds[i], G(grids[i]));
  const CeresScanMatcher3D matcher(o);
  Rigid3d pose;
  ceres_like::Summary s;
  matcher.Match(Vec3d(target3[0], target3[1], target3[2]), ToRigid(init7), pairs, &pose, &s);
  FromRigid(pose, out7);
  if (summary_out != nullptr) {
    summary_out[0] = s.initial_cost;
    summary_out[1] = s.final_cost;
    summary_out[2] = s.num_successful_steps;
    summary_out[3] = s.num_unsuccessful_steps;
    summary_out[4] = s.num_iterations;
    summary_out[5] = s.num_residual_evaluations;
    summary_out[6] = s.num_jacobian_evaluations;
    summary_out[7] = s.termination_type;
    summary_out[8] = 0;
    summary_out[9] = 0;
  }
}

// ---------------------------------------------------------------- front end
// opts (doubles): [0..2] hi adaptive filter (max_length, min_num_points, max_range), [3..5] lo,
// [6] use_online_csm, [7..10] rtcsm options, [11] occupied_space_weight_0, [12] weight_1,
// [13] translation_weight, [14] rotation_weight, [15] only_optimize_yaw, [16] nonmonotonic,
// [17] max_num_iterations, [18..20] motion filter (time, distance, angle),
// [21] high_resolution, [22] high_resolution_max_range, [23] low_resolution, [24] num_range_data,
// [25] hit_probability, [26] miss_probability, [27] num_free_space_voxels
void* orc_front_end_new(const double* o) {
  FrontEndOptions f;
  f.high_resolution_adaptive_voxel_filter = {static_cast<float>(o[0]), static_cast<float>(o[1]), static_cast<float>(o[2])};
  f.low_resolution_adaptive_voxel_filter = {static_cast<float>(o[3]), static_cast<float>(o[4]), static_cast<float>(o[5])};
  f.use_online_correlative_scan_matching = o[6] != 0;
  f.real_time_correlative_scan_matcher = {o[7], o[8], o[9], o[10]};
  f.ceres_scan_matcher.occupied_space_weight = {o[11], o[12]};
  f.ceres_scan_matcher.translation_weight = o[13];
  f.ceres_scan_matcher.rotation_weight = o[14];
  f.ceres_scan_matcher.only_optimize_yaw = o[15] != 0;
  f.ceres_scan_matcher.use_nonmonotonic_steps = o[16] != 0;
  f.ceres_scan_matcher.max_num_iterations = static_cast<int>(o[17]);
  f.motion_filter = {o[18], o[19], o[20]};
  f.submaps = {o[21], o[22], o[23], static_cast<int>(o[24]), o[25], o[26], static_cast<int>(o[27])};
  return new FrontEnd(f);
}
void orc_front_end_free(void* fe) { delete static_cast<FrontEnd*>(fe); }
void orc_front_end_set_threads(void* fe, int threads) { static_cast<FrontEnd*>(fe)->rtcsm_threads = threads; }
// out[0] dropped, out[1..7] pose_estimate, out[8..14] observation, out[15..21] initial ceres pose,
// out[22] rtcsm score, out[23] final cost, out[24] iterations, out[25] n_hi, out[26] n_lo
void orc_front_end_match(void* fe, const double* pose_prediction7, const float* origin3, const float* returns,
                         int n, double* out) {
  RangeData rd{Vec3f(origin3[0], origin3[1], origin3[2]), ToCloud(returns, n), {}};
  const MatchResult r = static_cast<FrontEnd*>(fe)->Match(ToRigid(pose_prediction7), rd);
  out[0] = r.dropped ? 1 : 0;
  FromRigid(r.pose_estimate, out + 1);
  FromRigid(r.pose_observation_in_submap, out + 8);
  FromRigid(r.initial_ceres_pose, out + 15);
  out[22] = r.rtcsm_score;
  out[23] = r.summary.final_cost;
  out[24] = r.summary.num_iterations;
  out[25] = static_cast<double>(r.num_high);
  out[26] = static_cast<double>(r.num_low);
}
int orc_front_end_insert(void* fe, int64_t time_ticks, const double* pose7, const double* gravity4) {
  return static_cast<FrontEnd*>(fe)->Insert(time_ticks, ToRigid(pose7),
                                            Quatd(gravity4[0], gravity4[1], gravity4[2], gravity4[3]));
}
int orc_front_end_num_active_submaps(void* fe) {
  return static_cast<int>(static_cast<FrontEnd*>(fe)->active_submaps().submaps().size());
}
int orc_front_end_matching_index(void* fe) { return static_cast<FrontEnd*>(fe)->active_submaps().matching_index(); }
// Borrowed pointers to the active submap's grids (owned by the front end).
void orc_front_end_active_submap(void* fe, int i, double* local_pose7, int* num_range_data, void** hi, void** lo) {
  const auto& s = static_cast<FrontEnd*>(fe)->active_submaps().submaps()[i];
  FromRigid(s->local_pose(), local_pose7);
  *num_range_data = s->num_range_data();
  *hi = &s->high_resolution_hybrid_grid();
  *lo = &s->low_resolution_hybrid_grid();
}

// ---------------------------------------------------------------- AddRangeData pre-processing
// opts = [scan_period, min_range, max_range, voxel_filter_size].  ranges: n x (x,y,z,t).
// Outputs: hits_in_local (capacity n x 3), kind (capacity n), counts[0] = hits after the first
// voxel filter, counts[1] = returns in tracking, counts[2] = misses in tracking;
// returns_tracking / misses_tracking (capacity n x 3 each), current_pose7 (float), origin3 (tracking).
void orc_deskew_and_filter(const double* opts, const double* prev7, const double* cur7, const float* ranges, int n,
                           const float* origin3, float* hits_in_local, int* kind, int* counts,
                           float* returns_tracking, float* misses_tracking, float* current_pose7,
                           float* origin_tracking3) {
  std::vector<TimedPoint> r(n);
  for (int i = 0; i < n; ++i) r[i] = TimedPoint{ranges[4 * i], ranges[4 * i + 1], ranges[4 * i + 2], ranges[4 * i + 3]};
  const DeskewOptions o{opts[0], static_cast<float>(opts[1]), static_cast<float>(opts[2]), static_cast<float>(opts[3])};
  const DeskewResult d = DeskewAndFilter(o, ToRigid(prev7), ToRigid(cur7), r, Vec3f(origin3[0], origin3[1], origin3[2]));
  counts[0] = static_cast<int>(d.hits_in_local.size());
  for (size_t i = 0; i < d.hits_in_local.size(); ++i) {
    hits_in_local[3 * i] = d.hits_in_local[i].x; hits_in_local[3 * i + 1] = d.hits_in_local[i].y;
    hits_in_local[3 * i + 2] = d.hits_in_local[i].z;
    kind[i] = d.kind[i];
  }
  counts[1] = static_cast<int>(d.filtered_in_tracking.returns.size());
  counts[2] = static_cast<int>(d.filtered_in_tracking.misses.size());
  for (size_t i = 0; i < d.filtered_in_tracking.returns.size(); ++i) {
    const Vec3f& p = d.filtered_in_tracking.returns[i];
    returns_tracking[3 * i] = p.x; returns_tracking[3 * i + 1] = p.y; returns_tracking[3 * i + 2] = p.z;
  }
  for (size_t i = 0; i < d.filtered_in_tracking.misses.size(); ++i) {
    const Vec3f& p = d.filtered_in_tracking.misses[i];
    misses_tracking[3 * i] = p.x; misses_tracking[3 * i + 1] = p.y; misses_tracking[3 * i + 2] = p.z;
  }
  FromRigidF(d.current_pose, current_pose7);
  origin_tracking3[0] = d.filtered_in_tracking.origin.x;
  origin_tracking3[1] = d.filtered_in_tracking.origin.y;
  origin_tracking3[2] = d.filtered_in_tracking.origin.z;
}

// RangeDataAccumulator: AddRangeData with an origin table and num_accumulated_range_data > 1.
void* orc_accumulator_new() { return new RangeDataAccumulator; }
void orc_accumulator_free(void* a) { delete static_cast<RangeDataAccumulator*>(a); }
void orc_accumulator_add(void* a, const double* opts, const double* prev7, const double* cur7, const float* ranges, int n,
                         const int* origin_index, const float* origins, int num_origins, float* current_pose7) {
  std::vector<TimedPoint> r(n);
  for (int i = 0; i < n; ++i) r[i] = TimedPoint{ranges[4 * i], ranges[4 * i + 1], ranges[4 * i + 2], ranges[4 * i + 3]};
  std::vector<int> oi;
  if (origin_index != nullptr) oi.assign(origin_index, origin_index + n);
  std::vector<Vec3f> og;
  for (int k = 0; k < num_origins; ++k) og.emplace_back(origins[3 * k], origins[3 * k + 1], origins[3 * k + 2]);
  const DeskewOptions o{opts[0], static_cast<float>(opts[1]), static_cast<float>(opts[2]), static_cast<float>(opts[3])};
  FromRigidF(static_cast<RangeDataAccumulator*>(a)->Add(o, ToRigid(prev7), ToRigid(cur7), r, oi, og), current_pose7);
}
int orc_accumulator_finish(void* a, const double* opts, float* returns_tracking, int capacity, float* origin_tracking3) {
  const DeskewOptions o{opts[0], static_cast<float>(opts[1]), static_cast<float>(opts[2]), static_cast<float>(opts[3])};
  const RangeData d = static_cast<RangeDataAccumulator*>(a)->Finish(o);
  const int n = static_cast<int>(d.returns.size());
  for (int i = 0; i < n && i < capacity; ++i) {
    returns_tracking[3 * i] = d.returns[i].x; returns_tracking[3 * i + 1] = d.returns[i].y; returns_tracking[3 * i + 2] = d.returns[i].z;
  }
  origin_tracking3[0] = d.origin.x; origin_tracking3[1] = d.origin.y; origin_tracking3[2] = d.origin.z;
  return n;
}

// ---------------------------------------------------------------- 2D (config 1, CPU only)
void* orc_pg_new(double resolution, double max_x, double max_y, int num_x_cells, int num_y_cells) {
  return new ProbabilityGrid(MapLimits{resolution, max_x, max_y, num_x_cells, num_y_cells});
}
void orc_pg_free(void* g) { delete static_cast<ProbabilityGrid*>(g); }
void orc_pg_set_probability(void* g, int x, int y, float p) {
  static_cast<ProbabilityGrid*>(g)->SetProbability(x, y, p);
}
float orc_pg_probability(void* g, int x, int y) {
  return static_cast<ProbabilityGrid*>(g)->GetProbability(x, y);
}
void orc_pg_cell_index(void* g, float px, float py, int* out2) {
  const Cell2 c = static_cast<ProbabilityGrid*>(g)->limits().GetCellIndex(px, py);
  out2[0] = c.x; out2[1] = c.y;
}
void orc_pg_limits(void* g, double* out5) {
  const MapLimits& l = static_cast<ProbabilityGrid*>(g)->limits();
  out5[0] = l.resolution; out5[1] = l.max_x; out5[2] = l.max_y; out5[3] = l.num_x_cells; out5[4] = l.num_y_cells;
}
void orc_pg_cells(void* g, uint16_t* out) {
  const std::vector<uint16>& c = static_cast<ProbabilityGrid*>(g)->cells();
  std::memcpy(out, c.data(), c.size() * sizeof(uint16_t));
}
void orc_pg_insert(void* g, const float* origin3, const float* returns, int n,
                   double hit_probability, double miss_probability, int insert_free_space) {
  InsertRangeData2D(static_cast<ProbabilityGrid*>(g), origin3[0], origin3[1], ToCloud(returns, n),
                    static_cast<float>(hit_probability), static_cast<float>(miss_probability),
                    insert_free_space != 0);
}
// init3 = [x, y, theta]; out3 likewise.  Returns the best score.
double orc_rtcsm2d_match(const double* opts, const double* init3, const float* pts, int n, void* g,
                         double* out3) {
  const RealTimeCorrelativeScanMatcher2D m(
      RealTimeCorrelativeScanMatcherOptions{opts[0], opts[1], opts[2], opts[3]});
  return m.Match(init3, ToCloud(pts, n), *static_cast<ProbabilityGrid*>(g), out3);
}
// Scores one candidate (scan_index 0 of an unrotated scan, integer offsets) --
// what the reference's own 2D test exercises.
float orc_rtcsm2d_score_single(const double* opts, const float* pts, int n, void* g,
                               int x_index_offset, int y_index_offset) {
  const RealTimeCorrelativeScanMatcher2D m(
      RealTimeCorrelativeScanMatcherOptions{opts[0], opts[1], opts[2], opts[3]});
  return m.ScoreSingle(ToCloud(pts, n), *static_cast<ProbabilityGrid*>(g), x_index_offset,
                       y_index_offset);
}

// ---------------------------------------------------------------- timing helper
// ---------------------------------------------------------------- fast correlative scan matcher 3D
// options: [branch_and_bound_depth, full_resolution_depth, min_rotational_score,
//           min_low_resolution_score, linear_xy_search_window, linear_z_search_window,
//           angular_search_window]
static FastCorrelativeScanMatcherOptions3D FastOptions(const double* o) {
  return FastCorrelativeScanMatcherOptions3D{static_cast<int>(o[0]), static_cast<int>(o[1]), o[2], o[3], o[4], o[5], o[6]};
}
static std::vector<std::pair<Histogram, float>> HistogramsAtAngles(const float* histograms, const float* angles,
                                                                 int num_nodes, int histogram_size) {
  std::vector<std::pair<Histogram, float>> v;
  for (int i = 0; i < num_nodes; ++i)
    v.emplace_back(Histogram(histograms + static_cast<size_t>(i) * histogram_size,
                             histograms + static_cast<size_t>(i + 1) * histogram_size),
                   angles[i]);
  return v;
}
static NodeData MakeNodeData(const double* gravity4, const float* hi, int n_hi, const float* lo, int n_lo,
                             const float* histogram, int histogram_size) {
  NodeData d;
  d.gravity_alignment = Quatd(gravity4[0], gravity4[1], gravity4[2], gravity4[3]);
  d.high_resolution_point_cloud = ToCloud(hi, n_hi);
  d.low_resolution_point_cloud = ToCloud(lo, n_lo);
  d.rotational_scan_matcher_histogram.assign(histogram, histogram + histogram_size);
  return d;
}
// out9: found, score, rotational_score, low_resolution_score, num_scored_candidates, num_discrete_scans
static void StoreFastResult(const FastMatchResult& r, double* pose7, double* out6) {
  out6[0] = r.found ? 1. : 0.;
  out6[1] = r.score;
  out6[2] = r.rotational_score;
  out6[3] = r.low_resolution_score;
  out6[4] = static_cast<double>(r.num_scored_candidates);
  out6[5] = r.num_discrete_scans;
  if (r.found) FromRigid(r.pose_estimate, pose7);
}

void* orc_fast_csm_new(void* hi_grid, void* lo_grid, const float* histograms, const float* angles, int num_nodes,
                       int histogram_size, const double* options7) {
  return new FastCorrelativeScanMatcher3D(*G(hi_grid), G(lo_grid),
                                          HistogramsAtAngles(histograms, angles, num_nodes, histogram_size),
                                          FastOptions(options7));
}
void orc_fast_csm_free(void* m) { delete static_cast<FastCorrelativeScanMatcher3D*>(m); }
int orc_fast_csm_max_depth(void* m) { return static_cast<FastCorrelativeScanMatcher3D*>(m)->stack().max_depth(); }
int64_t orc_fast_csm_stack_num_cells(void* m, int depth) {
  int64_t n = 0;
  static_cast<FastCorrelativeScanMatcher3D*>(m)->stack().Get(depth).ForEachCell([&](const Vec3i&, uint8) { ++n; });
  return n;
}
void orc_fast_csm_stack_cells(void* m, int depth, int* xyz, uint8_t* values) {
  int64_t i = 0;
  static_cast<FastCorrelativeScanMatcher3D*>(m)->stack().Get(depth).ForEachCell([&](const Vec3i& c, uint8 v) {
    xyz[3 * i] = c.x; xyz[3 * i + 1] = c.y; xyz[3 * i + 2] = c.z;
    values[i] = v;
    ++i;
  });
}
void orc_fast_csm_match(void* m, const double* node_pose7, const double* submap_pose7, const double* gravity4,
                        const float* hi, int n_hi, const float* lo, int n_lo, const float* histogram, int histogram_size,
                        float min_score, double* pose7, double* out6) {
  const NodeData d = MakeNodeData(gravity4, hi, n_hi, lo, n_lo, histogram, histogram_size);
  StoreFastResult(static_cast<FastCorrelativeScanMatcher3D*>(m)->Match(ToRigid(node_pose7), ToRigid(submap_pose7), d,
                                                                         min_score), pose7, out6);
}
void orc_fast_csm_match_full_submap(void* m, const double* node_rotation4, const double* submap_rotation4,
                                    const double* gravity4, const float* hi, int n_hi, const float* lo, int n_lo,
                                    const float* histogram, int histogram_size, float min_score, double* pose7,
                                    double* out6) {
  const NodeData d = MakeNodeData(gravity4, hi, n_hi, lo, n_lo, histogram, histogram_size);
  StoreFastResult(static_cast<FastCorrelativeScanMatcher3D*>(m)->MatchFullSubmap(
                      Quatd(node_rotation4[0], node_rotation4[1], node_rotation4[2], node_rotation4[3]),
                      Quatd(submap_rotation4[0], submap_rotation4[1], submap_rotation4[2], submap_rotation4[3]), d,
                      min_score), pose7, out6);
}
void orc_fast_csm_match_3dof(void* m, const double* pose_in_submap7, const double* gravity4, const float* hi, int n_hi,
                             const float* lo, int n_lo, const float* histogram, int histogram_size, float min_score,
                             double* pose7, double* out6) {
  const NodeData d = MakeNodeData(gravity4, hi, n_hi, lo, n_lo, histogram, histogram_size);
  StoreFastResult(static_cast<FastCorrelativeScanMatcher3D*>(m)->MatchWith3DofInitial(ToRigid(pose_in_submap7), d, min_score),
                  pose7, out6);
}
void orc_compute_histogram(const float* pts, int n, int histogram_size, float* out) {
  const Histogram h = ComputeHistogram(ToCloud(pts, n), histogram_size);
  std::memcpy(out, h.data(), sizeof(float) * histogram_size);
}
// ComputeHistogram's additions one by one: (bucket, value) in the order `histogram(bucket) += value` happens; returns how
// many there were (the first `capacity` are stored).
int orc_histogram_contributions(const float* pts, int n, int histogram_size, int* buckets, float* values, int capacity) {
  std::vector<std::pair<int, float>> trace;
  rsm_detail::ContributionTrace() = &trace;
  (void)ComputeHistogram(ToCloud(pts, n), histogram_size);
  rsm_detail::ContributionTrace() = nullptr;
  const int total = static_cast<int>(trace.size());
  for (int i = 0; i < total && i < capacity; ++i) {
    buckets[i] = trace[static_cast<size_t>(i)].first;
    values[i] = trace[static_cast<size_t>(i)].second;
  }
  return total;
}
// The order this machine's std::sort leaves (key, index) pairs in when compared by key only -- SortSlice's comparison
// (rotational_scan_matcher.cc:97-121); what the device's restatement of introsort is checked against.
void orc_std_sort_order(const float* keys, int n, int* order) {
  struct Pair {
    float key;
    int id;
    bool operator<(const Pair& o) const { return key < o.key; }
  };
  std::vector<Pair> v(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) v[static_cast<size_t>(i)] = Pair{keys[i], i};
  std::sort(v.begin(), v.end());
  for (int i = 0; i < n; ++i) order[i] = v[static_cast<size_t>(i)].id;
}
// MotionFilter (mapping/internal/motion_filter.cc:40-58) as an object, for motion_filter_test.cc.
void* orc_motion_filter_create(double max_time_seconds, double max_distance_meters, double max_angle_radians) {
  return new MotionFilter(MotionFilterOptions{max_time_seconds, max_distance_meters, max_angle_radians});
}
void orc_motion_filter_destroy(void* f) { delete static_cast<MotionFilter*>(f); }
int orc_motion_filter_is_similar(void* f, int64_t time_ticks, const double* pose7) {
  return static_cast<MotionFilter*>(f)->IsSimilar(time_ticks, ToRigid(pose7)) ? 1 : 0;
}

void orc_rotational_match(const float* histograms, const float* node_angles, int num_nodes, int histogram_size,
                          const float* scan_histogram, float initial_angle, const float* angles, int num_angles,
                          float* scores) {
  const RotationalScanMatcher rsm(HistogramsAtAngles(histograms, node_angles, num_nodes, histogram_size));
  const std::vector<float> s = rsm.Match(Histogram(scan_histogram, scan_histogram + histogram_size), initial_angle,
                                         std::vector<float>(angles, angles + num_angles));
  std::memcpy(scores, s.data(), sizeof(float) * num_angles);
}

// precomputation_grid_3d_test.cc:30-77 run natively (it interleaves std::mt19937 draws of two
// distributions): returns the largest |naive max - precomputed| over the 4 x 100 probes.
double orc_kat_precomputation_grid(void) {
  HybridGrid hybrid_grid(2.f);
  std::mt19937 rng(23847);
  std::uniform_int_distribution<int> coordinate_distribution(-50, 49);
  std::uniform_real_distribution<float> value_distribution(kMinProbability, kMaxProbability);
  for (int i = 0; i < 1000; ++i) {
    const auto x = coordinate_distribution(rng);
    const auto y = coordinate_distribution(rng);
    const auto z = coordinate_distribution(rng);
    hybrid_grid.SetProbability(Vec3i(x, y, z), value_distribution(rng));
  }
  std::vector<PrecomputationGrid3D> grids;
  double worst = 0.;
  for (int depth = 0; depth <= 3; ++depth) {
    if (depth == 0) {
      grids.push_back(ConvertToPrecomputationGrid(hybrid_grid));
    } else {
      const int s = 1 << (depth - 1);
      grids.push_back(PrecomputeGrid(grids.back(), false, Vec3i(s, s, s)));
    }
    const int width = 1 << depth;
    for (int i = 0; i < 100; ++i) {
      const auto x = coordinate_distribution(rng);
      const auto y = coordinate_distribution(rng);
      const auto z = coordinate_distribution(rng);
      float max_probability = 0.;
      for (int dx = 0; dx < width; ++dx)
        for (int dy = 0; dy < width; ++dy)
          for (int dz = 0; dz < width; ++dz)
            max_probability = std::max(max_probability, hybrid_grid.GetProbability(Vec3i(x + dx, y + dy, z + dz)));
      worst = std::max(worst, static_cast<double>(std::abs(
                                  max_probability - PrecomputationGrid3D::ToProbability(grids.back().value(Vec3i(x, y, z))))));
    }
  }
  return worst;
}

// transform/transform_test.cc:29-46 (TransformTest.GetAngle) run natively: 100 random (angle, axis) pairs from
// std::mt19937(42); returns the largest |angle - GetAngle(Rotation(AngleAxisVectorToRotationQuaternion(angle * axis)))|
// (the reference expects <= 1e-6).  Pins the two functions the RTCSM3D candidate generation is built from
// (real_time_correlative_scan_matcher_3d.cc:58-92 -> transform/transform.h:33-37,85-99).
double orc_kat_transform_get_angle(void) {
  std::mt19937 rng(42);
  std::uniform_real_distribution<float> angle_distribution(0.f, static_cast<float>(M_PI));
  std::uniform_real_distribution<float> position_distribution(-1.f, 1.f);
  double worst = 0.;
  for (int i = 0; i != 100; ++i) {
    const float angle = angle_distribution(rng);
    const float x = position_distribution(rng);
    const float y = position_distribution(rng);
    const float z = position_distribution(rng);
    const float n = std::sqrt(x * x + y * y + z * z);  // Eigen normalized(): v / sqrt(squaredNorm)
    const Vec3f axis(x / n, y / n, z / n);
    const Vec3f angle_axis(angle * axis.x, angle * axis.y, angle * axis.z);
    const float got = GetAngle(Rigid3f::Rotation(AngleAxisVectorToRotationQuaternion(angle_axis)));
    worst = std::max(worst, static_cast<double>(std::abs(angle - got)));
  }
  return worst;
}

// transform/rigid_transform_test.cc:33-94 (RigidTransformTest Identity3DTest / Inverse3DTest, float and double) run
// natively with the fixture's std::mt19937(42) draws.  The reference compares 4 x 4 matrices with Eigen's
// isApprox(eps): ||a - b||_F <= eps * min(||a||_F, ||b||_F), eps = numeric_limits<T>::epsilon().  Returns the largest
// ||a - b||_F / (eps * min(||a||_F, ||b||_F)) over the six expectations of the two tests (<= 1 passes).
double orc_kat_rigid_transform(int use_float) { return use_float ? RigidTransformKat<float>() : RigidTransformKat<double>(); }

// fast_correlative_scan_matcher_3d_test.cc:35-190.  mode 0: CorrectPoseForMatch (20 random poses),
// mode 1: CorrectPoseForMatchFullSubmap (1 pose).  Returns the number of failed expectations;
// worst3 = {max translation error, max rotation angle error, min score}.
int orc_kat_fast_csm(int mode, double* worst3) {
  const PointCloud point_cloud = {Vec3f(4.f, 0.f, 0.f), Vec3f(4.5f, 0.f, 0.f), Vec3f(5.f, 0.f, 0.f), Vec3f(5.5f, 0.f, 0.f),
                                  Vec3f(0.f, 4.f, 0.f), Vec3f(0.f, 4.5f, 0.f), Vec3f(0.f, 5.f, 0.f), Vec3f(0.f, 5.5f, 0.f),
                                  Vec3f(0.f, 0.f, 4.f), Vec3f(0.f, 0.f, 4.5f), Vec3f(0.f, 0.f, 5.f), Vec3f(0.f, 0.f, 5.5f)};
  std::mt19937 prng(42);
  std::uniform_real_distribution<float> distribution(-1.f, 1.f);
  const FastCorrelativeScanMatcherOptions3D options{6, 6, 0.1, 0.15, 0.8, 0.8, 0.3};
  const RangeDataInserter3D inserter(0.7, 0.4, 5);
  const float kMinScore = 0.1f;
  int failures = 0;
  worst3[0] = worst3[1] = 0.;
  worst3[2] = 1e9;
  const int rounds = mode == 0 ? 20 : 1;
  for (int i = 0; i != rounds; ++i) {
    const float x = 0.7f * distribution(prng);
    const float y = 0.7f * distribution(prng);
    const float z = 0.7f * distribution(prng);
    const float theta = 0.2f * distribution(prng);
    // Eigen::AngleAxisf(theta, UnitZ) -> Quaternionf: w = cos(theta/2), vec = sin(theta/2) * axis (float)
    const Rigid3f expected = Rigid3f::Translation(Vec3f(x, y, z)) *
                             Rigid3f::Rotation(Quatf(std::cos(0.5f * theta), 0.f, 0.f, std::sin(0.5f * theta)));
    HybridGrid grid(0.05f);
    PointCloud transformed;
    for (const Vec3f& p : point_cloud) transformed.push_back(expected * p);
    inserter.Insert(RangeData{expected.translation, transformed, {}}, &grid);
    grid.FinishUpdate();
    // HistogramsAtAnglesFromNodes (:114-127): one node at `expected`, identity gravity alignment
    const Rigid3d node_pose = expected.cast<double>() * Rigid3d::Rotation(QuatInverse(Quatd()));
    const std::vector<std::pair<Histogram, float>> nodes = {
        {Histogram(10, 0.f), static_cast<float>(GetYaw(node_pose.rotation))}};
    const FastCorrelativeScanMatcher3D matcher(grid, &grid, nodes, options);
    NodeData data;
    data.gravity_alignment = Quatd();
    data.high_resolution_point_cloud = point_cloud;
    data.low_resolution_point_cloud = point_cloud;
    data.rotational_scan_matcher_histogram = Histogram(10, 0.f);
    const FastMatchResult r = mode == 0 ? matcher.Match(Rigid3d(), Rigid3d(), data, kMinScore)
                                        : matcher.MatchFullSubmap(Quatd(), Quatd(), data, kMinScore);
    if (!r.found) {
      ++failures;
      continue;
    }
    if (!(kMinScore < r.score)) ++failures;
    if (!(0.09f < r.rotational_score)) ++failures;
    if (!(0.14f < r.low_resolution_score)) ++failures;
    // transform::IsNearly(expected, actual, 0.05f): |t| and rotation angle of the difference
    const Rigid3f actual = r.pose_estimate.cast<float>();
    const Rigid3f diff = expected.inverse() * actual;
    const double dt = diff.translation.norm();
    const double da = GetAngle(diff);
    worst3[0] = std::max(worst3[0], dt);
    worst3[1] = std::max(worst3[1], da);
    worst3[2] = std::min(worst3[2], static_cast<double>(r.score));
    if (!(dt < 0.05) || !(da < 0.05)) ++failures;
    NodeData far = data;
    far.low_resolution_point_cloud = {Vec3f(42.f, 42.f, 42.f)};
    const FastMatchResult r2 = mode == 0 ? matcher.Match(Rigid3d(), Rigid3d(), far, kMinScore)
                                         : matcher.MatchFullSubmap(Quatd(), Quatd(), far, kMinScore);
    if (r2.found) ++failures;
  }
  return failures;
}

// ---------------------------------------------------------------- IMU preintegration (integration_base.h)
void* orc_imu_new(const double* ba3, const double* bg3, const double* noise4) {
  return new IntegrationBase(Vec3d(ba3[0], ba3[1], ba3[2]), Vec3d(bg3[0], bg3[1], bg3[2]),
                             ImuNoise{noise4[0], noise4[1], noise4[2], noise4[3]});
}
void orc_imu_free(void* m) { delete static_cast<IntegrationBase*>(m); }
void orc_imu_push_back(void* m, double dt, const double* acc3, const double* gyr3) {
  static_cast<IntegrationBase*>(m)->push_back(dt, Vec3d(acc3[0], acc3[1], acc3[2]), Vec3d(gyr3[0], gyr3[1], gyr3[2]));
}
void orc_imu_repropagate(void* m, const double* ba3, const double* bg3) {
  static_cast<IntegrationBase*>(m)->repropagate(Vec3d(ba3[0], ba3[1], ba3[2]), Vec3d(bg3[0], bg3[1], bg3[2]));
}
// out: sum_dt, delta_p[3], delta_q[4] (w,x,y,z), delta_v[3], jacobian[225], covariance[225]
void orc_imu_get(void* m, double* out461) {
  const IntegrationBase& b = *static_cast<IntegrationBase*>(m);
  double* o = out461;
  *o++ = b.sum_dt;
  *o++ = b.delta_p.x; *o++ = b.delta_p.y; *o++ = b.delta_p.z;
  *o++ = b.delta_q.w; *o++ = b.delta_q.x; *o++ = b.delta_q.y; *o++ = b.delta_q.z;
  *o++ = b.delta_v.x; *o++ = b.delta_v.y; *o++ = b.delta_v.z;
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) *o++ = b.jacobian[i][j];
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) *o++ = b.covariance[i][j];
}
void orc_imu_evaluate(void* m, const double* si16, const double* sj16, const double* g3, double* residuals15) {
  auto v = [](const double* p) { return Vec3d(p[0], p[1], p[2]); };
  const std::array<double, 15> r = static_cast<IntegrationBase*>(m)->evaluate(
      v(si16), Quatd(si16[3], si16[4], si16[5], si16[6]), v(si16 + 7), v(si16 + 10), v(si16 + 13), v(sj16),
      Quatd(sj16[3], sj16[4], sj16[5], sj16[6]), v(sj16 + 7), v(sj16 + 10), v(sj16 + 13), v(g3));
  std::memcpy(residuals15, r.data(), sizeof(double) * 15);
}

double orc_now_seconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // extern "C"
