// ORACLE (test infrastructure, NOT product code).
//
// Config-1 ("plumbing, CPU only") pieces: the dense 2D probability grid and the
// 2D real-time correlative scan matcher.
//   mapping/2d/map_limits.h:37-88              MapLimits::GetCellIndex / Contains
//   mapping/2d/grid_2d.cc:42-51,76-101,168-171 Grid2D cells, FinishUpdate, ToFlatIndex
//   mapping/2d/grid_2d.cc:119-151              GrowLimits
//   mapping/2d/probability_grid.cc:27-73       SetProbability / ApplyLookupTable / GetProbability
//   mapping/2d/probability_grid_range_data_inserter_2d.cc:48-65   Insert
//   mapping/internal/2d/ray_casting.cc:23-215  subpixel supercover line + CastRays
//   mapping/internal/2d/scan_matching/correlative_scan_matcher_2d.cc:27-127
//   mapping/internal/2d/scan_matching/real_time_correlative_scan_matcher_2d.cc:40-135
#ifndef ORACLE_OM_GRID2D_H_
#define ORACLE_OM_GRID2D_H_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "om_rtcsm3d.h"

namespace oracle {

struct Cell2 {
  int x, y;
};

struct MapLimits {
  double resolution;
  double max_x, max_y;
  int num_x_cells, num_y_cells;

  // map_limits.h:69-76 -- note (y, x) swap: row = from max.y, column = from max.x.
  Cell2 GetCellIndex(float px, float py) const {
    return Cell2{RoundToInt((max_y - py) / resolution - 0.5),
                 RoundToInt((max_x - px) / resolution - 0.5)};
  }
  bool Contains(const Cell2& c) const {
    return 0 <= c.x && 0 <= c.y && c.x < num_x_cells && c.y < num_y_cells;
  }
};

class ProbabilityGrid {
 public:
  explicit ProbabilityGrid(const MapLimits& limits)
      : limits_(limits),
        cells_(static_cast<size_t>(limits.num_x_cells) * limits.num_y_cells,
               kUnknownCorrespondenceValue) {}

  const MapLimits& limits() const { return limits_; }
  const std::vector<uint16>& cells() const { return cells_; }

  int ToFlatIndex(const Cell2& c) const {
    if (!limits_.Contains(c)) std::abort();  // CHECK
    return limits_.num_x_cells * c.y + c.x;
  }
  void SetProbability(int x, int y, float probability) {
    uint16& cell = cells_[ToFlatIndex(Cell2{x, y})];
    if (cell != kUnknownProbabilityValue) std::abort();  // CHECK_EQ
    cell = CorrespondenceCostToValue(ProbabilityToCorrespondenceCost(probability));
  }
  bool ApplyLookupTable(const Cell2& c, const std::vector<uint16>& table) {
    const int flat = ToFlatIndex(c);
    uint16* cell = &cells_[flat];
    if (*cell >= kUpdateMarker) return false;
    update_indices_.push_back(flat);
    *cell = table[*cell];
    return true;
  }
  void FinishUpdate() {
    while (!update_indices_.empty()) {
      cells_[update_indices_.back()] -= kUpdateMarker;
      update_indices_.pop_back();
    }
  }
  // probability_grid.cc:69-73
  float GetProbability(int x, int y) const {
    const Cell2 c{x, y};
    if (!limits_.Contains(c)) return kMinProbability;
    return CorrespondenceCostToProbability(
        ValueToCorrespondenceCost(cells_[limits_.num_x_cells * y + x]));
  }
  // grid_2d.cc:119-151
  void GrowLimits(float px, float py) {
    if (!update_indices_.empty()) std::abort();
    while (!limits_.Contains(limits_.GetCellIndex(px, py))) {
      const int x_offset = limits_.num_x_cells / 2;
      const int y_offset = limits_.num_y_cells / 2;
      MapLimits grown{limits_.resolution,
                      limits_.max_x + limits_.resolution * y_offset,
                      limits_.max_y + limits_.resolution * x_offset,
                      2 * limits_.num_x_cells, 2 * limits_.num_y_cells};
      const int stride = grown.num_x_cells;
      const int offset = x_offset + stride * y_offset;
      std::vector<uint16> cells(static_cast<size_t>(grown.num_x_cells) * grown.num_y_cells,
                                kUnknownCorrespondenceValue);
      for (int i = 0; i < limits_.num_y_cells; ++i)
        for (int j = 0; j < limits_.num_x_cells; ++j)
          cells[offset + j + i * stride] = cells_[j + i * limits_.num_x_cells];
      cells_.swap(cells);
      limits_ = grown;
    }
  }

 private:
  MapLimits limits_;
  std::vector<uint16> cells_;
  std::vector<int> update_indices_;
};

namespace detail2d {

constexpr int kSubpixelScale = 1000;  // ray_casting.cc:24

// ray_casting.cc:29-162: marks every pixel touched by the segment begin->end,
// both given in 1/1000-pixel coordinates.
inline void CastRay(Cell2 begin, Cell2 end, const std::vector<uint16>& miss_table,
                    ProbabilityGrid* grid) {
  if (begin.x > end.x) std::swap(begin, end);
  if (begin.x < 0 || begin.y < 0 || end.y < 0) std::abort();  // CHECK_GE x3
  const int S = kSubpixelScale;
  if (begin.x / S == end.x / S) {  // vertical in full pixels
    Cell2 cur{begin.x / S, std::min(begin.y, end.y) / S};
    const int end_y = std::max(begin.y, end.y) / S;
    for (; cur.y <= end_y; ++cur.y) grid->ApplyLookupTable(cur, miss_table);
    return;
  }
  const int64_t dx = end.x - begin.x;
  const int64_t dy = end.y - begin.y;
  const int64_t denominator = 2 * S * dx;
  Cell2 cur{begin.x / S, begin.y / S};
  int64_t sub_y = (2 * (begin.y % S) + 1) * dx;
  const int first_pixel = 2 * S - 2 * (begin.x % S) - 1;
  const int last_pixel = 2 * (end.x % S) + 1;
  const int end_x = std::max(begin.x, end.x) / S;
  sub_y += dy * first_pixel;
  const bool up = dy > 0;
  // One body for both slopes: `over` = crossed the far pixel edge in y.
  auto over = [&]() { return up ? sub_y > denominator : sub_y < 0; };
  auto on_edge = [&]() { return up ? sub_y == denominator : sub_y == 0; };
  auto step_y = [&]() {
    if (up) { sub_y -= denominator; ++cur.y; } else { sub_y += denominator; --cur.y; }
  };
  for (;;) {
    grid->ApplyLookupTable(cur, miss_table);
    while (over()) {
      step_y();
      grid->ApplyLookupTable(cur, miss_table);
    }
    ++cur.x;
    if (on_edge()) step_y();
    if (cur.x == end_x) break;
    sub_y += dy * 2 * S;
  }
  sub_y += dy * last_pixel;
  grid->ApplyLookupTable(cur, miss_table);
  while (over()) {
    step_y();
    grid->ApplyLookupTable(cur, miss_table);
  }
}

}  // namespace detail2d

// ray_casting.cc:164-215 + probability_grid_range_data_inserter_2d.cc:48-65
// (returns only; the path never passes 2D misses).
inline void InsertRangeData2D(ProbabilityGrid* grid, float origin_x, float origin_y,
                              const PointCloud& returns, float hit_probability,
                              float miss_probability, bool insert_free_space) {
  const std::vector<uint16> hit_table =
      ComputeLookupTableToApplyCorrespondenceCostOdds(Odds(hit_probability));
  const std::vector<uint16> miss_table =
      ComputeLookupTableToApplyCorrespondenceCostOdds(Odds(miss_probability));
  // GrowAsNeeded
  float min_x = origin_x, max_x = origin_x, min_y = origin_y, max_y = origin_y;
  for (const Vec3f& h : returns) {
    min_x = std::min(min_x, h.x); max_x = std::max(max_x, h.x);
    min_y = std::min(min_y, h.y); max_y = std::max(max_y, h.y);
  }
  constexpr float kPadding = 1e-6f;
  grid->GrowLimits(min_x - kPadding * 1.f, min_y - kPadding * 1.f);
  grid->GrowLimits(max_x + kPadding * 1.f, max_y + kPadding * 1.f);

  const MapLimits& limits = grid->limits();
  const int S = detail2d::kSubpixelScale;
  const MapLimits super{limits.resolution / S, limits.max_x, limits.max_y,
                        limits.num_x_cells * S, limits.num_y_cells * S};
  const Cell2 begin = super.GetCellIndex(origin_x, origin_y);
  std::vector<Cell2> ends;
  ends.reserve(returns.size());
  for (const Vec3f& h : returns) {
    ends.push_back(super.GetCellIndex(h.x, h.y));
    grid->ApplyLookupTable(Cell2{ends.back().x / S, ends.back().y / S}, hit_table);
  }
  if (insert_free_space) {
    for (const Cell2& end : ends) detail2d::CastRay(begin, end, miss_table, grid);
  }
  grid->FinishUpdate();
}

// 2D real-time correlative scan matcher.
class RealTimeCorrelativeScanMatcher2D {
 public:
  explicit RealTimeCorrelativeScanMatcher2D(const RealTimeCorrelativeScanMatcherOptions& o)
      : options_(o) {}

  struct SearchParameters {  // correlative_scan_matcher_2d.cc:27-55
    int num_angular_perturbations;
    double angular_perturbation_step_size;
    double resolution;
    int num_scans;
    int num_linear_perturbations;
  };

  static SearchParameters MakeSearchParameters(double linear_window, double angular_window,
                                               const PointCloud& cloud, double resolution) {
    SearchParameters sp;
    sp.resolution = resolution;
    float max_scan_range = 3.f * resolution;  // double product narrowed to float
    for (const Vec3f& p : cloud) {
      const float range = std::sqrt(p.x * p.x + p.y * p.y);  // head<2>().norm()
      max_scan_range = std::max(range, max_scan_range);
    }
    const double kSafetyMargin = 1. - 1e-3;
    sp.angular_perturbation_step_size =
        kSafetyMargin * std::acos(1. - Pow2(resolution) / (2. * Pow2(max_scan_range)));
    sp.num_angular_perturbations =
        static_cast<int>(std::ceil(angular_window / sp.angular_perturbation_step_size));
    sp.num_scans = 2 * sp.num_angular_perturbations + 1;
    sp.num_linear_perturbations = static_cast<int>(std::ceil(linear_window / resolution));
    return sp;
  }

  // Rotation about z by `angle` as Eigen::AngleAxisf -> Quaternionf.
  static Rigid3f RotationZ(float angle) {
    return Rigid3f::Rotation(Quatf(std::cos(0.5f * angle), 0.f, 0.f, std::sin(0.5f * angle)));
  }

  // real_time_correlative_scan_matcher_2d.cc:74-108.  pose = [x, y, theta].
  double Match(const double* init3, const PointCloud& cloud, const ProbabilityGrid& grid,
               double* out3) const {
    const double initial_rotation = init3[2];
    const PointCloud rotated =
        TransformPointCloud(cloud, RotationZ(static_cast<float>(initial_rotation)));
    const SearchParameters sp =
        MakeSearchParameters(options_.linear_search_window, options_.angular_search_window,
                             rotated, grid.limits().resolution);
    // GenerateRotatedScans (:93-109) + DiscretizeScans (:111-127)
    std::vector<std::vector<Cell2>> discrete(sp.num_scans);
    double delta_theta = -sp.num_angular_perturbations * sp.angular_perturbation_step_size;
    const float tx = static_cast<float>(init3[0]), ty = static_cast<float>(init3[1]);
    for (int s = 0; s < sp.num_scans; ++s, delta_theta += sp.angular_perturbation_step_size) {
      const PointCloud scan = TransformPointCloud(rotated, RotationZ(static_cast<float>(delta_theta)));
      discrete[s].reserve(scan.size());
      for (const Vec3f& p : scan)
        discrete[s].push_back(grid.limits().GetCellIndex(p.x + tx, p.y + ty));
    }
    // GenerateExhaustiveSearchCandidates + ScoreCandidates + max_element (first max)
    float best_score = -1.f;
    double best_x = 0, best_y = 0, best_o = 0;
    bool have = false;
    const int L = sp.num_linear_perturbations;
    for (int s = 0; s < sp.num_scans; ++s)
      for (int xo = -L; xo <= L; ++xo)
        for (int yo = -L; yo <= L; ++yo) {
          const double cx = -yo * sp.resolution;
          const double cy = -xo * sp.resolution;
          const double orientation =
              (s - sp.num_angular_perturbations) * sp.angular_perturbation_step_size;
          const float score = Score(grid, discrete[s], xo, yo, cx, cy, orientation);
          if (!(score > 0.f)) std::abort();  // CHECK_GT
          if (!have || best_score < score) {  // std::max_element semantics
            have = true;
            best_score = score;
            best_x = cx; best_y = cy; best_o = orientation;
          }
        }
    out3[0] = init3[0] + best_x;
    out3[1] = init3[1] + best_y;
    out3[2] = initial_rotation + best_o;  // Rotation2Dd product = angle sum
    return best_score;
  }

  // The reference test's single-candidate scoring (unrotated scan, identity
  // translation, SearchParameters(0,0,0.,0.)).
  float ScoreSingle(const PointCloud& cloud, const ProbabilityGrid& grid, int xo, int yo) const {
    const PointCloud scan = TransformPointCloud(cloud, RotationZ(0.f));
    std::vector<Cell2> d;
    for (const Vec3f& p : scan) d.push_back(grid.limits().GetCellIndex(p.x + 0.f, p.y + 0.f));
    return Score(grid, d, xo, yo, -yo * 0., -xo * 0., 0.);
  }

 private:
  // real_time_correlative_scan_matcher_2d.cc:110-135
  float Score(const ProbabilityGrid& grid, const std::vector<Cell2>& scan, int xo, int yo,
              double cx, double cy, double orientation) const {
    float score = 0.f;
    for (const Cell2& c : scan) score += grid.GetProbability(c.x + xo, c.y + yo);
    score /= static_cast<float>(scan.size());
    score *= std::exp(-Pow2(std::hypot(cx, cy) * options_.translation_delta_cost_weight +
                            std::abs(orientation) * options_.rotation_delta_cost_weight));
    return score;
  }
  const RealTimeCorrelativeScanMatcherOptions options_;
};

}  // namespace oracle

#endif  // ORACLE_OM_GRID2D_H_
