// RealTimeCorrelativeScanMatcher2D over a dense ProbabilityGrid -- BASELINE config 1
// ("plumbing, no GPU"): by contract this one runs on the host.
//
//   mapping/internal/2d/scan_matching/real_time_correlative_scan_matcher_2d.cc:40-135
//   mapping/internal/2d/scan_matching/correlative_scan_matcher_2d.cc:27-127
//   mapping/2d/map_limits.h:69-88, mapping/2d/probability_grid.cc:69-73
//   mapping/probability_values.cc:27-36 (correspondence-cost decoding)
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/dliom.h"

namespace {

struct Limits {
  double resolution, max_x, max_y;
  int nx, ny;
  // map_limits.h:69-76: (row from max.y, column from max.x)
  void Cell(float px, float py, int* cx, int* cy) const {
    *cx = static_cast<int>(std::lround((max_y - py) / resolution - 0.5));
    *cy = static_cast<int>(std::lround((max_x - px) / resolution - 0.5));
  }
  bool Contains(int cx, int cy) const { return cx >= 0 && cy >= 0 && cx < nx && cy < ny; }
};

// value -> correspondence cost -> probability, all in float.
struct CostTable {
  std::vector<float> probability;  // 1 - cost, indexed by the 16-bit cell value
  CostTable() : probability(65536) {
    const float lo = 1.f - (1.f - 0.1f);  // kMinCorrespondenceCost
    const float hi = 1.f - 0.1f;          // kMaxCorrespondenceCost
    const float scale = (hi - lo) / 32766.f;
    for (int v = 0; v < 32768; ++v) {
      const float cost = v == 0 ? hi : v * scale + (lo - scale);
      probability[v] = 1.f - cost;
      probability[v + 32768] = probability[v];
    }
  }
};

struct P3 {
  float x, y, z;
};

// Rigid3f::Rotation(AngleAxisf(angle, UnitZ)) * p with Eigen's _transformVector order.
P3 RotateZ(float angle, const P3& v) {
  const float w = std::cos(0.5f * angle), qz = std::sin(0.5f * angle), qx = 0.f, qy = 0.f;
  float uvx = qy * v.z - qz * v.y, uvy = qz * v.x - qx * v.z, uvz = qx * v.y - qy * v.x;
  uvx += uvx;
  uvy += uvy;
  uvz += uvz;
  const float cx = qy * uvz - qz * uvy, cy = qz * uvx - qx * uvz, cz = qx * uvy - qy * uvx;
  return P3{((v.x + w * uvx) + cx) + 0.f, ((v.y + w * uvy) + cy) + 0.f, ((v.z + w * uvz) + cz) + 0.f};
}

}  // namespace

extern "C" int dliom_rtcsm2d_match(const dliom_rtcsm_options* o, const double initial_pose[3],
                                   const float* points_xyz, int64_t n, const uint16_t* cells,
                                   int num_x_cells, int num_y_cells, double resolution, double max_x,
                                   double max_y, double pose_estimate[3], double* score) {
  if (o == nullptr || initial_pose == nullptr || pose_estimate == nullptr || score == nullptr ||
      cells == nullptr || n < 0 || (n > 0 && points_xyz == nullptr) || num_x_cells <= 0 ||
      num_y_cells <= 0 || !(resolution > 0.))
    return DLIOM_ERR_INVALID_ARGUMENT;
  if (n == 0) return DLIOM_ERR_EMPTY_CLOUD;
  static const CostTable table;
  const Limits lim{resolution, max_x, max_y, num_x_cells, num_y_cells};
  // rotate by the initial yaw (rtcsm_2d.cc:81-85)
  const float yaw = static_cast<float>(initial_pose[2]);
  std::vector<P3> rotated(static_cast<size_t>(n));
  for (int64_t i = 0; i < n; ++i)
    rotated[i] = RotateZ(yaw, P3{points_xyz[3 * i], points_xyz[3 * i + 1], points_xyz[3 * i + 2]});
  // SearchParameters (correlative_scan_matcher_2d.cc:27-55)
  float max_scan_range = static_cast<float>(3.f * resolution);
  for (const P3& p : rotated) max_scan_range = std::max(std::sqrt(p.x * p.x + p.y * p.y), max_scan_range);
  const double kSafetyMargin = 1. - 1e-3;
  const double step =
      kSafetyMargin * std::acos(1. - (resolution * (resolution * 1.)) /
                                         (2. * static_cast<double>(max_scan_range * (max_scan_range * 1.f))));
  const int num_angular = static_cast<int>(std::ceil(o->angular_search_window / step));
  const int num_scans = 2 * num_angular + 1;
  const int num_linear = static_cast<int>(std::ceil(o->linear_search_window / resolution));
  const float tx = static_cast<float>(initial_pose[0]), ty = static_cast<float>(initial_pose[1]);
  float best_score = -1.f;
  double best_x = 0, best_y = 0, best_o = 0;
  bool have = false;
  std::vector<int> cx(static_cast<size_t>(n)), cy(static_cast<size_t>(n));
  double delta_theta = -num_angular * step;
  for (int s = 0; s < num_scans; ++s, delta_theta += step) {
    // GenerateRotatedScans + DiscretizeScans for this scan
    const float th = static_cast<float>(delta_theta);
    for (int64_t i = 0; i < n; ++i) {
      const P3 p = RotateZ(th, rotated[i]);
      lim.Cell(p.x + tx, p.y + ty, &cx[i], &cy[i]);
    }
    const double orientation = (s - num_angular) * step;
    for (int xo = -num_linear; xo <= num_linear; ++xo)
      for (int yo = -num_linear; yo <= num_linear; ++yo) {
        float sum = 0.f;
        for (int64_t i = 0; i < n; ++i) {
          const int x = cx[i] + xo, y = cy[i] + yo;
          sum += lim.Contains(x, y) ? table.probability[cells[static_cast<size_t>(num_x_cells) * y + x]] : 0.1f;
        }
        float sc = sum / static_cast<float>(n);
        const double px = -yo * resolution, py = -xo * resolution;
        const double arg = std::hypot(px, py) * o->translation_delta_cost_weight +
                           std::abs(orientation) * o->rotation_delta_cost_weight;
        sc *= std::exp(-(arg * (arg * 1.)));
        if (!(sc > 0.f)) return DLIOM_ERR_SCORE_NOT_POSITIVE;  // CHECK_GT(candidate.score, 0.f)
        if (!have || best_score < sc) {  // std::max_element: first maximum
          have = true;
          best_score = sc;
          best_x = px;
          best_y = py;
          best_o = orientation;
        }
      }
  }
  pose_estimate[0] = initial_pose[0] + best_x;
  pose_estimate[1] = initial_pose[1] + best_y;
  pose_estimate[2] = initial_pose[2] + best_o;
  *score = best_score;
  return DLIOM_OK;
}
