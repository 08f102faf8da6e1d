// RotationalScanMatcher::ComputeHistogram on the host
// (mapping/internal/3d/scan_matching/rotational_scan_matcher.cc:29-123,159-170), the per-scan O(N)
// step LocalTrajectoryBuilder3D runs after insertion (local_trajectory_builder_3d.cc:605-610) and
// whose result the loop-closure matcher consumes.  Order-dependent float accumulation over points
// sorted by angle inside 0.2 m height slices: kept on the host, operation by operation.
#include <algorithm>
#include <cmath>
#include <map>
#include <vector>

#include "../../include/dliom.h"

namespace {

struct P3 {
  float x, y, z;
};
constexpr float kMinDistance = 0.2f;
constexpr float kMaxDistance = 0.9f;
constexpr float kSliceHeight = 0.2f;

inline float norm2(float x, float y) { return std::sqrt(x * x + y * y); }

void add_value(float angle, float value, float* histogram, int size) {  // :35-50
  const float pi = static_cast<float>(M_PI);
  while (angle > pi) angle -= pi;
  while (angle < 0.f) angle += pi;
  const float zero_to_one = angle / pi;
  int bucket = static_cast<int>(std::lround(static_cast<float>(size) * zero_to_one - 0.5f));
  bucket = std::min(std::max(bucket, 0), size - 1);
  histogram[bucket] += value;
}

P3 centroid_of(const std::vector<P3>& slice) {  // :52-59
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (const P3& p : slice) {
    sx += p.x;
    sy += p.y;
    sz += p.z;
  }
  const float n = static_cast<float>(slice.size());
  return P3{sx / n, sy / n, sz / n};
}

std::vector<P3> sort_slice(const std::vector<P3>& slice) {  // :97-121
  struct Pair {
    bool operator<(const Pair& rhs) const { return angle < rhs.angle; }
    float angle;
    P3 point;
  };
  const P3 c = centroid_of(slice);
  std::vector<Pair> by_angle;
  by_angle.reserve(slice.size());
  for (const P3& p : slice) {
    const float dx = p.x - c.x, dy = p.y - c.y;
    if (norm2(dx, dy) < kMinDistance) continue;
    by_angle.push_back(Pair{std::atan2(dy, dx), p});
  }
  std::sort(by_angle.begin(), by_angle.end());
  std::vector<P3> out;
  out.reserve(by_angle.size());
  for (const Pair& p : by_angle) out.push_back(p.point);
  return out;
}

void add_slice(const std::vector<P3>& slice, float* histogram, int size) {  // :61-92
  if (slice.empty()) return;
  const P3 c = centroid_of(slice);
  P3 last = slice.front();
  for (const P3& p : slice) {
    const float dx = p.x - last.x, dy = p.y - last.y;
    const float ex = p.x - c.x, ey = p.y - c.y;
    const float distance = norm2(dx, dy), direction_norm = norm2(ex, ey);
    if (distance < kMinDistance || direction_norm < kMinDistance) continue;
    if (distance > kMaxDistance) {
      last = p;
      continue;
    }
    const float angle = std::atan2(dy, dx);
    const float dot = (dx / distance) * (ex / direction_norm) + (dy / distance) * (ey / direction_norm);
    add_value(angle, std::max(0.f, 1.f - std::abs(dot)), histogram, size);
  }
}

}  // namespace

extern "C" int dliom_rotational_histogram(const float* points_xyz, int64_t n, int histogram_size, float* histogram) {
  if (n < 0 || histogram_size <= 0 || histogram == nullptr || (n > 0 && points_xyz == nullptr))
    return DLIOM_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < histogram_size; ++i) histogram[i] = 0.f;
  std::map<int, std::vector<P3>> slices;
  for (int64_t i = 0; i < n; ++i) {
    const P3 p{points_xyz[3 * i], points_xyz[3 * i + 1], points_xyz[3 * i + 2]};
    slices[static_cast<int>(std::lround(p.z / kSliceHeight))].push_back(p);
  }
  for (const auto& s : slices) add_slice(sort_slice(s.second), histogram, histogram_size);
  return DLIOM_OK;
}
