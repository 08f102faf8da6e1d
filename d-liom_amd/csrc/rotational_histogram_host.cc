// RotationalScanMatcher::ComputeHistogram on the host
// (mapping/internal/3d/scan_matching/rotational_scan_matcher.cc:29-123,159-170), the per-scan O(N)
// step LocalTrajectoryBuilder3D runs after insertion (local_trajectory_builder_3d.cc:605-610) and
// whose result the loop-closure matcher consumes.  Order-dependent float accumulation over points
// sorted by angle inside 0.2 m height slices: kept on the host, operation by operation.
#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <map>
#include <thread>

#include <sched.h>
#include <vector>

#include "../../include/dliom.h"
#include "rotational.h"

namespace dliom {


// Eigen 3.3 vectorised redux over an aligned dynamic float vector (SSE2 packets of 4, two
// accumulators, (a0+a2)+(a1+a3), scalar tail): VectorXf::squaredNorm() / dot().
template <typename Term>
static float eigen_dyn_redux(int size, Term term) {
  const int ps = 4;
  const int end2 = (size / (2 * ps)) * (2 * ps), end1 = (size / ps) * ps;
  if (end1 == 0) {
    float r = term(0);
    for (int i = 1; i < size; ++i) r = r + term(i);
    return r;
  }
  float a[4], b[4];
  for (int l = 0; l < 4; ++l) a[l] = term(l);
  if (end1 > ps) {
    for (int l = 0; l < 4; ++l) b[l] = term(ps + l);
    for (int i = 2 * ps; i < end2; i += 2 * ps)
      for (int l = 0; l < 4; ++l) {
        a[l] = a[l] + term(i + l);
        b[l] = b[l] + term(i + ps + l);
      }
    for (int l = 0; l < 4; ++l) a[l] = a[l] + b[l];
    if (end1 > end2)
      for (int l = 0; l < 4; ++l) a[l] = a[l] + term(end2 + l);
  }
  float r = (a[0] + a[2]) + (a[1] + a[3]);
  for (int i = end1; i < size; ++i) r = r + term(i);
  return r;
}

std::vector<float> rotate_histogram(const std::vector<float>& h, float angle) {  // :125-144
  const int n = static_cast<int>(h.size());
  const float rotate_by_buckets = static_cast<float>(-angle * static_cast<float>(n) / M_PI);
  int full_buckets = static_cast<int>(std::lround(rotate_by_buckets - 0.5f));
  const float fraction = rotate_by_buckets - static_cast<float>(full_buckets);
  while (full_buckets < 0) full_buckets += n;
  std::vector<float> out(n);
  for (int i = 0; i != n; ++i)
    out[i] = fraction * h[(i + 1 + full_buckets) % n] + (1.f - fraction) * h[(i + full_buckets) % n];
  return out;
}

float match_histograms(const std::vector<float>& submap, const std::vector<float>& scan) {  // :146-157
  const int n = static_cast<int>(scan.size());
  const float scan_norm = std::sqrt(eigen_dyn_redux(n, [&](int i) { return scan[i] * scan[i]; }));
  const float submap_norm = std::sqrt(eigen_dyn_redux(n, [&](int i) { return submap[i] * submap[i]; }));
  const float normalization = scan_norm * submap_norm;
  if (normalization < 1e-3f) return 1.f;
  return eigen_dyn_redux(n, [&](int i) { return submap[i] * scan[i]; }) / normalization;
}


}  // namespace dliom

namespace {

struct P3 {
  float x, y, z;
};
constexpr float kMinDistance = 0.2f;
constexpr float kMaxDistance = 0.9f;
constexpr float kSliceHeight = 0.2f;

inline float norm2(float x, float y) { return std::sqrt(x * x + y * y); }

// AddValueToHistogram (:35-50) split in two: the bucket a value goes to ...
int bucket_of(float angle, int size) {
  const float pi = static_cast<float>(M_PI);
  while (angle > pi) angle -= pi;
  while (angle < 0.f) angle += pi;
  const float zero_to_one = angle / pi;
  const int bucket = static_cast<int>(std::lround(static_cast<float>(size) * zero_to_one - 0.5f));
  return std::min(std::max(bucket, 0), size - 1);
}
// ... and the addition itself, which is order dependent (float) and therefore stays sequential in slice order
struct Contribution {
  int bucket;
  float value;
};

P3 centroid_of(const P3* slice, size_t n) {  // :52-59
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (size_t i = 0; i < n; ++i) {
    sx += slice[i].x;
    sy += slice[i].y;
    sz += slice[i].z;
  }
  const float fn = static_cast<float>(n);
  return P3{sx / fn, sy / fn, sz / fn};
}

std::vector<P3> sort_slice(const P3* slice, size_t n) {  // :97-121
  struct Pair {
    bool operator<(const Pair& rhs) const { return angle < rhs.angle; }
    float angle;
    P3 point;
  };
  const P3 c = centroid_of(slice, n);
  std::vector<Pair> by_angle;
  by_angle.reserve(n);
  for (size_t i = 0; i < n; ++i) {
    const P3& p = slice[i];
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

// AddPointCloudSliceToHistogram (:61-92): the slice's contributions in the order the reference adds them
void slice_contributions(const std::vector<P3>& slice, int size, std::vector<Contribution>* out) {
  out->clear();
  if (slice.empty()) return;
  const P3 c = centroid_of(slice.data(), slice.size());
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
    out->push_back(Contribution{bucket_of(angle, size), std::max(0.f, 1.f - std::abs(dot))});
  }
}

// All threads of one call meet here between the phases (short phases: spin, then yield).
class SpinBarrier {
 public:
  explicit SpinBarrier(int n) : n_(n) {}
  void wait() {
    const int gen = generation_.load(std::memory_order_acquire);
    if (arrived_.fetch_add(1, std::memory_order_acq_rel) + 1 == n_) {
      arrived_.store(0, std::memory_order_relaxed);
      generation_.fetch_add(1, std::memory_order_release);
      return;
    }
    for (int spins = 0; generation_.load(std::memory_order_acquire) == gen; ++spins)
      if (spins > 2000) std::this_thread::yield();
  }

 private:
  const int n_;
  std::atomic<int> arrived_{0};
  std::atomic<int> generation_{0};
};

}  // namespace

// ComputeHistogram (:159-170).  The reference walks a std::map of height slices (key lround(z / 0.2), points in
// input order), sorts every slice by angle around its centroid and adds the slice's contributions to the histogram.
// Everything but those additions is independent per point or per slice: for whole scans (46 000 returns took 2.9 ms
// on one core of the GPU box's host, six times the device chain they follow) the keys, a stable counting sort into
// slices, and the per-slice sort + evaluation run on up to 8 host threads; the additions stay in slice order, so the
// bits do not depend on the thread count (tools/hist_bench.cc, dliom_rotational_histogram_mt).
extern "C" int dliom_rotational_histogram(const float* points_xyz, int64_t n, int histogram_size, float* histogram) {
  return dliom_rotational_histogram_mt(points_xyz, n, histogram_size, 0, histogram);
}

extern "C" int dliom_rotational_histogram_mt(const float* points_xyz, int64_t n, int histogram_size, int forced_threads,
                                             float* histogram) {
  if (n < 0 || histogram_size <= 0 || histogram == nullptr || (n > 0 && points_xyz == nullptr) || forced_threads < 0)
    return DLIOM_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < histogram_size; ++i) histogram[i] = 0.f;
  if (n == 0) return DLIOM_OK;
  unsigned hw = std::thread::hardware_concurrency();
  {  // hardware_concurrency() ignores cgroup / affinity limits; spinning barriers must not be oversubscribed
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
      const int allowed = CPU_COUNT(&set);
      if (allowed > 0) hw = std::min<unsigned>(hw == 0 ? static_cast<unsigned>(allowed) : hw, static_cast<unsigned>(allowed));
    }
  }
  const int T = static_cast<int>(std::max<int64_t>(
      1, std::min<int64_t>(n, forced_threads > 0 ? std::min(forced_threads, 64) : (n < 8192 ? 1 : std::min<unsigned>(hw == 0 ? 1 : hw, 8u)))));
  std::vector<int> key(static_cast<size_t>(n));
  std::vector<int> tmin(static_cast<size_t>(T), INT_MAX), tmax(static_cast<size_t>(T), INT_MIN);
  std::vector<std::vector<int64_t>> offset(static_cast<size_t>(T));  // per thread and slice: count, then write position
  std::vector<int64_t> slice_begin;
  std::vector<P3> flat;
  std::vector<std::vector<Contribution>> contributions;
  int kmin = 0;
  int64_t span = 0;
  bool sparse = false, out_of_memory = false;
  SpinBarrier barrier(T);
  std::atomic<int64_t> next(0);
  auto work = [&](int t) {
    const int64_t a = n * t / T, b = n * (t + 1) / T;
    int lo = INT_MAX, hi = INT_MIN;
    for (int64_t i = a; i < b; ++i) {
      const int k = static_cast<int>(std::lround(points_xyz[3 * i + 2] / kSliceHeight));
      key[static_cast<size_t>(i)] = k;
      lo = std::min(lo, k);
      hi = std::max(hi, k);
    }
    tmin[static_cast<size_t>(t)] = lo;
    tmax[static_cast<size_t>(t)] = hi;
    barrier.wait();
    if (t == 0) {
      int gmin = INT_MAX, gmax = INT_MIN;
      for (int u = 0; u < T; ++u) {
        gmin = std::min(gmin, tmin[static_cast<size_t>(u)]);
        gmax = std::max(gmax, tmax[static_cast<size_t>(u)]);
      }
      kmin = gmin;
      span = static_cast<int64_t>(gmax) - gmin + 1;
      sparse = span > 4 * n + 1024;  // absurdly spread heights: the map-based walk below
      if (!sparse) {
        try {
          for (int u = 0; u < T; ++u) offset[static_cast<size_t>(u)].assign(static_cast<size_t>(span), 0);
          slice_begin.assign(static_cast<size_t>(span) + 1, 0);
          flat.resize(static_cast<size_t>(n));
          contributions.resize(static_cast<size_t>(span));
        } catch (...) {  // out of memory: every worker leaves at the next barrier, the call reports it
          out_of_memory = true;
          sparse = true;
        }
      }
    }
    barrier.wait();
    if (sparse) return;
    std::vector<int64_t>& mine = offset[static_cast<size_t>(t)];
    for (int64_t i = a; i < b; ++i) ++mine[static_cast<size_t>(key[static_cast<size_t>(i)] - kmin)];
    barrier.wait();
    if (t == 0) {  // slice k holds thread 0's points, then thread 1's, ...: the input order
      int64_t running = 0;
      for (int64_t k = 0; k < span; ++k) {
        slice_begin[static_cast<size_t>(k)] = running;
        for (int u = 0; u < T; ++u) {
          const int64_t c = offset[static_cast<size_t>(u)][static_cast<size_t>(k)];
          offset[static_cast<size_t>(u)][static_cast<size_t>(k)] = running;
          running += c;
        }
      }
      slice_begin[static_cast<size_t>(span)] = running;
    }
    barrier.wait();
    for (int64_t i = a; i < b; ++i)
      flat[static_cast<size_t>(mine[static_cast<size_t>(key[static_cast<size_t>(i)] - kmin)]++)] =
          P3{points_xyz[3 * i], points_xyz[3 * i + 1], points_xyz[3 * i + 2]};
    barrier.wait();
    for (int64_t k = next.fetch_add(1); k < span; k = next.fetch_add(1)) {
      const int64_t first = slice_begin[static_cast<size_t>(k)], count = slice_begin[static_cast<size_t>(k) + 1] - first;
      if (count == 0) continue;
      slice_contributions(sort_slice(flat.data() + first, static_cast<size_t>(count)), histogram_size,
                          &contributions[static_cast<size_t>(k)]);
    }
  };
  if (T == 1) {
    work(0);
  } else {
    // all T workers meet at barriers sized for T: they are started only if every one of them can be (a failed
    // std::thread construction would otherwise leave the started ones spinning for a partner that never comes)
    std::vector<std::thread> pool;
    std::atomic<int> go(0);  // 0 wait, 1 run, -1 give up
    auto gated = [&](int t) {
      while (go.load(std::memory_order_acquire) == 0) std::this_thread::yield();
      if (go.load(std::memory_order_acquire) > 0) work(t);
    };
    bool all_started = true;
    try {
      for (int t = 1; t < T; ++t) pool.emplace_back(gated, t);
    } catch (...) {
      all_started = false;
    }
    if (!all_started) {
      go.store(-1, std::memory_order_release);
      for (std::thread& th : pool) th.join();
      return dliom_rotational_histogram_mt(points_xyz, n, histogram_size, 1, histogram);  // the single-thread path
    }
    go.store(1, std::memory_order_release);
    work(0);
    for (std::thread& th : pool) th.join();
  }
  if (out_of_memory) return DLIOM_ERR_CAPACITY;
  if (sparse) {
    std::map<int, std::vector<P3>> slices;
    for (int64_t i = 0; i < n; ++i)
      slices[key[static_cast<size_t>(i)]].push_back(P3{points_xyz[3 * i], points_xyz[3 * i + 1], points_xyz[3 * i + 2]});
    std::vector<Contribution> c;
    for (const auto& sl : slices) {
      slice_contributions(sort_slice(sl.second.data(), sl.second.size()), histogram_size, &c);
      for (const Contribution& k : c) histogram[k.bucket] += k.value;
    }
    return DLIOM_OK;
  }
  for (int64_t k = 0; k < span; ++k)
    for (const Contribution& c : contributions[static_cast<size_t>(k)]) histogram[c.bucket] += c.value;
  return DLIOM_OK;
}

// RotationalScanMatcher(histograms_at_angles).Match(histogram, initial_angle, angles)
// (rotational_scan_matcher.cc:174-194): the submap histogram is the sum of the node histograms
// rotated by their yaws; one score per angle.
extern "C" int dliom_rotational_scan_match(const float* node_histograms, const float* node_angles, int num_nodes,
                                           int histogram_size, const float* scan_histogram, float initial_angle,
                                           const float* angles, int num_angles, float* scores) {
  if (node_histograms == nullptr || node_angles == nullptr || num_nodes <= 0 || histogram_size <= 0 ||
      scan_histogram == nullptr || num_angles < 0 || (num_angles > 0 && (angles == nullptr || scores == nullptr)))
    return DLIOM_ERR_INVALID_ARGUMENT;
  std::vector<float> submap(histogram_size, 0.f);
  for (int k = 0; k < num_nodes; ++k) {
    const std::vector<float> h(node_histograms + static_cast<size_t>(k) * histogram_size,
                               node_histograms + static_cast<size_t>(k + 1) * histogram_size);
    const std::vector<float> r = dliom::rotate_histogram(h, node_angles[k]);
    for (int i = 0; i < histogram_size; ++i) submap[i] += r[i];
  }
  const std::vector<float> scan(scan_histogram, scan_histogram + histogram_size);
  for (int i = 0; i < num_angles; ++i)
    scores[i] = dliom::match_histograms(submap, dliom::rotate_histogram(scan, initial_angle + angles[i]));
  return DLIOM_OK;
}
