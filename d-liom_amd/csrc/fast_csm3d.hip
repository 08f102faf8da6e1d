// FastCorrelativeScanMatcher3D (loop-closure branch and bound) on the device.
//
//   mapping/internal/3d/scan_matching/precomputation_grid_3d.cc:49-82   uint8 max-pool pyramid
//   mapping/internal/3d/scan_matching/fast_correlative_scan_matcher_3d.cc
//       :57-77   PrecomputationGridStack3D        -> convert_level0_kernel, precompute_level_kernel
//       :264-304 DiscretizeScan                   -> discretize_kernel (cells per discrete scan)
//       :394-417 ScoreCandidates                  -> score_candidates_kernel (exact integer sums)
//       :439-492 BranchAndBound                   -> host recursion (same order, same std::sort),
//                                                    scores served from a cache that the device
//                                                    fills in batches
//   mapping/internal/3d/scan_matching/rotational_scan_matcher.cc:125-194 histogram matching (host)
//   mapping/internal/3d/scan_matching/low_resolution_matcher.cc:23-36   -> the exact sequential
//                                                    float sum kernels of rtcsm3d.hip
//
// Exactness: a candidate's score is ToProbability(sum / float(N)) of an INTEGER sum, so any
// summation order on the device gives the reference's float; the traversal (which candidates are
// expanded, in which order equal scores are visited) is replayed on the host with the reference's
// own comparisons and std::sort calls.  The pyramid levels are dense uint8 boxes over the bounding
// box of the submap's leaves (a max-pool is order independent); cells outside read 0 like
// HybridGridBase<uint8>::value() of an unset cell.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <vector>

#include "device_common.h"
#include "host_math.h"
#include "internal.h"
#include "rotational.h"

namespace dliom {

struct LevelView {
  const uint8_t* data;
  int lo[3];  // cell index of element (0,0,0)
  int n[3];   // cells per axis (0: empty level)
};

__device__ __forceinline__ unsigned level_value(const LevelView& l, int x, int y, int z) {
  const unsigned ux = static_cast<unsigned>(x - l.lo[0]), uy = static_cast<unsigned>(y - l.lo[1]),
                 uz = static_cast<unsigned>(z - l.lo[2]);
  if (ux >= static_cast<unsigned>(l.n[0]) || uy >= static_cast<unsigned>(l.n[1]) || uz >= static_cast<unsigned>(l.n[2]))
    return 0u;
  return l.data[(static_cast<size_t>(uz) * l.n[1] + uy) * l.n[0] + ux];
}

// Bounding box of the allocated leaves (leaf coordinates = voxel index >> 3): mm[0..2] min, [3..5] max.
__global__ void leaf_bbox_kernel(const int32_t* __restrict__ slot_coord, unsigned count, int* __restrict__ mm) {
  const unsigned s = blockIdx.x * blockDim.x + threadIdx.x + 1;  // slot 0 is the null leaf
  if (s >= count) return;
  for (int a = 0; a < 3; ++a) {
    const int c = slot_coord[3 * static_cast<size_t>(s) + a];
    atomicMin(&mm[a], c);
    atomicMax(&mm[3 + a], c);
  }
}

// ConvertToPrecomputationGrid: one workgroup per leaf, cell = lut8[value & 0x7fff].
__global__ void convert_level0_kernel(const int32_t* __restrict__ slot_coord, const uint16_t* __restrict__ pool,
                                      const uint8_t* __restrict__ lut8, uint8_t* __restrict__ out, int lox, int loy,
                                      int loz, int nx, int ny) {
  const size_t s = static_cast<size_t>(blockIdx.x) + 1;
  const int bx = slot_coord[3 * s] * 8 - lox, by = slot_coord[3 * s + 1] * 8 - loy, bz = slot_coord[3 * s + 2] * 8 - loz;
  for (int c = threadIdx.x; c < 512; c += blockDim.x) {
    const unsigned v = pool[s * 512 + c] & 0x7FFFu;
    out[(static_cast<size_t>(bz + (c >> 6)) * ny + (by + ((c >> 3) & 7))) * nx + (bx + (c & 7))] = lut8[v];
  }
}

// PrecomputeGrid as a gather: dst(t) = max over the source cells c and octants o with
// (c - shift o) [>> 1] == t  (8 sources at full resolution, 64 when the resolution halves).
__global__ void precompute_level_kernel(LevelView src, uint8_t* __restrict__ dst, int dlox, int dloy, int dloz, int dnx,
                                        int dny, int dnz, int shift, int half) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t total = static_cast<size_t>(dnx) * dny * dnz;
  if (i >= total) return;
  const int tx = static_cast<int>(i % dnx) + dlox, ty = static_cast<int>((i / dnx) % dny) + dloy,
            tz = static_cast<int>(i / (static_cast<size_t>(dnx) * dny)) + dloz;
  unsigned m = 0;
  const int sub = half ? 2 : 1;
  for (int oz = 0; oz < 2; ++oz)
    for (int sz = 0; sz < sub; ++sz)
      for (int oy = 0; oy < 2; ++oy)
        for (int sy = 0; sy < sub; ++sy)
          for (int ox = 0; ox < 2; ++ox)
            for (int sx = 0; sx < sub; ++sx) {
              const int cx = (half ? 2 * tx + sx : tx) + shift * ox, cy = (half ? 2 * ty + sy : ty) + shift * oy,
                        cz = (half ? 2 * tz + sz : tz) + shift * oz;
              m = max(m, level_value(src, cx, cy, cz));
            }
  dst[i] = static_cast<uint8_t>(m);
}

// DiscretizeScan: cell index of pose_s * p for every discrete scan s and point p.
__global__ void discretize_kernel(const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz,
                                  int n, const float* __restrict__ poses7, float resolution, int* __restrict__ cx,
                                  int* __restrict__ cy, int* __restrict__ cz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = blockIdx.y;
  const float* p = poses7 + 7 * s;
  const Quat4 q{p[3], p[4], p[5], p[6]};
  float rx, ry, rz;
  rotate_point(q, px[i], py[i], pz[i], rx, ry, rz);
  const size_t o = static_cast<size_t>(s) * n + i;
  cx[o] = cell_of(rx + p[0], resolution);
  cy[o] = cell_of(ry + p[1], resolution);
  cz[o] = cell_of(rz + p[2], resolution);
}

struct ScoreArgs {
  LevelView level;
  const int *cx, *cy, *cz;  // [scan][point] full-resolution cells
  int n;                    // points per scan
  int e;                    // reduction exponent max(0, depth - full_resolution_depth + 1)
  int start[3];             // search_window_start (negative window sizes)
};

// ScoreCandidates: blockIdx.x = candidate (scan, full-resolution offset), blockIdx.y = chunk of
// kScoreChunk points; lanes over points, block reduction, one atomicAdd per (candidate, chunk).
constexpr int kScoreChunk = 2048;
__global__ __launch_bounds__(256) void score_candidates_kernel(ScoreArgs a, const int4* __restrict__ candidates,
                                                               int num_candidates, int* __restrict__ sums) {
  __shared__ unsigned wave_sums[4];
  const int c = blockIdx.x;
  const int4 cand = candidates[c];
  const int ox = cand.y >> a.e, oy = cand.z >> a.e, oz = cand.w >> a.e;
  const int lsx = a.start[0] >> a.e, lsy = a.start[1] >> a.e, lsz = a.start[2] >> a.e;
  const size_t base = static_cast<size_t>(cand.x) * a.n;
  const int p_begin = blockIdx.y * kScoreChunk, p_end = min(a.n, p_begin + kScoreChunk);
  unsigned sum = 0;
  for (int p = p_begin + threadIdx.x; p < p_end; p += 256) {
    int x = a.cx[base + p], y = a.cy[base + p], z = a.cz[base + p];
    if (a.e > 0) {  // low-resolution cells (:285-301)
      x = ((x + a.start[0]) >> a.e) - lsx;
      y = ((y + a.start[1]) >> a.e) - lsy;
      z = ((z + a.start[2]) >> a.e) - lsz;
    }
    sum += level_value(a.level, x + ox, y + oy, z + oz);
  }
  sum = wave_sum_lane63(sum);
  if ((threadIdx.x & 63) == 63) wave_sums[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&sums[c], static_cast<int>(wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3]));
}


// ---- device frontier -----------------------------------------------------------------------------
// BranchAndBound (:439-492) asks for the scores of the children of every node whose score beats the best leaf found
// so far.  Served one sibling group per launch that is ~110 launches + synchronisations for a whole-submap window
// on a filtered scan (150 points): 3 ms, all of it latency.  Instead ONE chain of launches, no host in between:
//   1. the lowest-resolution candidates (host list) are scored;
//   2. one workgroup walks the branch the recursion walks first -- best candidate, best child, ... down to a leaf --
//      whose score theta is (up to ties and the low-resolution check) the bound the recursion prunes with from its
//      first leaf on;
//   3. level by level, the children of every node with score >= max(theta, min_score) are scored and appended to
//      that level's list: a superset of everything the recursion can ask for once it holds a leaf of score theta,
//      and (>=, upper bounds) of the branch it walks before;
//   4. all lists are packed into pinned host memory; one synchronisation.
// The host then replays the reference's recursion -- same comparisons, same std::sort calls -- out of a hash table;
// anything missing (a first leaf refused by the low-resolution matcher, a list that hit its capacity) is fetched
// on demand as before.  Exactness is untouched: scores are functions of exact integer sums.
constexpr int kMaxLevels = 16;
struct FrontierRec {
  int scan, ox, oy, oz, sum;
};
struct FrontierArgs {
  LevelView level[kMaxLevels];
  const int *cx, *cy, *cz;  // [scan][point] full-resolution cells
  int n;                    // points per scan
  int max_depth, full_resolution_depth;
  int linear_xy, linear_z;
  FrontierRec* pool;        // (max_depth + 1) lists of `cap` records, list d at pool + d * cap
  int* counts;              // [kMaxLevels] records per list, [kMaxLevels] overflow flag, [kMaxLevels + 1] theta bits
  int cap;
  float min_score;
};

__device__ __forceinline__ float frontier_probability(int sum, int n) {  // ScoreCandidates' float (:407-411)
  const float kMin = 0.1f, kMax = 1.f - 0.1f;
  return kMin + (static_cast<float>(sum) / static_cast<float>(n)) * ((kMax - kMin) / 255.f);
}

// Integer sum of one candidate at `depth`, by one wavefront (every lane returns it).
__device__ __forceinline__ int frontier_wave_sum(const FrontierArgs& a, int depth, int scan, int ox, int oy, int oz, int lane) {
  const int e = max(0, depth - a.full_resolution_depth + 1);
  const int sx = -a.linear_xy, sz = -a.linear_z;
  const int lsx = sx >> e, lsz = sz >> e;
  const LevelView& lv = a.level[depth];
  const size_t base = static_cast<size_t>(scan) * a.n;
  const int cox = ox >> e, coy = oy >> e, coz = oz >> e;
  unsigned sum = 0;
  for (int p = lane; p < a.n; p += 64) {
    int x = a.cx[base + p], y = a.cy[base + p], z = a.cz[base + p];
    if (e > 0) {  // low-resolution cells (:285-301)
      x = ((x + sx) >> e) - lsx;
      y = ((y + sx) >> e) - lsx;
      z = ((z + sz) >> e) - lsz;
    }
    sum += level_value(lv, x + cox, y + coy, z + coz);
  }
  return __builtin_amdgcn_readlane(static_cast<int>(wave_sum_lane63(sum)), 63);
}

// 1. sums of the uploaded lowest-resolution list (list max_depth), one wavefront per candidate
__global__ __launch_bounds__(256) void frontier_top_kernel(FrontierArgs a, int count) {
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (w >= count) return;
  FrontierRec* r = a.pool + static_cast<size_t>(a.max_depth) * a.cap + w;
  const int sum = frontier_wave_sum(a, a.max_depth, r->scan, r->ox, r->oy, r->oz, lane);
  if (lane == 0) r->sum = sum;
}

// 2. the branch the recursion walks first; writes theta
__global__ __launch_bounds__(512) void frontier_greedy_kernel(FrontierArgs a, int top_count) {
  __shared__ unsigned long long best_key[8];
  __shared__ int child_sum[8];
  __shared__ int cur[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const FrontierRec* top = a.pool + static_cast<size_t>(a.max_depth) * a.cap;
  unsigned long long key = 0ull;
  for (int i = threadIdx.x; i < top_count; i += blockDim.x)
    key = max(key, (static_cast<unsigned long long>(static_cast<unsigned>(top[i].sum)) << 32) | static_cast<unsigned>(0x7FFFFFFF - i));
  for (int m = 32; m >= 1; m >>= 1) key = max(key, __shfl_xor(key, m));
  if (lane == 0) best_key[wave] = key;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long k = 0ull;
    for (int w = 0; w < 8; ++w) k = max(k, best_key[w]);
    const int i = 0x7FFFFFFF - static_cast<int>(k & 0xFFFFFFFFull);
    cur[0] = top[i].scan;
    cur[1] = top[i].ox;
    cur[2] = top[i].oy;
    cur[3] = top[i].oz;
    child_sum[0] = static_cast<int>(k >> 32);
  }
  __syncthreads();
  int sum = child_sum[0];
  __syncthreads();
  for (int depth = a.max_depth; depth >= 1; --depth) {
    const int hw = 1 << (depth - 1);
    const int scan = cur[0], ox = cur[1] + ((wave & 1) ? hw : 0), oy = cur[2] + ((wave & 2) ? hw : 0), oz = cur[3] + ((wave & 4) ? hw : 0);
    const bool valid = ox <= a.linear_xy && oy <= a.linear_xy && oz <= a.linear_z;  // children_of (:468-484)
    const int s = valid ? frontier_wave_sum(a, depth - 1, scan, ox, oy, oz, lane) : -1;
    if (lane == 0) child_sum[wave] = s;
    __syncthreads();
    int bw = 0;
    for (int w = 1; w < 8; ++w)
      if (child_sum[w] > child_sum[bw]) bw = w;
    sum = child_sum[bw];
    __syncthreads();
    if (threadIdx.x == 0) {
      cur[1] += (bw & 1) ? hw : 0;
      cur[2] += (bw & 2) ? hw : 0;
      cur[3] += (bw & 4) ? hw : 0;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float theta = fmaxf(frontier_probability(sum, a.n), a.min_score);
    a.counts[kMaxLevels + 1] = __float_as_int(theta);
  }
}

// 3. list `depth` -> list depth - 1: the children of every node with score >= theta, one wavefront per child
__global__ __launch_bounds__(256) void frontier_level_kernel(FrontierArgs a, int depth) {
  const int lane = threadIdx.x & 63;
  const int waves = (gridDim.x * blockDim.x) >> 6;
  const int parents = min(a.counts[depth], a.cap);
  const float theta = __int_as_float(a.counts[kMaxLevels + 1]);
  const FrontierRec* src = a.pool + static_cast<size_t>(depth) * a.cap;
  FrontierRec* dst = a.pool + static_cast<size_t>(depth - 1) * a.cap;
  const int hw = 1 << (depth - 1);
  for (int item = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; item < 8 * parents; item += waves) {
    const FrontierRec pr = src[item >> 3];
    if (!(frontier_probability(pr.sum, a.n) >= theta)) continue;
    const int c = item & 7;
    const int ox = pr.ox + ((c & 1) ? hw : 0), oy = pr.oy + ((c & 2) ? hw : 0), oz = pr.oz + ((c & 4) ? hw : 0);
    if (ox > a.linear_xy || oy > a.linear_xy || oz > a.linear_z) continue;
    const int sum = frontier_wave_sum(a, depth - 1, pr.scan, ox, oy, oz, lane);
    if (lane == 0) {
      const int at = atomicAdd(&a.counts[depth - 1], 1);
      if (at < a.cap)
        dst[at] = FrontierRec{pr.scan, ox, oy, oz, sum};
      else
        a.counts[kMaxLevels] = 1;  // overflow: the host fetches what is missing on demand
    }
  }
}

// 4. [counts | overflow | theta | records of list max_depth, ..., 0] -> pinned host memory
__global__ __launch_bounds__(256) void frontier_pack_kernel(FrontierArgs a, int* __restrict__ out, int out_records) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
  if (tid < kMaxLevels + 2) out[tid] = a.counts[tid];
  int first = 0;
  for (int depth = a.max_depth; depth >= 0; --depth) {
    const int cnt = min(a.counts[depth], a.cap);
    const int* src = reinterpret_cast<const int*>(a.pool + static_cast<size_t>(depth) * a.cap);
    for (int i = tid; i < 5 * cnt; i += nthreads)
      if (first + i / 5 < out_records) out[kMaxLevels + 2 + 5 * first + i] = src[i];
    first += cnt;
  }
}

// ---- host side ---------------------------------------------------------------------------------
// rotate_histogram / match_histograms live in rotational_histogram.cc (host-only, CPU-testable).
using Histogram = std::vector<float>;

static QF quat_inverse(const QF& q) {  // Eigen QuaternionBase::inverse()
  const float n2 = (q.x * q.x + q.z * q.z) + (q.y * q.y + q.w * q.w);
  if (n2 > 0.f) return QF{q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
  return QF{0.f, 0.f, 0.f, 0.f};
}
static float get_yaw(const QF& q) {  // transform/transform.h:41-46
  const F3 d = qrot(q, F3{1.f, 0.f, 0.f});
  return std::atan2(d.y, d.x);
}
static PoseF pose_mul_f(const PoseF& a, const PoseF& b) {  // rigid_transform.h:206-212
  return PoseF{add3(qrot(a.q, b.t), a.t), qnormalized(qmul(a.q, b.q))};
}
static PoseF pose_inverse_f(const PoseF& a) {  // rigid_transform.h:167-171
  const QF r{a.q.w, -a.q.x, -a.q.y, -a.q.z};
  const F3 t = qrot(r, a.t);
  return PoseF{F3{-t.x, -t.y, -t.z}, r};
}

}  // namespace dliom

using namespace dliom;

struct dliom_fast_csm {
  dliom_ctx* ctx = nullptr;
  dliom_fast_csm_options options;
  float resolution = 0.f;
  int width_in_voxels = 0;
  const dliom_grid* lo_grid = nullptr;
  Histogram submap_histogram;
  struct Level {
    uint8_t* d = nullptr;
    int lo[3] = {0, 0, 0};
    int n[3] = {0, 0, 0};
    size_t cells() const { return static_cast<size_t>(n[0]) * n[1] * n[2]; }
    LevelView view() const { return LevelView{d, {lo[0], lo[1], lo[2]}, {n[0], n[1], n[2]}}; }
  };
  std::vector<Level> levels;
  int max_depth() const { return static_cast<int>(levels.size()) - 1; }
  ~dliom_fast_csm() {
    for (Level& l : levels)
      if (l.d != nullptr) (void)hipFree(l.d);
  }
};

namespace {

struct Candidate {
  int scan_index;
  int offset[3];
  float score = -std::numeric_limits<float>::infinity();
  float low_resolution_score = 0.f;
  bool operator<(const Candidate& o) const { return score < o.score; }
  bool operator>(const Candidate& o) const { return score > o.score; }
};

struct Search {
  const dliom_fast_csm* m;
  dliom_ctx* ctx;  // the CALLER's context: scratch, stream, pinned block (the matcher itself is read-only)
  int linear_xy, linear_z;
  double angular_window;
  // per match
  std::vector<PoseF> scan_poses;
  std::vector<float> rotational_scores;
  int n_hi = 0;
  int *d_cx = nullptr, *d_cy = nullptr, *d_cz = nullptr;
  const dliom_cloud* lo_cloud = nullptr;
  float min_low_resolution_score_f = 0.f;
  long long scored = 0;
  long long launches = 0;
  const std::vector<Candidate>* top = nullptr;  // the sorted lowest-resolution candidates
  float initial_min_score = 0.f;
  float frontier_threshold = std::numeric_limits<float>::infinity();  // last wavefront prefetch ran with this
  // score cache: (depth, scan, offset) -> integer sum.  Open addressing, linear probing (a match looks up and inserts
  // ~1e4 keys: std::unordered_map's node allocations were a third of the host time of a whole-submap match)
  struct ScoreCache {
    static constexpr uint64_t kEmpty = ~0ull;
    std::vector<uint64_t> keys;
    std::vector<int> vals;
    size_t used = 0;
    ScoreCache() { rehash(1u << 12); }
    static size_t mix(uint64_t k) {
      k ^= k >> 33;
      k *= 0xff51afd7ed558ccdull;
      k ^= k >> 33;
      return static_cast<size_t>(k);
    }
    void rehash(size_t cap) {
      std::vector<uint64_t> ok;
      std::vector<int> ov;
      ok.swap(keys);
      ov.swap(vals);
      keys.assign(cap, kEmpty);
      vals.assign(cap, 0);
      used = 0;
      for (size_t i = 0; i < ok.size(); ++i)
        if (ok[i] != kEmpty) put(ok[i], ov[i]);
    }
    const int* get(uint64_t k) const {
      const size_t mask = keys.size() - 1;
      for (size_t i = mix(k) & mask;; i = (i + 1) & mask) {
        if (keys[i] == k) return &vals[i];
        if (keys[i] == kEmpty) return nullptr;
      }
    }
    bool has(uint64_t k) const { return get(k) != nullptr; }
    void put(uint64_t k, int v) {
      if (2 * (used + 1) > keys.size()) rehash(2 * keys.size());
      const size_t mask = keys.size() - 1;
      for (size_t i = mix(k) & mask;; i = (i + 1) & mask) {
        if (keys[i] == k) {
          vals[i] = v;
          return;
        }
        if (keys[i] == kEmpty) {
          keys[i] = k;
          vals[i] = v;
          ++used;
          return;
        }
      }
    }
  } cache;
  static uint64_t key(int depth, int scan, const int* o) {
    // offsets fit 14 bits + sign for any window the 8-bit grid extent allows; scans < 2^16
    return (static_cast<uint64_t>(depth & 0xF) << 60) | (static_cast<uint64_t>(scan & 0xFFFF) << 44) |
           (static_cast<uint64_t>((o[0] + 8192) & 0x3FFF) << 28) | (static_cast<uint64_t>((o[1] + 8192) & 0x3FFF) << 14) |
           static_cast<uint64_t>((o[2] + 8192) & 0x3FFF);
  }
};

// Integer sums of `list` at `depth` (device), in list order.
int device_sums(Search& s, int depth, const std::vector<Candidate>& list, std::vector<int>* sums) {
  const dliom_fast_csm* m = s.m;
  dliom_ctx* ctx = s.ctx;
  const size_t k = list.size();
  sums->assign(k, 0);
  if (k == 0) return DLIOM_OK;
  const size_t cbytes = (k * 16 + 255) & ~static_cast<size_t>(255);
  DLIOM_TRY(ctx->cand.reserve(cbytes + k * 4));
  // candidate list and sums travel through the pinned block when they fit (no staging copies)
  const bool pinned = cbytes + k * 4 <= ctx->pinned_bytes - 8192;
  std::vector<int> pageable;
  int* host = static_cast<int*>(ctx->pinned);
  if (!pinned) {
    pageable.resize(4 * k + k);
    host = pageable.data();
  }
  for (size_t i = 0; i < k; ++i) {
    host[4 * i] = list[i].scan_index;
    host[4 * i + 1] = list[i].offset[0];
    host[4 * i + 2] = list[i].offset[1];
    host[4 * i + 3] = list[i].offset[2];
  }
  int4* d_cand = ctx->cand.as<int4>();
  int* d_sums = reinterpret_cast<int*>(static_cast<char*>(ctx->cand.p) + cbytes);
  DLIOM_HIP_TRY(hipMemcpyAsync(d_cand, host, k * 16, hipMemcpyHostToDevice, ctx->stream));
  DLIOM_HIP_TRY(hipMemsetAsync(d_sums, 0, k * 4, ctx->stream));
  ScoreArgs a;
  a.level = m->levels[depth].view();
  a.cx = s.d_cx;
  a.cy = s.d_cy;
  a.cz = s.d_cz;
  a.n = s.n_hi;
  a.e = std::max(0, depth - m->options.full_resolution_depth + 1);
  a.start[0] = -s.linear_xy;
  a.start[1] = -s.linear_xy;
  a.start[2] = -s.linear_z;
  const unsigned chunks = static_cast<unsigned>((s.n_hi + kScoreChunk - 1) / kScoreChunk);
  hipLaunchKernelGGL(score_candidates_kernel, dim3(static_cast<unsigned>(k), chunks), dim3(256), 0, ctx->stream, a, d_cand,
                     static_cast<int>(k), d_sums);
  DLIOM_HIP_TRY(hipGetLastError());
  int* host_sums = host + 4 * k;
  DLIOM_HIP_TRY(hipMemcpyAsync(host_sums, d_sums, k * 4, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  std::memcpy(sums->data(), host_sums, k * 4);
  s.scored += static_cast<long long>(k);
  ++s.launches;
  return DLIOM_OK;
}

inline float to_probability(float value) {  // precomputation_grid_3d.h:31-34
  const float kMin = 0.1f, kMax = 1.f - 0.1f;
  return kMin + value * ((kMax - kMin) / 255.f);
}

// ScoreCandidates (:394-417) with the sums served from the cache (filled here when missing).
int score_candidates(Search& s, int depth, std::vector<Candidate>* candidates) {
  std::vector<Candidate> missing;
  for (const Candidate& c : *candidates)
    if (!s.cache.has(Search::key(depth, c.scan_index, c.offset))) missing.push_back(c);
  if (!missing.empty()) {
    std::vector<int> sums;
    DLIOM_TRY(device_sums(s, depth, missing, &sums));
    for (size_t i = 0; i < missing.size(); ++i) s.cache.put(Search::key(depth, missing[i].scan_index, missing[i].offset), sums[i]);
  }
  for (Candidate& c : *candidates) {
    const int sum = *s.cache.get(Search::key(depth, c.scan_index, c.offset));
    c.score = to_probability(sum / static_cast<float>(s.n_hi));
  }
  std::sort(candidates->begin(), candidates->end(), std::greater<Candidate>());
  return DLIOM_OK;
}

void children_of(const Search& s, const Candidate& c, int candidate_depth, std::vector<Candidate>* out) {  // :468-484
  const int half_width = 1 << (candidate_depth - 1);
  for (int z : {0, half_width}) {
    if (c.offset[2] + z > s.linear_z) break;
    for (int y : {0, half_width}) {
      if (c.offset[1] + y > s.linear_xy) break;
      for (int x : {0, half_width}) {
        if (c.offset[0] + x > s.linear_xy) break;
        Candidate h;
        h.scan_index = c.scan_index;
        h.offset[0] = c.offset[0] + x;
        h.offset[1] = c.offset[1] + y;
        h.offset[2] = c.offset[2] + z;
        out->push_back(h);
      }
    }
  }
}

// Batch prefetch: the children of every candidate of `siblings` from position `first` on whose
// score still beats `min_score` -- the group the recursion is about to walk through.
int prefetch_children(Search& s, const std::vector<Candidate>& siblings, size_t first, int candidate_depth,
                      float min_score) {
  std::vector<Candidate> batch;
  for (size_t i = first; i < siblings.size() && batch.size() < 4096; ++i) {
    if (siblings[i].score <= min_score) break;
    std::vector<Candidate> ch;
    children_of(s, siblings[i], candidate_depth, &ch);
    for (const Candidate& c : ch)
      if (!s.cache.has(Search::key(candidate_depth - 1, c.scan_index, c.offset))) batch.push_back(c);
  }
  if (batch.empty()) return DLIOM_OK;
  std::vector<int> sums;
  DLIOM_TRY(device_sums(s, candidate_depth - 1, batch, &sums));
  for (size_t i = 0; i < batch.size(); ++i) s.cache.put(Search::key(candidate_depth - 1, batch[i].scan_index, batch[i].offset), sums[i]);
  return DLIOM_OK;
}

// Wavefront prefetch: once a first match has raised the bound, every node the recursion can still
// expand has a score above `threshold` -- score the children of ALL such nodes level by level
// (one launch per level) instead of one launch per sibling group.
int prefetch_frontier(Search& s, float threshold) {
  const int top_depth = s.m->max_depth();
  std::vector<Candidate> frontier;
  for (const Candidate& c : *s.top)
    if (c.score > threshold) frontier.push_back(c);
  for (int depth = top_depth; depth >= 1 && !frontier.empty(); --depth) {
    std::vector<Candidate> children;
    for (const Candidate& c : frontier) children_of(s, c, depth, &children);
    if (children.size() > (1u << 18)) break;  // a flat score landscape: stay with on-demand batches
    std::vector<Candidate> missing;
    for (const Candidate& c : children)
      if (!s.cache.has(Search::key(depth - 1, c.scan_index, c.offset))) missing.push_back(c);
    if (!missing.empty()) {
      std::vector<int> sums;
      DLIOM_TRY(device_sums(s, depth - 1, missing, &sums));
      for (size_t i = 0; i < missing.size(); ++i)
        s.cache.put(Search::key(depth - 1, missing[i].scan_index, missing[i].offset), sums[i]);
    }
    frontier.clear();
    for (Candidate& c : children) {
      c.score = to_probability(*s.cache.get(Search::key(depth - 1, c.scan_index, c.offset)) / static_cast<float>(s.n_hi));
      if (c.score > threshold) frontier.push_back(c);
    }
  }
  s.frontier_threshold = threshold;
  return DLIOM_OK;
}

// The device frontier (kernels above): one chain of launches and one synchronisation that puts the lowest-resolution
// candidates and everything the recursion can reach below them into the cache.  Does nothing (DLIOM_OK) for searches
// it is not made for; the recursion then fetches scores on demand.
int device_frontier(Search& s, const std::vector<Candidate>& lowest, float min_score) {
  const dliom_fast_csm* m = s.m;
  dliom_ctx* ctx = s.ctx;
  const int max_depth = m->max_depth();
  constexpr int kCap = 8192;
  constexpr size_t kHead = 256, kUpload = 256 * 1024;
  const size_t k = lowest.size();
  // one wavefront per candidate: made for the clouds the reference matches (adaptive voxel filter, ~150-200 points);
  // with every return of a scan (65 536 points) the single workgroup of step 2 alone takes 7 ms -- those searches keep
  // the on-demand batches, whose block-per-chunk scoring kernel fills the chip
  constexpr int kMaxPoints = 8192;
  if (k == 0 || k > static_cast<size_t>(kCap) || max_depth + 1 > kMaxLevels || max_depth < 1 || ctx->pinned_bytes < (1u << 20) ||
      s.n_hi > kMaxPoints)
    return DLIOM_OK;
  const size_t pool_bytes = static_cast<size_t>(max_depth + 1) * kCap * sizeof(FrontierRec);
  DLIOM_TRY(ctx->cand.reserve(kHead + pool_bytes));
  int* d_counts = ctx->cand.as<int>();
  FrontierRec* d_pool = reinterpret_cast<FrontierRec*>(static_cast<char*>(ctx->cand.p) + kHead);
  int* h = static_cast<int*>(ctx->pinned);
  std::memset(h, 0, kHead);
  h[max_depth] = static_cast<int>(k);
  FrontierRec* h_top = reinterpret_cast<FrontierRec*>(static_cast<char*>(ctx->pinned) + kHead);
  for (size_t i = 0; i < k; ++i) h_top[i] = FrontierRec{lowest[i].scan_index, lowest[i].offset[0], lowest[i].offset[1], lowest[i].offset[2], 0};
  DLIOM_HIP_TRY(hipMemcpyAsync(d_counts, h, kHead, hipMemcpyHostToDevice, ctx->stream));
  DLIOM_HIP_TRY(hipMemcpyAsync(d_pool + static_cast<size_t>(max_depth) * kCap, h_top, k * sizeof(FrontierRec), hipMemcpyHostToDevice,
                               ctx->stream));
  FrontierArgs a;
  for (int d = 0; d < kMaxLevels; ++d) a.level[d] = d <= max_depth ? m->levels[d].view() : LevelView{nullptr, {0, 0, 0}, {0, 0, 0}};
  a.cx = s.d_cx;
  a.cy = s.d_cy;
  a.cz = s.d_cz;
  a.n = s.n_hi;
  a.max_depth = max_depth;
  a.full_resolution_depth = m->options.full_resolution_depth;
  a.linear_xy = s.linear_xy;
  a.linear_z = s.linear_z;
  a.pool = d_pool;
  a.counts = d_counts;
  a.cap = kCap;
  a.min_score = min_score;
  hipLaunchKernelGGL(frontier_top_kernel, dim3(static_cast<unsigned>((k + 3) / 4)), dim3(256), 0, ctx->stream, a, static_cast<int>(k));
  hipLaunchKernelGGL(frontier_greedy_kernel, dim3(1), dim3(512), 0, ctx->stream, a, static_cast<int>(k));
  for (int depth = max_depth; depth >= 1; --depth)
    hipLaunchKernelGGL(frontier_level_kernel, dim3(256), dim3(256), 0, ctx->stream, a, depth);
  int* out = reinterpret_cast<int*>(static_cast<char*>(ctx->pinned) + kUpload);
  const int out_records = static_cast<int>((ctx->pinned_bytes - kUpload - 8192 - (kMaxLevels + 2) * 4) / sizeof(FrontierRec));
  hipLaunchKernelGGL(frontier_pack_kernel, dim3(64), dim3(256), 0, ctx->stream, a, out, out_records);
  DLIOM_HIP_TRY(hipGetLastError());
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  const FrontierRec* rec = reinterpret_cast<const FrontierRec*>(out + kMaxLevels + 2);
  int at = 0;
  for (int depth = max_depth; depth >= 0 && at < out_records; --depth) {
    const int cnt = std::min(std::min(out[depth], kCap), out_records - at);
    for (int i = 0; i < cnt; ++i) {
      const FrontierRec& r = rec[at + i];
      const int o[3] = {r.ox, r.oy, r.oz};
      s.cache.put(Search::key(depth, r.scan, o), r.sum);
    }
    at += cnt;
  }
  s.scored += at;
  s.launches += max_depth + 3;
  float theta;
  std::memcpy(&theta, &out[kMaxLevels + 1], 4);
  s.frontier_threshold = theta;
  return DLIOM_OK;
}

PoseF pose_from_candidate(const Search& s, const Candidate& c) {  // :431-437
  const float r = s.m->resolution;
  const PoseF t{F3{r * static_cast<float>(c.offset[0]), r * static_cast<float>(c.offset[1]), r * static_cast<float>(c.offset[2])},
                QF{1.f, 0.f, 0.f, 0.f}};
  return pose_mul_f(t, s.scan_poses[c.scan_index]);
}

int low_resolution_score(Search& s, const PoseF& pose, float* score) {  // low_resolution_matcher.cc:23-36
  const float p7[7] = {pose.t.x, pose.t.y, pose.t.z, pose.q.w, pose.q.x, pose.q.y, pose.q.z};
  float sum = 0.f;
  DLIOM_TRY(sequential_probability_sums(s.ctx, *s.lo_cloud, s.m->lo_grid, p7, 1, &sum));
  *score = sum / static_cast<float>(s.lo_cloud->n);
  return DLIOM_OK;
}

// BranchAndBound (:439-492).  `status` carries device errors out of the recursion.
Candidate branch_and_bound(Search& s, const std::vector<Candidate>& candidates, int candidate_depth, float min_score,
                           int* status) {
  Candidate unsuccessful;
  unsuccessful.scan_index = 0;
  unsuccessful.offset[0] = unsuccessful.offset[1] = unsuccessful.offset[2] = 0;
  if (*status != DLIOM_OK) return unsuccessful;
  if (candidate_depth == 0) {
    for (const Candidate& c : candidates) {
      if (c.score <= min_score) return unsuccessful;
      float low = 0.f;
      *status = low_resolution_score(s, pose_from_candidate(s, c), &low);
      if (*status != DLIOM_OK) return unsuccessful;
      if (low >= s.m->options.min_low_resolution_score) {
        Candidate best = c;
        best.low_resolution_score = low;
        return best;
      }
    }
    return unsuccessful;
  }
  Candidate best = unsuccessful;
  best.score = min_score;
  for (size_t i = 0; i < candidates.size(); ++i) {
    const Candidate& c = candidates[i];
    if (c.score <= min_score) break;
    std::vector<Candidate> higher;
    children_of(s, c, candidate_depth, &higher);
    if (!higher.empty() && !s.cache.has(Search::key(candidate_depth - 1, higher[0].scan_index, higher[0].offset))) {
      if (best.score > s.initial_min_score && best.score < s.frontier_threshold) {
        *status = prefetch_frontier(s, best.score);  // a match exists: everything still reachable, per level
        if (*status != DLIOM_OK) return unsuccessful;
      }
      if (!s.cache.has(Search::key(candidate_depth - 1, higher[0].scan_index, higher[0].offset))) {
        *status = prefetch_children(s, candidates, i, candidate_depth, best.score);
        if (*status != DLIOM_OK) return unsuccessful;
      }
    }
    *status = score_candidates(s, candidate_depth - 1, &higher);
    if (*status != DLIOM_OK) return unsuccessful;
    best = std::max(best, branch_and_bound(s, higher, candidate_depth - 1, best.score, status));
    if (*status != DLIOM_OK) return unsuccessful;
  }
  return best;
}

int run_search(Search& s, const dliom_cloud& hi_cloud, float min_score, dliom_fast_csm_result* r) {
  const dliom_fast_csm* m = s.m;
  dliom_ctx* ctx = s.ctx;
  std::memset(r, 0, sizeof(*r));
  const int num_scans = static_cast<int>(s.scan_poses.size());
  r->num_discrete_scans = num_scans;
  if (num_scans == 0 || hi_cloud.n == 0) return DLIOM_OK;  // no candidates: nullptr in the reference
  s.n_hi = static_cast<int>(hi_cloud.n);
  // discrete scans on the device
  const size_t cells = static_cast<size_t>(num_scans) * s.n_hi;
  const size_t pose_bytes = (static_cast<size_t>(num_scans) * 28 + 255) & ~static_cast<size_t>(255);
  DLIOM_TRY(ctx->sums.reserve(pose_bytes + 3 * cells * 4));
  float* d_poses = ctx->sums.as<float>();
  s.d_cx = reinterpret_cast<int*>(static_cast<char*>(ctx->sums.p) + pose_bytes);
  s.d_cy = s.d_cx + cells;
  s.d_cz = s.d_cy + cells;
  std::vector<float> poses(7 * static_cast<size_t>(num_scans));
  for (int i = 0; i < num_scans; ++i) {
    const PoseF& p = s.scan_poses[i];
    const float v[7] = {p.t.x, p.t.y, p.t.z, p.q.w, p.q.x, p.q.y, p.q.z};
    std::memcpy(&poses[7 * i], v, sizeof(v));
  }
  // through the pinned block when they fit (its first 208 KB belong to device_frontier's upload): no synchronisation
  // here, the stream orders the kernels behind the copy
  constexpr size_t kPosesAt = 208 * 1024, kPosesMax = 48 * 1024;
  const bool poses_pinned = poses.size() * 4 <= kPosesMax && ctx->pinned_bytes >= (1u << 20);
  const float* poses_src = poses.data();
  if (poses_pinned) {
    std::memcpy(static_cast<char*>(ctx->pinned) + kPosesAt, poses.data(), poses.size() * 4);
    poses_src = reinterpret_cast<const float*>(static_cast<char*>(ctx->pinned) + kPosesAt);
  }
  DLIOM_HIP_TRY(hipMemcpyAsync(d_poses, poses_src, poses.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(discretize_kernel, dim3((s.n_hi + 255) / 256, num_scans), dim3(256), 0, ctx->stream, hi_cloud.d_x,
                     hi_cloud.d_y, hi_cloud.d_z, s.n_hi, d_poses, m->resolution, s.d_cx, s.d_cy, s.d_cz);
  DLIOM_HIP_TRY(hipGetLastError());
  if (!poses_pinned) DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));  // pageable source

  // lowest-resolution candidates (:358-392, 419-429)
  const int step = 1 << m->max_depth();
  {
    // the reference would allocate them all as well; refuse searches that cannot fit instead of
    // exhausting host memory (a whole-submap window with a shallow pyramid is ~1e8 per scan)
    const double per_axis_xy = std::floor((2.0 * s.linear_xy + step) / step), per_axis_z = std::floor((2.0 * s.linear_z + step) / step);
    if (per_axis_xy * per_axis_xy * per_axis_z * num_scans > 3.0e7) return DLIOM_ERR_CAPACITY;
    // the score cache packs (depth, scan, offsets) into 64 bits: 14-bit offsets, 16-bit scan index (Search::key)
    if (s.linear_xy > 8191 || s.linear_z > 8191 || num_scans > 65535) return DLIOM_ERR_CAPACITY;
  }
  std::vector<Candidate> lowest;
  for (int scan = 0; scan != num_scans; ++scan)
    for (int z = -s.linear_z; z <= s.linear_z; z += step)
      for (int y = -s.linear_xy; y <= s.linear_xy; y += step)
        for (int x = -s.linear_xy; x <= s.linear_xy; x += step) {
          Candidate c;
          c.scan_index = scan;
          c.offset[0] = x;
          c.offset[1] = y;
          c.offset[2] = z;
          lowest.push_back(c);
        }
  DLIOM_TRY(device_frontier(s, lowest, min_score));
  DLIOM_TRY(score_candidates(s, m->max_depth(), &lowest));
  s.top = &lowest;
  s.initial_min_score = min_score;
  int status = DLIOM_OK;
  const Candidate best = branch_and_bound(s, lowest, m->max_depth(), min_score, &status);
  DLIOM_TRY(status);
  r->num_scored_candidates = s.scored;
  r->num_score_launches = s.launches;
  if (best.score > min_score) {
    r->found = 1;
    r->score = best.score;
    const PoseF p = pose_from_candidate(s, best);
    const double out[7] = {p.t.x, p.t.y, p.t.z, p.q.w, p.q.x, p.q.y, p.q.z};
    std::memcpy(r->pose_estimate, out, sizeof(out));
    r->rotational_score = s.rotational_scores[best.scan_index];
    r->low_resolution_score = best.low_resolution_score;
  }
  return DLIOM_OK;
}

// GenerateDiscreteScans (:306-356): poses and rotational scores of the scans worth discretising.
void generate_discrete_scans(Search& s, const dliom_fast_csm_node_data& data, float max_norm, const PoseF& node,
                             const PoseF& submap) {
  const dliom_fast_csm* m = s.m;
  float max_scan_range = 3.f * m->resolution;
  max_scan_range = std::max(max_norm, max_scan_range);
  const float kSafetyMargin = 1.f - 1e-2f;
  const float res2 = m->resolution * m->resolution, range2 = max_scan_range * max_scan_range;
  const float angular_step_size = kSafetyMargin * std::acos(1.f - res2 / (2.f * range2));
  const int angular_window_size = static_cast<int>(std::lround(s.angular_window / angular_step_size));
  std::vector<float> angles;
  for (int rz = -angular_window_size; rz <= angular_window_size; ++rz) angles.push_back(rz * angular_step_size);
  const PoseF node_to_submap = pose_mul_f(pose_inverse_f(submap), node);
  // gravity_alignment.inverse() in double, then cast<float>()
  const double* g = data.gravity_alignment;
  const double n2 = (g[1] * g[1] + g[3] * g[3]) + (g[2] * g[2] + g[0] * g[0]);
  QF gi{0.f, 0.f, 0.f, 0.f};
  if (n2 > 0.) gi = QF{static_cast<float>(g[0] / n2), static_cast<float>(-g[1] / n2), static_cast<float>(-g[2] / n2),
                       static_cast<float>(-g[3] / n2)};
  const float initial_angle = get_yaw(qmul(node_to_submap.q, gi));
  const Histogram scan_histogram(data.rotational_scan_matcher_histogram,
                                 data.rotational_scan_matcher_histogram + m->submap_histogram.size());
  const QF submap_rotation_inverse = quat_inverse(submap.q);
  for (size_t i = 0; i != angles.size(); ++i) {
    const float score = match_histograms(m->submap_histogram, rotate_histogram(scan_histogram, initial_angle + angles[i]));
    if (score < m->options.min_rotational_score) continue;
    const QF yaw = angle_axis_to_quaternion(F3{0.f, 0.f, angles[i]});
    s.scan_poses.push_back(PoseF{node_to_submap.t, qmul(qmul(submap_rotation_inverse, yaw), node.q)});
    s.rotational_scores.push_back(score);
  }
}

PoseF to_pose_f(const double* p) {  // Rigid3d::cast<float>()
  return PoseF{F3{static_cast<float>(p[0]), static_cast<float>(p[1]), static_cast<float>(p[2])},
               QF{static_cast<float>(p[3]), static_cast<float>(p[4]), static_cast<float>(p[5]), static_cast<float>(p[6])}};
}

struct StagedClouds {
  dliom_cloud* hi = nullptr;
  dliom_cloud* lo = nullptr;
  ~StagedClouds() {
    if (hi) dliom_cloud_destroy(hi);
    if (lo) dliom_cloud_destroy(lo);
  }
};

int stage(dliom_ctx* ctx, const dliom_fast_csm_node_data* data, StagedClouds* c) {
  if (data == nullptr || data->num_high_resolution_points < 0 || data->num_low_resolution_points <= 0 ||
      data->rotational_scan_matcher_histogram == nullptr ||
      (data->num_high_resolution_points > 0 && data->high_resolution_points == nullptr) ||
      data->low_resolution_points == nullptr)
    return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_TRY(dliom_cloud_create(ctx, data->high_resolution_points, data->num_high_resolution_points, &c->hi));
  DLIOM_TRY(dliom_cloud_create(ctx, data->low_resolution_points, data->num_low_resolution_points, &c->lo));
  return DLIOM_OK;
}

}  // namespace

extern "C" {

int dliom_fast_csm_create(dliom_ctx* ctx, const dliom_grid* hi, const dliom_grid* lo, const float* node_histograms,
                          const float* node_angles, int num_nodes, int histogram_size,
                          const dliom_fast_csm_options* o, dliom_fast_csm** out) {
  if (ctx == nullptr || hi == nullptr || lo == nullptr || o == nullptr || out == nullptr || num_nodes <= 0 ||
      histogram_size <= 0 || node_histograms == nullptr || node_angles == nullptr)
    return DLIOM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (o->branch_and_bound_depth < 1 || o->full_resolution_depth < 1 || o->branch_and_bound_depth > 15)
    return DLIOM_ERR_INVALID_ARGUMENT;  // CHECK_GE(depth, 1) (:60-61)
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  std::unique_ptr<dliom_fast_csm> m(new dliom_fast_csm);
  m->ctx = ctx;
  m->options = *o;
  m->resolution = hi->resolution;
  m->width_in_voxels = 64 << hi->bits;
  m->lo_grid = lo;
  // RotationalScanMatcher ctor (:174-182)
  m->submap_histogram.assign(histogram_size, 0.f);
  for (int k = 0; k < num_nodes; ++k) {
    const Histogram h(node_histograms + static_cast<size_t>(k) * histogram_size,
                      node_histograms + static_cast<size_t>(k + 1) * histogram_size);
    const Histogram r = rotate_histogram(h, node_angles[k]);
    for (int i = 0; i < histogram_size; ++i) m->submap_histogram[i] += r[i];
  }
  // level 0 over the bounding box of the leaves
  int64_t count = 0;
  DLIOM_TRY(const_cast<dliom_grid*>(hi)->refresh_count(&count));
  m->levels.resize(o->branch_and_bound_depth);
  if (count > 1) {
    DLIOM_TRY(ctx->misc.reserve(32768 + 64));
    int* d_mm = reinterpret_cast<int*>(static_cast<char*>(ctx->misc.p) + 32768);
    const int init[6] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, INT32_MIN};
    DLIOM_HIP_TRY(hipMemcpyAsync(d_mm, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(leaf_bbox_kernel, dim3(static_cast<unsigned>((count + 255) / 256)), dim3(256), 0, ctx->stream,
                       hi->d_slot_coord, static_cast<unsigned>(count), d_mm);
    int mm[6];
    DLIOM_HIP_TRY(hipMemcpyAsync(mm, d_mm, sizeof(mm), hipMemcpyDeviceToHost, ctx->stream));
    DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    // ConvertToPrecomputationGrid's value map (:53-57), host float arithmetic like the reference
    std::vector<uint8_t> lut(32768, 0);
    const float kMin = 0.1f, kMax = 1.f - 0.1f;
    for (int v = 1; v < 32768; ++v) {
      const float kScale = (kMax - kMin) / 32766.f;
      const float p = v * kScale + (kMin - kScale);
      const long cell = std::lround((p - kMin) * (255.f / (kMax - kMin)));
      lut[v] = static_cast<uint8_t>(std::min(255l, std::max(0l, cell)));
    }
    uint8_t* d_lut = static_cast<uint8_t*>(ctx->misc.p);
    DLIOM_HIP_TRY(hipMemcpyAsync(d_lut, lut.data(), 32768, hipMemcpyHostToDevice, ctx->stream));
    dliom_fast_csm::Level& l0 = m->levels[0];
    for (int a = 0; a < 3; ++a) {
      l0.lo[a] = mm[a] * 8;
      l0.n[a] = (mm[3 + a] - mm[a] + 1) * 8;
    }
    DLIOM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&l0.d), l0.cells()));
    DLIOM_HIP_TRY(hipMemsetAsync(l0.d, 0, l0.cells(), ctx->stream));
    hipLaunchKernelGGL(convert_level0_kernel, dim3(static_cast<unsigned>(count - 1)), dim3(256), 0, ctx->stream,
                       hi->d_slot_coord, hi->d_pool, d_lut, l0.d, l0.lo[0], l0.lo[1], l0.lo[2], l0.n[0], l0.n[1]);
    DLIOM_HIP_TRY(hipGetLastError());
    int last_width = 1;
    for (int depth = 1; depth != o->branch_and_bound_depth; ++depth) {  // :65-76
      const bool half = depth >= o->full_resolution_depth;
      const int next_width = 1 << depth;
      const int per_high = 1 << std::max(0, depth - o->full_resolution_depth);
      const int shift = (next_width - last_width + (per_high - 1)) / per_high;
      const dliom_fast_csm::Level& src = m->levels[depth - 1];
      dliom_fast_csm::Level& dst = m->levels[depth];
      for (int a = 0; a < 3; ++a) {
        const int lo_c = src.lo[a] - shift, hi_c = src.lo[a] + src.n[a] - 1;
        dst.lo[a] = half ? (lo_c >> 1) : lo_c;
        dst.n[a] = (half ? (hi_c >> 1) : hi_c) - dst.lo[a] + 1;
      }
      DLIOM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dst.d), dst.cells()));
      hipLaunchKernelGGL(precompute_level_kernel, dim3(static_cast<unsigned>((dst.cells() + 255) / 256)), dim3(256), 0,
                         ctx->stream, src.view(), dst.d, dst.lo[0], dst.lo[1], dst.lo[2], dst.n[0], dst.n[1], dst.n[2],
                         shift, half ? 1 : 0);
      DLIOM_HIP_TRY(hipGetLastError());
      last_width = next_width;
    }
    DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  *out = m.release();
  return DLIOM_OK;
}

int dliom_fast_csm_destroy(dliom_fast_csm* m) {
  if (m == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  (void)hipDeviceSynchronize();
  delete m;
  return DLIOM_OK;
}

int dliom_fast_csm_level(const dliom_fast_csm* m, int depth, int32_t lo[3], int32_t dims[3], uint8_t* values,
                         int64_t capacity) {
  if (m == nullptr || lo == nullptr || dims == nullptr || depth < 0 || depth > m->max_depth())
    return DLIOM_ERR_INVALID_ARGUMENT;
  const dliom_fast_csm::Level& l = m->levels[depth];
  for (int a = 0; a < 3; ++a) {
    lo[a] = l.lo[a];
    dims[a] = l.n[a];
  }
  if (values != nullptr) {
    if (capacity < static_cast<int64_t>(l.cells())) return DLIOM_ERR_CAPACITY;
    if (l.cells() > 0) DLIOM_HIP_TRY(hipMemcpy(values, l.d, l.cells(), hipMemcpyDeviceToHost));
  }
  return DLIOM_OK;
}

int dliom_fast_csm_match(dliom_ctx* ctx, const dliom_fast_csm* m, const double global_node_pose[7], const double global_submap_pose[7],
                         const dliom_fast_csm_node_data* data, float min_score, dliom_fast_csm_result* result) {
  if (ctx == nullptr || m == nullptr || global_node_pose == nullptr || global_submap_pose == nullptr || result == nullptr)
    return DLIOM_ERR_INVALID_ARGUMENT;
  if (ctx->device != m->ctx->device) return DLIOM_ERR_INVALID_ARGUMENT;  // the pyramid lives on the creating device
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  StagedClouds c;
  DLIOM_TRY(stage(ctx, data, &c));
  Search s;
  s.m = m;
  s.ctx = ctx;
  s.linear_xy = static_cast<int>(std::lround(m->options.linear_xy_search_window / m->resolution));  // :154-156
  s.linear_z = static_cast<int>(std::lround(m->options.linear_z_search_window / m->resolution));
  s.angular_window = m->options.angular_search_window;
  s.lo_cloud = c.lo;
  generate_discrete_scans(s, *data, c.hi->max_norm, to_pose_f(global_node_pose), to_pose_f(global_submap_pose));
  return run_search(s, *c.hi, min_score, result);
}

int dliom_fast_csm_match_full_submap(dliom_ctx* ctx, const dliom_fast_csm* m, const double global_node_rotation[4],
                                     const double global_submap_rotation[4], const dliom_fast_csm_node_data* data,
                                     float min_score, dliom_fast_csm_result* result) {
  if (ctx == nullptr || m == nullptr || global_node_rotation == nullptr || global_submap_rotation == nullptr || result == nullptr)
    return DLIOM_ERR_INVALID_ARGUMENT;
  if (ctx->device != m->ctx->device) return DLIOM_ERR_INVALID_ARGUMENT;  // the pyramid lives on the creating device
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  StagedClouds c;
  DLIOM_TRY(stage(ctx, data, &c));
  Search s;
  s.m = m;
  s.ctx = ctx;
  // :209-216
  const int w = (m->width_in_voxels + 1) / 2 + static_cast<int>(std::lround(c.hi->max_norm / m->resolution + 0.5f));
  s.linear_xy = w;
  s.linear_z = w;
  s.angular_window = M_PI;
  s.lo_cloud = c.lo;
  const double node[7] = {0, 0, 0, global_node_rotation[0], global_node_rotation[1], global_node_rotation[2],
                          global_node_rotation[3]};
  const double submap[7] = {0, 0, 0, global_submap_rotation[0], global_submap_rotation[1], global_submap_rotation[2],
                            global_submap_rotation[3]};
  generate_discrete_scans(s, *data, c.hi->max_norm, to_pose_f(node), to_pose_f(submap));
  return run_search(s, *c.hi, min_score, result);
}

int dliom_fast_csm_match_with_3dof_initial(dliom_ctx* ctx, const dliom_fast_csm* m, const double pose_in_submap_guess[7],
                                           const dliom_fast_csm_node_data* data, float min_score,
                                           dliom_fast_csm_result* result) {
  if (ctx == nullptr || m == nullptr || pose_in_submap_guess == nullptr || result == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  if (ctx->device != m->ctx->device) return DLIOM_ERR_INVALID_ARGUMENT;  // the pyramid lives on the creating device
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  StagedClouds c;
  DLIOM_TRY(stage(ctx, data, &c));
  Search s;
  s.m = m;
  s.ctx = ctx;
  s.linear_xy = static_cast<int>(std::lround(m->options.linear_xy_search_window / m->resolution));
  s.linear_z = static_cast<int>(std::lround(m->options.linear_z_search_window / m->resolution));
  s.angular_window = m->options.angular_search_window;
  s.lo_cloud = c.lo;
  s.scan_poses.push_back(to_pose_f(pose_in_submap_guess));  // :181-184
  s.rotational_scores.push_back(static_cast<float>(m->options.min_rotational_score + 0.01));
  return run_search(s, *c.hi, min_score, result);
}

}  // extern "C"
