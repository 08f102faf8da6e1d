// RotationalScanMatcher::ComputeHistogram on the device
// (mapping/internal/3d/scan_matching/rotational_scan_matcher.cc:29-123,159-170), the per-scan O(N) step that
// LocalTrajectoryBuilder3D runs after insertion (local_trajectory_builder_3d.cc:605-610) on
//   TransformPointCloud(filtered_range_data_in_tracking.returns, Rigid3f::Rotation(gravity_alignment.cast<float>()))
// -- the filtered cloud is already in HBM; the host version (rotational_histogram.cc) cost more than the whole
// device chain it followed.
//
// What makes it order dependent, and how each dependence is kept (bit for bit):
//   * slices: key lround(z / 0.2f), points in INPUT order inside a slice (std::map of vectors, :162-165).  Kernel 1
//     counts the keys (4096 bins, |z| < 409 m); kernel 2 runs one workgroup per non-empty slice, whose 16 waves each
//     compact their contiguous 1/16 of the input (count, prefix over the waves, write): input order, no atomics.
//   * ComputeCentroid (:52-59): a SEQUENTIAL float sum.  One thread per coordinate walks the slice in LDS.
//   * SortSlice (:97-121): atan2f of the reference's libm (glibc 2.35 flt-32 = fdlibm, no FMA; restated below and
//     pinned against the host's atan2f by tests/test_cpu_host.py), then std::sort by angle ONLY.  Two returns of a slice
//     share an angle in three scans out of four, and which of them comes first decides `last_point` below: the order
//     libstdc++'s std::sort leaves equal keys in is part of the result.  A slice is first sorted by (angle, position)
//     with a bitonic sort; if two equal angles end up next to each other, introsort itself is replayed on the slice
//     (wave_sort_arrangement: median-of-three + unguarded partition of every segment above 16 elements as
//     data-parallel rounds, heap sort where the depth limit 2 lg n is used up -- one slice in four, input order being
//     close to a worst case of the median-of-three -- and a stable sort for the final insertion sort).
//   * AddPointCloudSliceToHistogram (:61-92): `last_point` only moves when a point is more than 0.9 m from it -- a
//     sequential state machine.  One wave evaluates 64 sorted points against the current anchor at once, takes the
//     lanes before the first jump, moves the anchor and goes on.
//   * histogram(bucket) += value (:49): float additions in slice order, then point order.  The slices' contribution
//     lists lie back to back in slice order; kernel 3: one wave per bucket queues the positions of its entries in
//     order, fetches the values and one thread adds them one after the other.
// Slices of more than 4096 points (the floor of every real scan) take the path of rothist_big.h: the same steps on arrays
// in HBM, a radix sort, std::sort's order of equal angles restricted to the segments that hold ties, and the `last_point`
// chain by pointer doubling.  The sequential float sums (centroids, bucket additions) are replayed in parallel, exactly
// (exact_sum.h).  Whether a cloud has such slices is only known on the device: the context remembers what the previous
// cloud needed and enqueues the big path (or not) accordingly; a cloud that needed it without having it is run again.
// Limits (DLIOM_ERR_CAPACITY, the host entry point has none): |z| < 409.6 m, at most 63 slices above 4096 points.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include <hipcub/hipcub.hpp>

#include "device_common.h"
#include "exact_sum.h"

namespace dliom {
namespace rothist {

constexpr int kBins = 4096;
constexpr int kBinOrigin = 2048;
constexpr int kMaxSlice = 4096;
constexpr int kThreads = 1024;
constexpr int kExactSumFrom = 2048;  // bucket additions: this many and more are replayed in parallel (exact_sum.h)
constexpr float kMinDistance = 0.2f;
constexpr float kMaxDistance = 0.9f;
constexpr float kSliceHeight = 0.2f;

// ---- glibc 2.35 sysdeps/ieee754/flt-32/{s_atanf.c, e_atan2f.c} (fdlibm), the algorithm the reference's
// common::atan2 -> std::atan2(float, float) runs on the host; every operation rounded to float, no contraction.
__device__ __forceinline__ float fd_atanf(float x) {
  const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
  const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
  const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
                        9.0908870101e-02f, -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
                        4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
  const int hx = __float_as_int(x);
  const int ix = hx & 0x7fffffff;
  int id;
  if (ix >= 0x4c000000) {  // |x| >= 2^25
    if (ix > 0x7f800000) return x + x;
    return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3ee00000) {  // |x| < 0.4375
    if (ix < 0x31000000) return x;
    id = -1;
  } else {
    x = fabsf(x);
    if (ix < 0x3f980000) {
      if (ix < 0x3f300000) {
        id = 0;
        x = (2.0f * x - 1.0f) / (2.0f + x);
      } else {
        id = 1;
        x = (x - 1.0f) / (x + 1.0f);
      }
    } else {
      if (ix < 0x401c0000) {
        id = 2;
        x = (x - 1.5f) / (1.0f + 1.5f * x);
      } else {
        id = 3;
        x = -1.0f / x;
      }
    }
  }
  const float z = x * x;
  const float w = z * z;
  const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
  return hx < 0 ? -r : r;
}
__device__ __forceinline__ float fd_atan2f(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  const int hx = __float_as_int(x), hy = __float_as_int(y);
  const int ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
  if (hx == 0x3f800000) return fd_atanf(y);
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
  if (iy == 0) {
    if (m < 2) return y;
    return m == 2 ? pi + tiny : -pi - tiny;
  }
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    const float pi_o_4 = 7.8539818525e-01f;
    if (iy == 0x7f800000) {
      if (m == 0) return pi_o_4 + tiny;
      if (m == 1) return -pi_o_4 - tiny;
      if (m == 2) return 3.0f * pi_o_4 + tiny;
      return -3.0f * pi_o_4 - tiny;
    }
    if (m == 0) return 0.0f;
    if (m == 1) return -0.0f;
    if (m == 2) return pi + tiny;
    return -pi - tiny;
  }
  if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int k = (iy - ix) >> 23;
  float z;
  if (k > 60)
    z = pi_o_2 + 0.5f * pi_lo;
  else if (hx < 0 && k < -60)
    z = 0.0f;
  else
    z = fd_atanf(fabsf(y / x));
  if (m == 0) return z;
  if (m == 1) return -z;
  if (m == 2) return pi - (z - pi_lo);
  return (z - pi_lo) - pi;
}

__device__ __forceinline__ float norm2(float x, float y) { return sqrtf(x * x + y * y); }

// AddValueToHistogram's bucket (:35-48)
__device__ __forceinline__ int bucket_of(float angle, int size) {
  const float pi = 3.14159274101257324f;  // static_cast<float>(M_PI)
  while (angle > pi) angle -= pi;
  while (angle < 0.f) angle += pi;
  const float zero_to_one = angle / pi;
  const int bucket = lround_away(static_cast<float>(size) * zero_to_one - 0.5f);
  return min(max(bucket, 0), size - 1);
}

// ---- kernel 1: rotated points, slice keys, key counts ------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void prepare_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ z, int n, Quat4 q, int rotate,
                                                           float* __restrict__ rx, float* __restrict__ ry,
                                                           float* __restrict__ rz, short* __restrict__ keys,
                                                           unsigned* __restrict__ bin_counts, unsigned* __restrict__ flags) {
  __shared__ unsigned hist[kBins];
  for (int b = threadIdx.x; b < kBins; b += kThreads) hist[b] = 0u;
  __syncthreads();
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i < n) {
    float px = x[i], py = y[i], pz = z[i];
    if (rotate) {
      // Rigid3f::Rotation(q) * point = rotation * point + translation with a zero translation (rigid_transform.h:214-219)
      float ox, oy, oz;
      rotate_point(q, px, py, pz, ox, oy, oz);
      px = ox + 0.f;
      py = oy + 0.f;
      pz = oz + 0.f;
    }
    rx[i] = px;
    ry[i] = py;
    rz[i] = pz;
    const float kf = pz / kSliceHeight;
    int key = 0x7fff;  // no slice has this key (the padding's): a rejected point is in nobody's compaction
    if (!(fabsf(kf) < 2047.f) || !(fabsf(px) < 3.0e38f) || !(fabsf(py) < 3.0e38f)) {  // also NaN / inf
      atomicOr(flags, 1u);
    } else {
      key = lround_away(kf);
      atomicAdd(&hist[key + kBinOrigin], 1u);
    }
    keys[i] = static_cast<short>(key);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kBins; b += kThreads)
    if (hist[b] != 0u) atomicAdd(&bin_counts[b], hist[b]);
}

// ---- kernel 2: one workgroup per non-empty slice ------------------------------------------------------------------
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned* wave_sums, unsigned* total) {
  // 1024 threads: inclusive scan inside the wave, then over the 16 waves
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // inclusive scan over the wave by DPP moves (row_shr inside the rows of 16 lanes, then row_bcast:15 / :31; a lane
  // without a source adds 0) -- six instructions a step fewer than __shfl_up's ds_bpermute and no LDS queue
  unsigned incl = v;
#define DLIOM_SCAN_STEP(ctrl, mask) \
  incl += static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(incl), ctrl, mask, 0xf, false))
  DLIOM_SCAN_STEP(0x111, 0xf);  // row_shr:1
  DLIOM_SCAN_STEP(0x112, 0xf);  // row_shr:2
  DLIOM_SCAN_STEP(0x114, 0xf);  // row_shr:4
  DLIOM_SCAN_STEP(0x118, 0xf);  // row_shr:8
  DLIOM_SCAN_STEP(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
  DLIOM_SCAN_STEP(0x143, 0xc);  // row_bcast:31 into rows 2 and 3
#undef DLIOM_SCAN_STEP
  __syncthreads();  // wave_sums may still be read from an earlier call
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  unsigned before = 0u, all = 0u;
  for (int w = 0; w < kThreads / 64; ++w) {
    const unsigned s = wave_sums[w];
    if (w < wave) before += s;
    all += s;
  }
  *total = all;
  return before + incl - v;
}

// sum = ((acc + a[0]) + a[1]) + ... in exactly that order, by ONE thread: 64 values per step (sixteen 16-byte LDS
// reads in flight), then 64 dependent additions.  `a` is 16-byte aligned; entries past n up to the next multiple of 64
// are read and must be readable (they are replaced by +0, which changes nothing).
__device__ __forceinline__ float thread_sequential_sum(const float* a, int n, float acc) {
  for (int i0 = 0; i0 < n; i0 += 64) {
    float4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const float4*>(a + i0 + 4 * k);
    const int left = n - i0;
    if (left >= 64) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        acc += v[k].x;
        acc += v[k].y;
        acc += v[k].z;
        acc += v[k].w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        acc += 4 * k < left ? v[k].x : 0.f;
        acc += 4 * k + 1 < left ? v[k].y : 0.f;
        acc += 4 * k + 2 < left ? v[k].z : 0.f;
        acc += 4 * k + 3 < left ? v[k].w : 0.f;
      }
    }
  }
  return acc;
}

// float -> unsigned with the same order as operator< on floats (NaN excluded), -0 folded onto +0 first
__device__ __forceinline__ unsigned ordered_bits(float f) {
  if (f == 0.f) f = 0.f;
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}


// ---- std::sort's order of EQUAL keys ---------------------------------------------------------------------------------
// SortSlice sorts (angle, point) pairs by angle only (rotational_scan_matcher.cc:97-121); two returns of one slice share
// an angle in three scans out of four (on walls, where two returns of one 0.2 m slice stand above each other, in every
// slice), and the order libstdc++'s introsort leaves them in decides which one becomes `last_point`.  So the algorithm is
// reproduced, not just a sorted order (tests/cpp/std_sort_model.cc is the formulation in plain C++, checked against the
// real std::sort on arrays full of ties):
//   __introsort_loop      = median-of-three + unguarded partition of every segment above 16 elements; a partition as
//                           data-parallel steps -- the k-th stop of the left pointer swaps with the k-th stop of the
//                           right pointer while they have not crossed; heap sort where the depth limit 2 lg n is used up;
//   __final_insertion_sort = a STABLE sort of what the partitions left (insertion never moves an element past an equal one).
// Only segments that still hold two TIED elements (elements whose key occurs more than once) are partitioned: equal keys can
// only be told apart by where the partitions put them, a segment without two of them has a unique sorted order, and the
// final sort is stable -- what is left of the other segments does not matter.
// Round 4: a segment is partitioned by ONE WAVE (ballots and lane ranks, no workgroup barrier), the segments wait in a
// queue in LDS that the workgroup's 16 waves serve.  The block-wide rounds this replaces cost ~30 000 cycles each (fifteen
// barriers) however little was left to do: 100 us for the one tied pair of a wall slice, 230 us on slices full of ties.
// `a`: items (ordered angle bits << 32 | position in the slice).
struct SortScratch {
  unsigned short *tmp_l, *tmp_r;  // [kMaxSlice + 8] each: the stops of the two pointers of the partition in progress
  const unsigned char* tied;      // by position in the slice (the items' low 16 bits)
  struct Queue* queue;
};
// A ring: the queue is served level by level, the segments of one level are disjoint and every one has > 16 elements, so
// the entries alive at any time (the rest of this level + the children appended so far) are <= 2 m / 17.  The segments
// EVER queued are not: a path of lopsided partitions queues one per level (round 6's soak: 1 277 for 9 716 descending
// keys in tied pairs against the 1 207 = 2 m / 17 + 64 this used to hold in all).
constexpr int kQueueCap = 1024;
struct Queue {
  unsigned reserved, capacity, level_begin, overflow;  // capacity: entries of seg[] in use (kQueueCap unless the owner has more room)
  uint2 seg[kQueueCap];  // x = first | last << 16, y = partitions left on this path
};

// exclusive prefix counts of 4 consecutive flags per thread (position 4 t + k) over the workgroup; out[p] for p in
// [0, 4096], out[4096] = total
__device__ __forceinline__ void blocked_prefix(const unsigned (&flag)[4], unsigned short* out, unsigned* wave_sums) {
  const unsigned mine = flag[0] + flag[1] + flag[2] + flag[3];
  unsigned total;
  const unsigned base = block_exclusive_scan(mine, wave_sums, &total);
  const int p0 = 4 * static_cast<int>(threadIdx.x);
  unsigned run = base;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    out[p0 + k] = static_cast<unsigned short>(run);
    run += flag[k];
  }
  if (threadIdx.x == kThreads - 1) out[kMaxSlice] = static_cast<unsigned short>(total);
}

// An item of the replay: (key, position in the slice), compared by key only.  64 bits (ordered angle bits << 32 | position)
// for the slices that fit LDS anyway; 32 bits (dense rank of the angle << 16 | position) for the floor slices of
// rothist_big.h, which fit LDS only that way.
__device__ __forceinline__ unsigned item_key(unsigned long long x) { return static_cast<unsigned>(x >> 32); }
__device__ __forceinline__ unsigned item_id(unsigned long long x) { return static_cast<unsigned>(x) & 0xffffu; }
__device__ __forceinline__ unsigned item_key(unsigned x) { return x >> 16; }
__device__ __forceinline__ unsigned item_id(unsigned x) { return x & 0xffffu; }

// bits/stl_heap.h: __adjust_heap (with its __push_heap), __make_heap, __sort_heap on items compared by their keys
template <class Item>
__device__ inline void adjust_heap_keys(Item* first, int hole, int len, Item value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (item_key(first[child]) < item_key(first[child - 1])) --child;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && item_key(first[parent]) < item_key(value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}
template <class Item>
__device__ inline void heap_sort_keys(Item* first, int len) {
  if (len >= 2)
    for (int parent = (len - 2) / 2;; --parent) {
      adjust_heap_keys(first, parent, len, first[parent]);
      if (parent == 0) break;
    }
  for (int last = len; last > 1;) {
    --last;
    const Item value = first[last];
    first[last] = first[0];
    adjust_heap_keys(first, 0, last, value);
  }
}

// The queue is served level by level: the segments queued so far are partitioned, one wave each, their children are
// appended (one atomic counter), a barrier, and the appended ones are the next level.  (A first version let idle waves
// spin on their next entry instead of meeting at a barrier; it hung on the device and was not worth the minutes.)
__device__ __forceinline__ void queue_init(Queue* q, unsigned capacity = kQueueCap) {  // all threads; a barrier must follow
  if (threadIdx.x == 0) {
    q->reserved = 0u;
    q->capacity = capacity;
    q->level_begin = 0u;
    q->overflow = 0u;
  }
}
__device__ __forceinline__ void queue_push(Queue* q, int first, int last, int depth) {  // one lane
  const unsigned slot = atomicAdd(&q->reserved, 1u);
  if (slot - q->level_begin < q->capacity)
    q->seg[slot % q->capacity] = make_uint2(static_cast<unsigned>(first) | (static_cast<unsigned>(last) << 16), static_cast<unsigned>(depth));
  else
    atomicExch(&q->overflow, 1u);
}

// std::__introsort_loop on the queued segments (see above).  All threads of the workgroup call this after the queue has
// been filled and a barrier; returns after a barrier, false if the ring overflowed (cannot happen with a capacity of
// 2 m / 17 or more).
template <class Item>
__device__ bool wave_sort_arrangement(Item* a, const SortScratch& sc) {
  Queue* q = sc.queue;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned level_begin = 0u;
  for (int level = 0; level < 4 * 64; ++level) {  // (a path has at most 2 lg n partitions; the bound only guards the loop)
    const unsigned level_end = q->reserved;
    const bool overflowed = q->overflow != 0u;  // (read here, where nobody appends: the same answer in every thread)
    if (threadIdx.x == 0) q->level_begin = level_begin;
    __syncthreads();  // everybody has read the level's end before anybody appends to the queue
    if (level_begin >= level_end || overflowed) break;
    for (unsigned e = level_begin + static_cast<unsigned>(wave); e < level_end; e += kThreads / 64) {
    const uint2 entry = q->seg[e % q->capacity];
    const int first = static_cast<int>(entry.x & 0xffffu), last = static_cast<int>(entry.x >> 16);
    const int depth = static_cast<int>(entry.y);
    if (depth == 0) {
      // std::sort's depth limit (2 lg n partitions on one path): it heap-sorts what is left of such a segment --
      // std::__partial_sort(first, last, last) = __make_heap + __sort_heap, restated; sequential
      if (lane == 0) heap_sort_keys(a + first, last - first);
      continue;
    }
    // (a) __move_median_to_first(first, first + 1, mid, last - 1)
    if (lane == 0) {
      const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
      const unsigned ka = item_key(a[ia]), kb = item_key(a[ib]), kc = item_key(a[ic]);
      int md;
      if (ka < kb) {
        if (kb < kc) md = ib;
        else if (ka < kc) md = ic;
        else md = ia;
      } else if (ka < kc) md = ia;
      else if (kb < kc) md = ic;
      else md = ib;
      const Item t = a[first];
      a[first] = a[md];
      a[md] = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned pivot = item_key(a[first]);
    // (b) where the two pointers of __unguarded_partition(first + 1, last, first) stop: !(x < pivot) from the left,
    //     !(pivot < x) from the right; both lists in ascending order of position
    unsigned short* stops_l = sc.tmp_l + first + 1;
    unsigned short* stops_r = sc.tmp_r + first + 1;
    int cnt_l = 0, cnt_r = 0;
    for (int base = first + 1; base < last; base += 64) {
      const int p = base + lane;
      const bool in = p < last;
      const unsigned x = in ? item_key(a[p]) : 0u;
      const bool ge = in && !(x < pivot), le = in && !(pivot < x);
      const unsigned long long ml = __builtin_amdgcn_ballot_w64(ge), mr = __builtin_amdgcn_ballot_w64(le);
      if (ge) stops_l[cnt_l + __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(ml >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(ml), 0u))] =
          static_cast<unsigned short>(p);
      if (le) stops_r[cnt_r + __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mr >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mr), 0u))] =
          static_cast<unsigned short>(p);
      cnt_l += __builtin_popcountll(ml);
      cnt_r += __builtin_popcountll(mr);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // (c) the k-th stop from the left swaps with the k-th stop from the right while the pointers have not crossed
    //     (positions from the left grow with k, from the right they fall: the valid k are 0 .. K - 1)
    const int lim = min(cnt_l, cnt_r);
    int K = lim;
    for (int k0 = 0; k0 < lim; k0 += 64) {
      const int k = k0 + lane;
      const bool v = k < lim && stops_l[k] < stops_r[cnt_r - 1 - k];
      const unsigned long long mv = __builtin_amdgcn_ballot_w64(v);
      if (mv != ~0ull) {
        K = k0 + __builtin_ctzll(~mv);
        break;
      }
    }
    for (int k0 = 0; k0 < K; k0 += 64) {
      const int k = k0 + lane;
      if (k < K) {
        const int il = stops_l[k], ir = stops_r[cnt_r - 1 - k];
        const Item xl = a[il], xr = a[ir];
        a[il] = xr;
        a[ir] = xl;
      }
    }
    int cut = 1 << 30;  // where the left pointer stops next
    if (K < cnt_l) cut = stops_l[K];
    if (K > 0) cut = min(cut, static_cast<int>(stops_r[cnt_r - K]));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // (d) [first, cut) and [cut, last): queued if they are above the threshold and hold two tied elements
    int tied_l = 0, tied_r = 0;
    for (int base = first; base < last; base += 64) {
      const int p = base + lane;
      const bool t = p < last && sc.tied[item_id(a[p])] != 0;
      tied_l += __builtin_popcountll(__builtin_amdgcn_ballot_w64(t && p < cut));
      tied_r += __builtin_popcountll(__builtin_amdgcn_ballot_w64(t && p >= cut));
    }
    if (lane == 0) {
      if (cut - first > 16 && tied_l >= 2) queue_push(q, first, cut, depth - 1);
      if (last - cut > 16 && tied_r >= 2) queue_push(q, cut, last, depth - 1);
    }
    }
    __syncthreads();  // the level's partitions and appended entries are done
    level_begin = level_end;
  }
  __syncthreads();
  return q->overflow == 0u;
}

#ifdef DLIOM_EXPERIMENTS
__device__ unsigned long long dbg_stamps[64 * 16];
__device__ unsigned long long dbg_acc[128 * 8];
#endif

// Sorts skey[0, pow2) ascending (pow2 a power of two >= 64, <= kMaxSlice; unused entries hold ~0).
__device__ __forceinline__ void bitonic_sort_keys(unsigned long long* skey, int pow2) {
  // bitonic sort, up to four keys per thread in registers (key i = t + r * 1024): exchanges at distance < 64 are lane
  // shuffles, at distance >= 1024 stay inside the thread, and only the distances 64 .. 512 go through LDS
  constexpr int kR = kMaxSlice / kThreads;
  const int rows = max(1, pow2 / kThreads);  // keys per thread in use
  unsigned long long v[kR];
#pragma unroll
  for (int r = 0; r < kR; ++r) {
    const int i = static_cast<int>(threadIdx.x) + r * kThreads;
    v[r] = i < pow2 ? skey[i] : ~0ull;
  }
  for (int k = 2; k <= pow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= kThreads) {  // partner key in the same thread: rows (0,1),(2,3) for j = 1024, (0,2),(1,3) for j = 2048
        static_assert(kR == 4, "the in-thread exchanges below are written for four keys per thread");
        auto exchange = [&](unsigned long long& a, unsigned long long& b, int row) {
          const int i = static_cast<int>(threadIdx.x) + row * kThreads;
          const bool up = (i & k) == 0;
          const unsigned long long lo = a < b ? a : b, hi = a < b ? b : a;
          a = up ? lo : hi;
          b = up ? hi : lo;
        };
        if (j == kThreads) {
          exchange(v[0], v[1], 0);
          exchange(v[2], v[3], 2);
        } else {
          exchange(v[0], v[2], 0);
          exchange(v[1], v[3], 1);
        }
      } else if (j >= 64) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kR; ++r) {
          const int i = static_cast<int>(threadIdx.x) + r * kThreads;
          if (r < rows && i < pow2) skey[i] = v[r];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kR; ++r) {
          const int i = static_cast<int>(threadIdx.x) + r * kThreads;
          if (r < rows && i < pow2) {
            const unsigned long long other = skey[i ^ j];
            const bool keep_min = ((i & j) == 0) == ((i & k) == 0);
            v[r] = keep_min ? (v[r] < other ? v[r] : other) : (v[r] < other ? other : v[r]);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < kR; ++r) {
          if (r >= rows) break;  // uniform
          const int i = static_cast<int>(threadIdx.x) + r * kThreads;
          const unsigned lo32 = static_cast<unsigned>(__shfl_xor(static_cast<int>(v[r] & 0xffffffffull), j, 64));
          const unsigned hi32 = static_cast<unsigned>(__shfl_xor(static_cast<int>(v[r] >> 32), j, 64));
          const unsigned long long other = (static_cast<unsigned long long>(hi32) << 32) | lo32;
          const bool keep_min = ((i & j) == 0) == ((i & k) == 0);
          v[r] = keep_min ? (v[r] < other ? v[r] : other) : (v[r] < other ? other : v[r]);
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kR; ++r) {
    const int i = static_cast<int>(threadIdx.x) + r * kThreads;
    if (i < pow2) skey[i] = v[r];
  }
  __syncthreads();
}

#include "rothist_big.h"

__global__ __launch_bounds__(kThreads) void slice_kernel(const float* __restrict__ rx, const float* __restrict__ ry,
                                                         const float* __restrict__ rz, const short* __restrict__ keys, int n,
                                                         const unsigned* __restrict__ bin_counts, int histogram_size,
                                                         float squared_jump, unsigned char* __restrict__ c_bucket,
                                                         float* __restrict__ c_value, unsigned* __restrict__ flags, int with_big) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long lds_dyn[];  // [sort keys kMaxSlice | sx | sy | sz]; later [.. | px py of the sorted points]
  unsigned long long* skey = lds_dyn;
  float* sx = reinterpret_cast<float*>(skey + kMaxSlice);
  float* sy = sx + kMaxSlice;
  float* sz = sy + kMaxSlice;
  // std::sort's order of equal keys (wave_sort_arrangement) and later steps: eight arrays of kMaxSlice + 8 u16
  unsigned short* u16_base = reinterpret_cast<unsigned short*>(sz + kMaxSlice);
  constexpr int kU16 = kMaxSlice + 8;
  unsigned short* idx_of = u16_base + 7 * kU16;  // arrangement position -> position in the slice
  unsigned char* tied_flags = reinterpret_cast<unsigned char*>(idx_of);  // by position in the slice; dead once idx_of is written
  unsigned char* act_flags = reinterpret_cast<unsigned char*>(u16_base + 8 * kU16);  // [kMaxSlice]; later the chain's marks
  static_assert(sizeof(Queue) <= static_cast<size_t>(kU16) * 2, "the queue fits one of the arrays");
  const SortScratch sort_scratch{u16_base, u16_base + kU16, tied_flags, reinterpret_cast<Queue*>(u16_base + 2 * kU16)};
  // std::sort's input and arrangement (the plainly sorted keys stay in skey): arrays 3 .. 6
  static_assert(4 * static_cast<size_t>(kU16) * 2 >= static_cast<size_t>(kMaxSlice) * 8, "four of the arrays hold the replay's items");
  unsigned long long* replay = reinterpret_cast<unsigned long long*>(u16_base + 3 * kU16 + 4);  // (+ 4: 8-byte aligned)
  unsigned short* scratch_g = u16_base;         // prefix counts of the valid points (before the replay); later the chain's pointers
  unsigned short* scratch_l = u16_base + kU16;  // ... and, between them, the tied elements' places and the final order
  __shared__ unsigned wave_sums[kThreads / 64];
  __shared__ int wave_max[kThreads / 64];
  __shared__ unsigned sh_bin, sh_count, sh_begin, sh_valid, sh_written;
  __shared__ float sh_centroid[3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // ---- which slice: the (blockIdx.x)-th non-empty bin, then every gridDim.x-th; offsets = prefix of the counts
  unsigned my_counts[kBins / kThreads], my_sum = 0u, my_nonempty = 0u;
  int my_big = 0;
#pragma unroll
  for (int k = 0; k < kBins / kThreads; ++k) {
    my_counts[k] = bin_counts[threadIdx.x * (kBins / kThreads) + k];
    my_sum += my_counts[k];
    my_nonempty += my_counts[k] != 0u ? 1u : 0u;
    my_big |= my_counts[k] > static_cast<unsigned>(kMaxSlice) ? 1 : 0;
  }
  unsigned total_points, total_slices;
  const unsigned points_before = block_exclusive_scan(my_sum, wave_sums, &total_points);
  const unsigned slices_before = block_exclusive_scan(my_nonempty, wave_sums, &total_slices);
  {
    // slices above kMaxSlice belong to the kernels of rothist_big.h.  flags[1] tells the host whether the cloud had any
    // (what the next cloud's launch plan is made from); flags[0] bit 3: it had, and those kernels were not enqueued
    const int any_big = __syncthreads_or(my_big);
    if (blockIdx.x == 0 && threadIdx.x == 0 && any_big) {
      flags[1] = 1u;
      if (!with_big) atomicOr(flags, 8u);
    }
  }
#ifdef DLIOM_EXPERIMENTS
#define DLIOM_STAMP(k) if (threadIdx.x == 0 && blockIdx.x < 64) dbg_stamps[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter()
#else
#define DLIOM_STAMP(k)
#endif
  DLIOM_STAMP(0);
  for (unsigned ordinal = blockIdx.x; ordinal < total_slices; ordinal += gridDim.x) {
    __syncthreads();
    {
      unsigned pb = points_before, sb = slices_before;
#pragma unroll
      for (int k = 0; k < kBins / kThreads; ++k) {
        if (my_counts[k] != 0u) {
          if (sb == ordinal) {
            sh_bin = threadIdx.x * (kBins / kThreads) + k;
            sh_count = my_counts[k];
            sh_begin = pb;
          }
          ++sb;
        }
        pb += my_counts[k];
      }
    }
    __syncthreads();
    const int key = static_cast<int>(sh_bin) - kBinOrigin;
    const int count = static_cast<int>(sh_count);
    const unsigned begin = sh_begin;
    if (count > kMaxSlice) continue;  // big_prepare_kernel / big_slice_kernel
    // ---- the slice's points in input order: wave w compacts its contiguous share of the input, 512 keys (8 per lane,
    //      one 16-byte load) per step; a lane's keys are consecutive, so input order = (step, lane, position in lane)
    const int blocks512 = (n + 511) / 512;
    const int per_wave = (blocks512 + kThreads / 64 - 1) / (kThreads / 64);
    const int blk_lo = wave * per_wave, blk_hi = min(blocks512, blk_lo + per_wave);
    unsigned mine = 0u;
    // (up to eight steps per wave -- clouds of 65 536 points -- the lane's eight match bits of every step stay in a
    // register: the second pass then reads no keys and compares nothing)
    unsigned long long kept_hits = 0ull;
    const bool keep_hits = per_wave <= 8;
    for (int blk = blk_lo; blk < blk_hi; ++blk) {
      const int i0 = blk * 512 + lane * 8;
      uint4 kk = make_uint4(0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu);  // no key is 0x7fff
      if (i0 < n) kk = *reinterpret_cast<const uint4*>(keys + i0);                 // the key array is padded to 512
      const unsigned w4[4] = {kk.x, kk.y, kk.z, kk.w};
      unsigned h8 = 0u;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const short kv = static_cast<short>((w4[t >> 1] >> ((t & 1) * 16)) & 0xffffu);
        h8 |= (i0 + t < n && kv == key) ? (1u << t) : 0u;
      }
      mine += static_cast<unsigned>(__builtin_popcount(h8));
      if (keep_hits) kept_hits |= static_cast<unsigned long long>(h8) << (8 * (blk - blk_lo));
    }
    DLIOM_STAMP(8);
    // per-wave totals -> where this wave's points start
    unsigned wave_total = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) wave_total += __shfl_xor(wave_total, d, 64);
    __syncthreads();
    if (lane == 0) wave_sums[wave] = wave_total;
    __syncthreads();
    unsigned at = 0u;
    for (int w = 0; w < wave; ++w) at += wave_sums[w];
    DLIOM_STAMP(9);
    for (int blk = blk_lo; blk < blk_hi; ++blk) {
      const int i0 = blk * 512 + lane * 8;
      unsigned hits = 0u;
      if (keep_hits) {
        hits = static_cast<unsigned>(kept_hits >> (8 * (blk - blk_lo))) & 0xffu;
      } else {
        uint4 kk = make_uint4(0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu);
        if (i0 < n) kk = *reinterpret_cast<const uint4*>(keys + i0);
        const unsigned w4[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const short kv = static_cast<short>((w4[t >> 1] >> ((t & 1) * 16)) & 0xffffu);
          if (i0 + t < n && kv == key) hits |= 1u << t;
        }
      }
      const unsigned cnt = static_cast<unsigned>(__builtin_popcount(hits));
      unsigned incl = cnt;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
      }
      unsigned slot = at + incl - cnt;
      while (hits != 0u) {
        const int t = __builtin_ctz(hits);
        sx[slot] = rx[i0 + t];
        sy[slot] = ry[i0 + t];  // (z decided the slice and is not read again)
        ++slot;
        hits &= hits - 1u;
      }
      at += __shfl(incl, 63, 64);
    }
    __syncthreads();
    DLIOM_STAMP(1);
    // ---- SortSlice: centroid (sequential float sums), angles, sort by angle
    if (lane == 0 && wave < 2)  // one thread per coordinate (on different SIMDs); z's sum is never read.  At most 4096
      // additions (14 us): the parallel replay of exact_sum.h pays from about that size on and is used by the big slices
      sh_centroid[wave] = thread_sequential_sum(wave == 0 ? sx : sy, count, 0.f) / static_cast<float>(count);
    __syncthreads();
    DLIOM_STAMP(2);
    int pow2 = 64;
    while (pow2 < count) pow2 <<= 1;
    {
      const float cx = sh_centroid[0], cy = sh_centroid[1];
      unsigned valid = 0u;
      for (int i = threadIdx.x; i < pow2; i += kThreads) {
        unsigned long long k64 = ~0ull;  // padding and skipped points sort to the end
        if (i < count) {
          const float dx = sx[i] - cx, dy = sy[i] - cy;
          if (!(norm2(dx, dy) < kMinDistance)) {
            k64 = (static_cast<unsigned long long>(ordered_bits(fd_atan2f(dy, dx))) << 32) | static_cast<unsigned>(i);
            ++valid;
          }
        }
        skey[i] = k64;
      }
      unsigned total_valid;
      block_exclusive_scan(valid, wave_sums, &total_valid);
      if (threadIdx.x == 0) sh_valid = total_valid;
    }
    __syncthreads();
    DLIOM_STAMP(3);
    bitonic_sort_keys(skey, pow2);
    DLIOM_STAMP(13);
    const int m = static_cast<int>(sh_valid);
    // equal angles next to each other: their order is std::sort's, not ours
    bool tied = false;
    {
      for (int i = threadIdx.x; i < count; i += kThreads) tied_flags[i] = 0;
      __syncthreads();
      int t = 0;
      for (int j = threadIdx.x; j + 1 < m; j += kThreads)
        if ((skey[j] >> 32) == (skey[j + 1] >> 32)) {
          tied_flags[static_cast<unsigned>(skey[j]) & 0xffffu] = 1;
          tied_flags[static_cast<unsigned>(skey[j + 1]) & 0xffffu] = 1;
          t = 1;
        }
      tied = __syncthreads_or(t) != 0;
    }
    if (tied) {
      // std::sort's input: the valid (angle, position) pairs in input order (blocked: thread t owns positions 4 t ...)
      const float cx = sh_centroid[0], cy = sh_centroid[1];
      const int p0 = 4 * static_cast<int>(threadIdx.x);
      unsigned long long item[4];
      unsigned ok[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = p0 + k;
        ok[k] = 0u;
        item[k] = 0ull;
        if (i < count) {
          const float dx = sx[i] - cx, dy = sy[i] - cy;
          if (!(norm2(dx, dy) < kMinDistance)) {
            ok[k] = 1u;
            item[k] = (static_cast<unsigned long long>(ordered_bits(fd_atan2f(dy, dx))) << 32) | static_cast<unsigned>(i);
          }
        }
      }
      __syncthreads();
      blocked_prefix(ok, scratch_g, wave_sums);
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (ok[k]) replay[scratch_g[p0 + k]] = item[k];
      __syncthreads();
      queue_init(sort_scratch.queue);
      __syncthreads();
      if (threadIdx.x == 0 && m > 16) {
        int depth = 0;
        for (int v = m; v > 1; v >>= 1) ++depth;
        queue_push(sort_scratch.queue, 0, m, 2 * depth);  // std::__lg(n) * 2
      }
      __syncthreads();
      DLIOM_STAMP(14);
      const bool done = wave_sort_arrangement(replay, sort_scratch);
      DLIOM_STAMP(15);
      if (!done) {  // the ring overflowed (cannot happen for m <= kMaxSlice): refuse, the host path takes the cloud
        if (threadIdx.x == 0) atomicOr(flags, 4u);
        continue;
      }
      // the final insertion sort is stable: among equal angles the arrangement's order stays.  skey is sorted by (angle,
      // position); every group of equal angles is put into the order of its members' places in the arrangement.
      unsigned short* pos_of = scratch_g;  // by position in the slice (tied elements only)
      unsigned short* order = scratch_l;   // j-th element of std::sort's result -> position in the slice
      for (int q = threadIdx.x; q < m; q += kThreads) {
        const unsigned id = static_cast<unsigned>(replay[q]) & 0xffffu;
        if (tied_flags[id]) pos_of[id] = static_cast<unsigned short>(q);
      }
      __syncthreads();
      for (int j = threadIdx.x; j < m; j += kThreads) {
        const unsigned long long it = skey[j];
        const unsigned id = static_cast<unsigned>(it) & 0xffffu;
        int dst = j;
        if (tied_flags[id]) {
          const unsigned key = static_cast<unsigned>(it >> 32);
          int gs = j, ge = j + 1;
          while (gs > 0 && static_cast<unsigned>(skey[gs - 1] >> 32) == key) --gs;
          while (ge < m && static_cast<unsigned>(skey[ge] >> 32) == key) ++ge;
          const unsigned mine = pos_of[id];
          int r = 0;
          for (int u = gs; u < ge; ++u) r += pos_of[static_cast<unsigned>(skey[u]) & 0xffffu] < mine ? 1 : 0;
          dst = gs + r;
        }
        order[dst] = static_cast<unsigned short>(id);
      }
      __syncthreads();
    }
    DLIOM_STAMP(4);
    // ---- the sorted slice, contiguous (x into the z array -- z is not needed any more -- and y behind the sort keys'
    //      low words is not possible: y goes to a second pass over sy via registers)
    float* px_sorted = sz;
    float my_py[kMaxSlice / kThreads];
#pragma unroll
    for (int r = 0; r < kMaxSlice / kThreads; ++r) {
      const int j = threadIdx.x + r * kThreads;
      my_py[r] = 0.f;
      if (j < m) {
        unsigned idx = static_cast<unsigned>(skey[j]) & 0xffffu;
        if (tied) idx = scratch_l[j];  // `order` of the branch above
        px_sorted[j] = sx[idx];
        my_py[r] = sy[idx];
      }
    }
    __syncthreads();
    float* py_sorted = sx;  // sx has been read out
#pragma unroll
    for (int r = 0; r < kMaxSlice / kThreads; ++r) {
      const int j = threadIdx.x + r * kThreads;
      if (j < m) py_sorted[j] = my_py[r];
    }
    __syncthreads();
    // ---- AddPointCloudSliceToHistogram: centroid of the SORTED points (sequential again; z is not used below)
    if (lane == 0 && wave < 2)
      sh_centroid[wave] = thread_sequential_sum(wave == 0 ? px_sorted : py_sorted, m, 0.f) / static_cast<float>(m);
    __syncthreads();
    DLIOM_STAMP(5);
    // (a) which points can never contribute: closer than kMinDistance to the centroid (:73-75)
    unsigned char* dead = reinterpret_cast<unsigned char*>(skey);                 // [kMaxSlice] bytes
    unsigned short* anchor_of = reinterpret_cast<unsigned short*>(dead + kMaxSlice);  // [kMaxSlice]: last_point, 0xFFFF = jump
    unsigned char* cb = reinterpret_cast<unsigned char*>(anchor_of + kMaxSlice);  // contributions in order: bucket ...
    float* cv = sy;                                                                // ... and value (sy has been read out)
    {
      const float cx = sh_centroid[0], cy = sh_centroid[1];
      for (int j = threadIdx.x; j < m; j += kThreads) dead[j] = norm2(px_sorted[j] - cx, py_sorted[j] - cy) < kMinDistance ? 1 : 0;
    }
    __syncthreads();
    // (b) the chain of `last_point`s (:70-80): last_point moves to the first live point farther than kMaxDistance from
    //     it.  fl(sqrt(s)) > kMaxDistance is a threshold on s itself (sqrt is monotone and correctly rounded): the
    //     comparison needs no square root.  Walking the chain costs a step per jump, and on a floor-like slice nearly
    //     every point is one; instead next(i) for every i at once, then the nodes on the path 0 -> next(0) -> ... by
    //     pointer doubling (marks spread along next^(2^d) while the pointers are squared; rothist_big.h).
    {
      unsigned short* ja = scratch_g;  // the sort's scratch is free
      unsigned short* jb = scratch_l;
      unsigned char* mark = act_flags;
      const int p0 = 4 * static_cast<int>(threadIdx.x);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = p0 + k;
        if (i < m) {
          const float ax = px_sorted[i], ay = py_sorted[i];
          int j = i + 1;
          for (; j < m; ++j) {
            if (dead[j]) continue;
            const float dx = px_sorted[j] - ax, dy = py_sorted[j] - ay;
            if (dx * dx + dy * dy >= squared_jump) break;
          }
          ja[i] = static_cast<unsigned short>(j);
          mark[i] = i == 0 ? 1 : 0;
        }
      }
      if (threadIdx.x == 0) {
        ja[m] = static_cast<unsigned short>(m);
        jb[m] = static_cast<unsigned short>(m);
        mark[m] = 0;  // (m <= kMaxSlice: the arrays have kMaxSlice + 8 entries, act_flags kMaxSlice + 8 bytes)
      }
      __syncthreads();
      for (int d = 0; (1 << d) < 2 * m; ++d) {
        // (a level in which no marked node has a successor 2^d steps on has run off the end of the path -- after d
        // levels the first 2^d nodes are marked, so that is all of them: on a wall the anchor moves every fifth to tenth
        // point, the path has m / 6 nodes and the last three or four levels would only square pointers)
        int reached = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = p0 + k;
          if (i < m) {
            const unsigned short t = ja[i];
            if (mark[i]) {
              mark[t] = 1;
              reached |= t < m ? 1 : 0;
            }
            jb[i] = ja[t];
          }
        }
        if (__syncthreads_or(reached) == 0) break;
        unsigned short* t = ja;
        ja = jb;
        jb = t;
      }
      // last_point of point j = the last marked position before it; a marked point (a jump) contributes nothing
      int last_marked = -1;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (p0 + k < m && mark[p0 + k]) last_marked = p0 + k;
      int anchor = max(0, block_exclusive_max(last_marked, -1, wave_max));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = p0 + k;
        if (j < m) {
          const bool jump = mark[j] != 0 && j != 0;
          anchor_of[j] = jump ? 0xFFFFu : static_cast<unsigned short>(anchor);
          if (mark[j]) anchor = j;
        }
      }
    }
    __syncthreads();
    // (c) every point against its last_point, all threads; contributions keep the order of the points
    {
      const float cx = sh_centroid[0], cy = sh_centroid[1];
      unsigned running = 0u;
      for (int j0 = 0; j0 < m; j0 += kThreads) {
        const int j = j0 + static_cast<int>(threadIdx.x);
        bool emit = false;
        float value = 0.f;
        int bucket = 0;
        if (j < m && dead[j] == 0 && anchor_of[j] != 0xFFFFu) {
          const int a = anchor_of[j];
          const float px = px_sorted[j], py = py_sorted[j];
          const float dx = px - px_sorted[a], dy = py - py_sorted[a];
          const float distance = norm2(dx, dy);
          if (!(distance < kMinDistance)) {
            const float ex = px - cx, ey = py - cy;
            const float direction_norm = norm2(ex, ey);
            const float dot = (dx / distance) * (ex / direction_norm) + (dy / distance) * (ey / direction_norm);
            bucket = bucket_of(fd_atan2f(dy, dx), histogram_size);
            value = fmaxf(0.f, 1.f - fabsf(dot));
            emit = true;
          }
        }
        unsigned round_total;
        const unsigned rank = block_exclusive_scan(emit ? 1u : 0u, wave_sums, &round_total);
        if (emit) {
          cb[running + rank] = static_cast<unsigned char>(bucket);
          cv[running + rank] = value;
        }
        running += round_total;
      }
      if (threadIdx.x == 0) sh_written = running;
    }
    __syncthreads();
    DLIOM_STAMP(6);
    if (threadIdx.x == 0 && blockIdx.x < 64) {
#ifdef DLIOM_EXPERIMENTS
      dbg_stamps[blockIdx.x * 16 + 10] = static_cast<unsigned long long>(count);
      dbg_stamps[blockIdx.x * 16 + 11] = static_cast<unsigned long long>(m);
      dbg_stamps[blockIdx.x * 16 + 12] = static_cast<unsigned long long>(sh_written);
#endif
    }
    // ---- stable partition of the slice's contributions by bucket into its region of c_value: kernel 3 then finds the
    //      values of (slice, bucket) contiguous and in order.  Wave w counts / writes buckets w, w + 16, ...
    {  // the slice's list goes to its region of the contribution arrays (entries it does not use keep bucket 255)
      const int E = static_cast<int>(sh_written);
      for (int e = threadIdx.x; e < E; e += kThreads) {
        c_bucket[begin + e] = cb[e];
        c_value[begin + e] = cv[e];
      }
    }
    DLIOM_STAMP(7);
  }
}

// ---- kernel 3: the additions, in the reference's order ------------------------------------------------------------
// The slices' contribution lists lie in slice order in ONE array (a slice's list starts where its points start; entries
// it did not use keep the bucket 255 the array was filled with): the order of the additions is the order of the array.
// One wave per bucket: (1) scan the bucket bytes, 1024 entries per step, and queue the positions of its own entries in
// order (wave prefix sums); (2) fetch their values, every lane busy; (3) one thread adds them one after the other.
constexpr int kAccCap = 24576;   // queue entries (dynamic LDS, 96 KB): a wall-dominated scan puts ~40 % of its points in one bucket
constexpr int kAccWaves = kThreads / 64;  // waves per bucket in the parallel scan (the exact sum is written for 1024 threads)
constexpr int kAccHoldSteps = 16;  // steps of 1024 entries a wave keeps as one match bit per (lane, row): 2^18 entries per bucket

// (2) + (3): values of the queued positions, every lane of the workgroup busy, then ONE thread adds them in order.
__device__ __forceinline__ float fetch_and_sum(float* queue, unsigned queued, const float* __restrict__ c_value, float sum,
                                               int tid, int nthreads, bool workgroup, exact_sum::Scratch<1>* es) {
  if (workgroup) __syncthreads();
  else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  for (unsigned e0 = 0; e0 < queued; e0 += static_cast<unsigned>(nthreads) * 16u) {  // sixteen gathers in flight per lane
    float got[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const unsigned e = e0 + static_cast<unsigned>(nthreads) * u + tid;
      got[u] = e < queued ? c_value[__float_as_uint(queue[e])] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const unsigned e = e0 + static_cast<unsigned>(nthreads) * u + tid;
      if (e < queued) queue[e] = got[u];
    }
  }
  if (workgroup) __syncthreads();
  else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  if (workgroup && queued >= static_cast<unsigned>(kExactSumFrom)) {
    // histogram(bucket) += value, one after the other (:49): a wall-dominated scan puts 18 000 of them into one bucket.
    // The same float is reached by composing the additions as parity functions (exact_sum.h); every thread gets it.
    const float* const arrays[1] = {queue};
    const float start[1] = {sum};
    float out[1];
    exact_sum::block_sequential_sums<1>(arrays, static_cast<int>(queued), start, out, *es);
    return out[0];
  }
  if (tid == 0) sum = thread_sequential_sum(queue, static_cast<int>(queued), sum);
  if (workgroup) __syncthreads();
  else __builtin_amdgcn_wave_barrier();
  return sum;
}

// One WORKGROUP per bucket.  With one wave per bucket the scan of the ~46 steps was a chain of 46 dependent
// load -> ballot -> queue rounds on 120 of the chip's 1024 SIMDs: 100 us, half of the whole histogram.  Now wave w scans
// its contiguous eighth of the array keeping one match bit per (lane, row) in registers, the waves' counts are prefixed,
// and every wave writes its positions where they belong -- the queue is in array order as before.  Arrays of more than
// 2^16 entries, or more than kAccCap entries in one bucket, take the serial path (wave 0, queue drained as it fills).
__global__ __launch_bounds__(64 * kAccWaves) void accumulate_kernel(const unsigned char* __restrict__ c_bucket,
                                                                    const float* __restrict__ c_value, int n_padded,
                                                                    int histogram_size, float* __restrict__ histogram) {
  extern __shared__ __attribute__((aligned(16))) float queue[];  // kAccCap + 64: positions (as bits), then the values in place
  __shared__ unsigned wave_count[kAccWaves];
  __shared__ exact_sum::Scratch<1> es;
  const int bucket = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (bucket >= histogram_size) return;
  const int steps = n_padded / 1024;
  const unsigned char want8 = static_cast<unsigned char>(bucket);
  const int per_wave = (steps + kAccWaves - 1) / kAccWaves;
  bool parallel = per_wave <= kAccHoldSteps;
  if (parallel) {
    const int s_begin = wave * per_wave, s_end = min(steps, s_begin + per_wave);
    unsigned bits[kAccHoldSteps / 2];  // 16 rows of step 2k in the low half, of step 2k + 1 in the high half
#pragma unroll
    for (int k = 0; k < kAccHoldSteps / 2; ++k) bits[k] = 0u;
    unsigned mine = 0u;
#pragma unroll
    for (int k = 0; k < kAccHoldSteps; ++k) {
      const int s = s_begin + k;
      if (s < s_end) {
        unsigned char row[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) row[t] = c_bucket[s * 1024 + t * 64 + lane];
        unsigned b = 0u;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const bool hit = row[t] == want8;
          b |= (hit ? 1u : 0u) << t;
          mine += static_cast<unsigned>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(hit)));
        }
        bits[k >> 1] |= b << (16 * (k & 1));
      }
    }
    if (lane == 0) wave_count[wave] = mine;
    __syncthreads();
    unsigned base = 0u, total = 0u;
#pragma unroll
    for (int w = 0; w < kAccWaves; ++w) {
      if (w < wave) base += wave_count[w];
      total += wave_count[w];
    }
    if (total <= static_cast<unsigned>(kAccCap)) {
#pragma unroll
      for (int k = 0; k < kAccHoldSteps; ++k) {
        const int s = s_begin + k;
        if (s < s_end) {
          const unsigned b = bits[k >> 1] >> (16 * (k & 1));
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const bool hit = ((b >> t) & 1u) != 0u;
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
            if (hit)
              queue[base + __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mask >> 32),
                                                     __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mask), 0u))] =
                  __uint_as_float(static_cast<unsigned>(s) * 1024u + 64u * t + lane);
            base += static_cast<unsigned>(__builtin_popcountll(mask));
          }
        }
      }
      const float sum = fetch_and_sum(queue, total, c_value, 0.f, static_cast<int>(threadIdx.x), 64 * kAccWaves, true, &es);
      if (threadIdx.x == 0) histogram[bucket] = sum;
      return;
    }
    parallel = false;  // a bucket with more entries than the queue holds
  }
  if (wave != 0) return;
  // ---- serial path: one wave, 1024 entries per step as 16 rows of 64 (lane l of row t looks at entry s * 1024 + t * 64 + l,
  // so a row's matches are neighbours in the array AND in the queue: consecutive lanes -> consecutive LDS words)
  float sum = 0.f;
  unsigned queued = 0u;
  for (int s = 0; s < steps; ++s) {
    unsigned char row[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) row[t] = c_bucket[s * 1024 + t * 64 + lane];
    unsigned long long masks[16];
    unsigned step_total = 0u;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      masks[t] = __builtin_amdgcn_ballot_w64(row[t] == want8);
      step_total += static_cast<unsigned>(__builtin_popcountll(masks[t]));
    }
    if (step_total == 0u) continue;
    if (queued + step_total > kAccCap) {
      sum = fetch_and_sum(queue, queued, c_value, sum, lane, 64, false, nullptr);
      sum = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sum)));
      queued = 0u;
    }
    unsigned base = queued;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if (row[t] == want8)
        queue[base + __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(masks[t] >> 32),
                                               __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(masks[t]), 0u))] =
            __uint_as_float(static_cast<unsigned>(s) * 1024u + 64u * t + lane);
      base += static_cast<unsigned>(__builtin_popcountll(masks[t]));
    }
    queued += step_total;
  }
  sum = fetch_and_sum(queue, queued, c_value, sum, lane, 64, false, nullptr);
  if (lane == 0) histogram[bucket] = sum;
}

}  // namespace rothist
}  // namespace dliom

using namespace dliom;

#ifdef DLIOM_EXPERIMENTS
extern "C" int dliom_exp_rothist_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(rothist::dbg_stamps), sizeof(unsigned long long) * 64 * 16) == hipSuccess ? 0 : -2;
}
#endif

#ifdef DLIOM_EXPERIMENTS
extern "C" int dliom_exp_rothist_big_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(rothist::dbg_big), sizeof(unsigned long long) * 64 * 16) == hipSuccess ? 0 : -2;
}
#endif

#ifdef DLIOM_EXPERIMENTS
extern "C" int dliom_exp_set_coop_min(int v) {
  return hipMemcpyToSymbol(HIP_SYMBOL(rothist::dbg_coop_min), &v, sizeof(v)) == hipSuccess ? 0 : -2;
}
#endif

#ifdef DLIOM_EXPERIMENTS
extern "C" int dliom_exp_exact_sum_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(exact_sum::dbg_es), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -2;
}
#endif

#ifdef DLIOM_EXPERIMENTS
extern "C" int dliom_exp_rothist_acc_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(rothist::dbg_acc), sizeof(unsigned long long) * 128 * 8) == hipSuccess ? 0 : -2;
}
#endif

namespace dliom {
namespace rothist {
// dliom_diag_std_sort_order: the slice kernel's sort (plain sort; on ties std::sort's partitions + stable sort) on bare keys
__global__ __launch_bounds__(kThreads) void std_sort_order_kernel(const float* __restrict__ keys, int n, int* __restrict__ order,
                                                                  int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long lds_dyn[];
  unsigned long long* skey = lds_dyn;
  unsigned short* u16_base = reinterpret_cast<unsigned short*>(skey + kMaxSlice);
  constexpr int kU16 = kMaxSlice + 8;
  unsigned short* idx_of = u16_base + 7 * kU16;
  unsigned char* tied_flags = reinterpret_cast<unsigned char*>(idx_of);
  const SortScratch sc{u16_base, u16_base + kU16, tied_flags, reinterpret_cast<Queue*>(u16_base + 2 * kU16)};
  int pow2 = 64;
  while (pow2 < n) pow2 <<= 1;
  // as in slice_kernel: a plain sort finds the elements whose key occurs more than once ...
  for (int i = threadIdx.x; i < pow2; i += kThreads)
    skey[i] = i < n ? (static_cast<unsigned long long>(ordered_bits(keys[i])) << 32) | static_cast<unsigned>(i) : ~0ull;
  __syncthreads();
  bitonic_sort_keys(skey, pow2);
  for (int i = threadIdx.x; i < n; i += kThreads) tied_flags[i] = 0;
  __syncthreads();
  for (int j = threadIdx.x; j + 1 < n; j += kThreads)
    if ((skey[j] >> 32) == (skey[j + 1] >> 32)) {
      tied_flags[static_cast<unsigned>(skey[j]) & 0xffffu] = 1;
      tied_flags[static_cast<unsigned>(skey[j + 1]) & 0xffffu] = 1;
    }
  __syncthreads();
  // ... then std::sort's input again, its partitions on the segments that hold ties, and the stable sort
  for (int i = threadIdx.x; i < pow2; i += kThreads)
    skey[i] = i < n ? (static_cast<unsigned long long>(ordered_bits(keys[i])) << 32) | static_cast<unsigned>(i) : ~0ull;
  __syncthreads();
  queue_init(sc.queue);
  __syncthreads();
  if (threadIdx.x == 0 && n > 16) {
    int depth = 0;
    for (int v = n; v > 1; v >>= 1) ++depth;
    queue_push(sc.queue, 0, n, 2 * depth);
  }
  __syncthreads();
  const bool done = wave_sort_arrangement(skey, sc);
  if (!done) {
    if (threadIdx.x == 0) *status = 1;
    return;
  }
  for (int q = threadIdx.x; q < pow2; q += kThreads) {
    if (q < n) {
      const unsigned long long it = skey[q];
      idx_of[q] = static_cast<unsigned short>(it & 0xffffu);
      skey[q] = (it & 0xffffffff00000000ull) | static_cast<unsigned>(q);
    } else {
      skey[q] = ~0ull;
    }
  }
  __syncthreads();
  bitonic_sort_keys(skey, pow2);
  for (int j = threadIdx.x; j < n; j += kThreads) order[j] = idx_of[static_cast<unsigned>(skey[j]) & 0xffffu];
  if (threadIdx.x == 0) *status = 0;
}
}  // namespace rothist
}  // namespace dliom

namespace {
// Carves the arrays of rothist_big.h out of `base` (returns the bytes used); `entries` = points + 64 (every slice works at
// offset begin + ordinal and keeps one sentinel behind its last entry)
size_t carve_big_arrays(char* base, size_t entries, dliom::rothist::BigArrays* A) {
  size_t at = 0;
  auto take = [&](size_t bytes) {
    char* p = base == nullptr ? nullptr : base + at;
    at += (bytes + 255) & ~static_cast<size_t>(255);
    return p;
  };
  const size_t e = entries + 64;
  A->key_in = reinterpret_cast<unsigned long long*>(take(e * 8));
  A->key_out = reinterpret_cast<unsigned long long*>(take(e * 8));
  A->arr = reinterpret_cast<unsigned long long*>(take(e * 8));
  A->bx = reinterpret_cast<float*>(take(e * 4));
  A->by = reinterpret_cast<float*>(take(e * 4));
  A->spx = reinterpret_cast<float*>(take(e * 4));
  A->spy = reinterpret_cast<float*>(take(e * 4));
  A->val_in = reinterpret_cast<unsigned*>(take(e * 4));
  A->val_out = reinterpret_cast<unsigned*>(take(e * 4));
  unsigned** u32s[] = {&A->seg_first, &A->seg_last, &A->g, &A->l, &A->tmp_l, &A->tmp_r, &A->cut, &A->tpre, &A->pos_of, &A->sorted_id,
                       &A->jump_a, &A->jump_b};
  for (unsigned** q : u32s) *q = reinterpret_cast<unsigned*>(take(e * 4));
  unsigned char** u8s[] = {&A->act, &A->fl, &A->tied, &A->dead, &A->mark};
  for (unsigned char** q : u8s) *q = reinterpret_cast<unsigned char*>(take(e));
  A->valid = reinterpret_cast<unsigned*>(take(256));
  return at;
}
}  // namespace

extern "C" int dliom_diag_std_sort_order(dliom_ctx* ctx, const float* keys, int n, int32_t* order) {
  using namespace rothist;
  if (ctx == nullptr || (n > 0 && (keys == nullptr || order == nullptr)) || n < 0 || n > (1 << 22)) return DLIOM_ERR_INVALID_ARGUMENT;
  if (n == 0) return DLIOM_OK;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  if (n > kMaxSlice) {
    // the path of slices above kMaxSlice (rothist_big.h): radix sort + the order of equal keys from introsort's partitions
    BigArrays A;
    const size_t entries = static_cast<size_t>(n);
    const size_t big_bytes = carve_big_arrays(nullptr, entries, &A);
    size_t temp_bytes = 0;
    DLIOM_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, A.key_in, A.key_out, A.val_in, A.val_out, n, 0, 32, ctx->stream));
    const size_t keys_at = (big_bytes + 255) & ~static_cast<size_t>(255);
    const size_t order_at = keys_at + ((static_cast<size_t>(n) * 4 + 255) & ~static_cast<size_t>(255));
    const size_t temp_at = order_at + ((static_cast<size_t>(n) * 4 + 256 + 255) & ~static_cast<size_t>(255));
    DLIOM_TRY(ctx->misc.reserve(temp_at + temp_bytes + 256));
    char* base = static_cast<char*>(ctx->misc.p);
    carve_big_arrays(base, entries, &A);
    float* d_keys = reinterpret_cast<float*>(base + keys_at);
    int* d_order = reinterpret_cast<int*>(base + order_at);
    int* d_status = d_order + n;
    DLIOM_HIP_TRY(hipMemcpyAsync(d_keys, keys, static_cast<size_t>(n) * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(big_sort_items_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_keys, n, A.key_in, A.val_in);
    DLIOM_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(base + temp_at, temp_bytes, A.key_in, A.key_out, A.val_in, A.val_out, n, 0, 32,
                                                     ctx->stream));
    if ((ctx->func_attr_set & kFuncAttrHistogramBig) == 0u) {
      DLIOM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(big_sort_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        static_cast<int>(kBigLdsBytes)));
      DLIOM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(big_slice_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        static_cast<int>(kBigLdsBytes)));
      ctx->func_attr_set |= kFuncAttrHistogramBig;
    }
    hipLaunchKernelGGL(big_sort_order_kernel, dim3(1), dim3(kThreads), kBigLdsBytes, ctx->stream, n, A, d_order, d_status);
    DLIOM_HIP_TRY(hipGetLastError());
    int status = 0;
    DLIOM_HIP_TRY(hipMemcpyAsync(order, d_order, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost, ctx->stream));
    DLIOM_HIP_TRY(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
    DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return status == 0 ? DLIOM_OK : DLIOM_ERR_CAPACITY;
  }
  DLIOM_TRY(ctx->misc.reserve(static_cast<size_t>(kMaxSlice) * 8 + 256));
  float* d_keys = ctx->misc.as<float>();
  int* d_order = reinterpret_cast<int*>(d_keys + kMaxSlice);
  int* d_status = d_order + kMaxSlice;
  DLIOM_HIP_TRY(hipMemcpyAsync(d_keys, keys, static_cast<size_t>(n) * 4, hipMemcpyHostToDevice, ctx->stream));
  const size_t lds = static_cast<size_t>(kMaxSlice) * 8 + 8 * static_cast<size_t>(kMaxSlice + 8) * 2 + kMaxSlice + 64;
  if ((ctx->func_attr_set & kFuncAttrStdSortDiag) == 0u) {
    DLIOM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(std_sort_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds)));
    ctx->func_attr_set |= kFuncAttrStdSortDiag;
  }
  hipLaunchKernelGGL(std_sort_order_kernel, dim3(1), dim3(kThreads), lds, ctx->stream, d_keys, n, d_order, d_status);
  DLIOM_HIP_TRY(hipGetLastError());
  int status = 0;
  DLIOM_HIP_TRY(hipMemcpyAsync(order, d_order, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return status == 0 ? DLIOM_OK : DLIOM_ERR_CAPACITY;  // std::sort's depth limit (heap sort from there): not reproduced
}

// The exact parallel replay of a sequential float sum (exact_sum.h) on bare values, for the tests: k arrays of n floats
// (values: k x n row major), each started at acc0[k] -> sums[k].
namespace dliom {
namespace rothist {
__global__ __launch_bounds__(kThreads) void sequential_sums_kernel(const float* __restrict__ values, int n, const float* __restrict__ acc0,
                                                                   float* __restrict__ sums) {
  __shared__ exact_sum::Scratch<1> es;
  const float* const arrays[1] = {values + static_cast<size_t>(blockIdx.x) * n};
  const float start[1] = {acc0[blockIdx.x]};
  float out[1];
  exact_sum::block_sequential_sums<1>(arrays, n, start, out, es);
  if (threadIdx.x == 0) sums[blockIdx.x] = out[0];
}
}  // namespace rothist
}  // namespace dliom

extern "C" int dliom_diag_sequential_sums(dliom_ctx* ctx, const float* values, int k, int n, const float* acc0, float* sums) {
  if (ctx == nullptr || values == nullptr || acc0 == nullptr || sums == nullptr || k <= 0 || k > 65535 || n < 0 ||
      static_cast<int64_t>(k) * n > (int64_t{1} << 28))
    return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  const size_t v_bytes = (static_cast<size_t>(k) * n * 4 + 255) & ~static_cast<size_t>(255);
  DLIOM_TRY(ctx->misc.reserve(v_bytes + static_cast<size_t>(k) * 8 + 256));
  float* d_v = ctx->misc.as<float>();
  float* d_a = reinterpret_cast<float*>(static_cast<char*>(ctx->misc.p) + v_bytes);
  float* d_s = d_a + k;
  if (n > 0) DLIOM_HIP_TRY(hipMemcpyAsync(d_v, values, static_cast<size_t>(k) * n * 4, hipMemcpyHostToDevice, ctx->stream));
  DLIOM_HIP_TRY(hipMemcpyAsync(d_a, acc0, static_cast<size_t>(k) * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(rothist::sequential_sums_kernel, dim3(static_cast<unsigned>(k)), dim3(rothist::kThreads), 0, ctx->stream, d_v, n, d_a, d_s);
  DLIOM_HIP_TRY(hipGetLastError());
  DLIOM_HIP_TRY(hipMemcpyAsync(sums, d_s, static_cast<size_t>(k) * 4, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return DLIOM_OK;
}

namespace {
constexpr int kRetryWithBigPath = 1000;  // read_histogram: the cloud has slices above kMaxSlice and their kernels were not enqueued

// Enqueues the kernels and the read-back of [histogram | flags | had-big-slices] into `pinned_dst` on `stream`; no
// synchronisation.  with_big: the kernels of rothist_big.h are part of the chain.
int enqueue_histogram(dliom_ctx* ctx, hipStream_t stream, dliom::DevBuf& scratch, void* pinned_dst, const dliom_cloud* cloud,
                      const float rotation_wxyz[4], int histogram_size, bool with_big, unsigned* done_word = nullptr,
                      unsigned done_seq = 0) {
  using namespace rothist;
  const int n = static_cast<int>(cloud->n);
  // scratch: [rx | ry | rz | c_value] floats, [keys] shorts (padded to 512), [c_bucket] bytes (padded to 1024),
  // [bin_counts | flags], [histogram], then the arrays of the big path and the radix sort's temporary storage
  const size_t N = static_cast<size_t>(n);
  const size_t n_padded = (N + 1023) & ~static_cast<size_t>(1023);
  const size_t f_bytes = n_padded * 4;
  const size_t k_bytes = n_padded * 2;
  const size_t b_bytes = n_padded;
  const size_t counts_bytes = (kBins + 64) * 4;
  const size_t hist_bytes = 1024;
  const size_t small_bytes = 4 * f_bytes + k_bytes + b_bytes + counts_bytes + hist_bytes;
  BigArrays A{};
  size_t big_bytes = 0, temp_bytes = 0;
  const int sort_items = n + 64;
  if (with_big) {
    big_bytes = carve_big_arrays(nullptr, N, &A);
    DLIOM_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, A.key_in, A.key_out, A.val_in, A.val_out, sort_items, 0,
                                                     kBigKeyBits, stream));
  }
  DLIOM_TRY(scratch.reserve(small_bytes + big_bytes + temp_bytes + 512));
  char* base = static_cast<char*>(scratch.p);
  float* rx = reinterpret_cast<float*>(base);
  float* ry = reinterpret_cast<float*>(base + f_bytes);
  float* rz = reinterpret_cast<float*>(base + 2 * f_bytes);
  float* c_value = reinterpret_cast<float*>(base + 3 * f_bytes);
  short* keys = reinterpret_cast<short*>(base + 4 * f_bytes);
  unsigned char* c_bucket = reinterpret_cast<unsigned char*>(base + 4 * f_bytes + k_bytes);
  unsigned* bin_counts = reinterpret_cast<unsigned*>(base + 4 * f_bytes + k_bytes + b_bytes);
  unsigned* flags = bin_counts + kBins;  // [0] refusals, [1] the cloud has slices above kMaxSlice
  float* d_hist = reinterpret_cast<float*>(base + 4 * f_bytes + k_bytes + b_bytes + counts_bytes);
  char* big_base = base + small_bytes;
  void* sort_temp = big_base + big_bytes;
  if (with_big) carve_big_arrays(big_base, N, &A);
  {
    FillJob fills[3] = {{bin_counts, counts_bytes, 0u}, {c_bucket, b_bytes, 0xFFFFFFFFu}, {nullptr, 0, 0u}};  // bucket 255 = no entry
    if (with_big) fills[2] = FillJob{A.key_in, (static_cast<size_t>(sort_items) * 8 + 3) & ~static_cast<size_t>(3), 0xFFFFFFFFu};  // padding keys sort last
    DLIOM_TRY(fill_multi(ctx, fills, with_big ? 3 : 2, stream));
  }
  Quat4 q{1.f, 0.f, 0.f, 0.f};
  if (rotation_wxyz != nullptr) q = Quat4{rotation_wxyz[0], rotation_wxyz[1], rotation_wxyz[2], rotation_wxyz[3]};
  const unsigned blocks = static_cast<unsigned>((N + kThreads - 1) / kThreads);
  hipLaunchKernelGGL(prepare_kernel, dim3(blocks), dim3(kThreads), 0, stream, cloud->d_x, cloud->d_y, cloud->d_z, n, q,
                     rotation_wxyz != nullptr ? 1 : 0, rx, ry, rz, keys, bin_counts, flags);
  const size_t lds = static_cast<size_t>(kMaxSlice) * (8 + 12) + 8 * static_cast<size_t>(kMaxSlice + 8) * 2 + kMaxSlice + 64 + 1024;
  const size_t acc_lds = static_cast<size_t>(kAccCap + 64) * 4;
  if ((ctx->func_attr_set & kFuncAttrHistogram) == 0u) {
    DLIOM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(slice_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds)));
    DLIOM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(accumulate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(acc_lds)));
    ctx->func_attr_set |= kFuncAttrHistogram;
  }
  // the smallest float s with fl(sqrt(s)) > kMaxDistance: `distance > kMaxDistance` as a comparison of squared lengths
  static const float squared_jump = [] {
    float s2 = kMaxDistance * kMaxDistance;
    while (std::sqrt(s2) > kMaxDistance) s2 = std::nextafter(s2, 0.f);
    while (!(std::sqrt(s2) > kMaxDistance)) s2 = std::nextafter(s2, 2.f);
    return s2;
  }();
  hipStream_t big_stream = stream;
  if (with_big) {
    // the big slices' chain (prepare, radix sort, one workgroup per slice) beside the small slices' kernel: both only
    // read what prepare_kernel wrote and write disjoint regions of the contribution arrays
    if (ctx->hist_big_stream == nullptr) {
      hipStream_t st = nullptr;
      hipEvent_t e1 = nullptr, e2 = nullptr;
      const bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
                      hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&e2, hipEventDisableTiming) == hipSuccess;
      if (!ok) {
        if (e2 != nullptr) (void)hipEventDestroy(e2);
        if (e1 != nullptr) (void)hipEventDestroy(e1);
        if (st != nullptr) (void)hipStreamDestroy(st);
        return DLIOM_ERR_HIP;
      }
      ctx->hist_big_stream = st;
      ctx->hist_fork = e1;
      ctx->hist_join = e2;
    }
    big_stream = ctx->hist_big_stream;
    DLIOM_HIP_TRY(hipEventRecord(ctx->hist_fork, stream));
    DLIOM_HIP_TRY(hipStreamWaitEvent(big_stream, ctx->hist_fork, 0));
    // From here on kernels may be running on big_stream: an error return must not leave them unjoined (the next call may
    // reallocate or reuse the scratch they read and write) -- every failure waits for that stream before it returns.
    auto forked = [&](hipError_t e) {
      if (e == hipSuccess) return true;
      set_last_error("rotational histogram, big-slice chain", e, __FILE__, __LINE__);
      (void)hipStreamSynchronize(big_stream);
      return false;
    };
    hipLaunchKernelGGL(big_prepare_kernel, dim3(kMaxBig), dim3(kThreads), 0, big_stream, rx, ry, keys, n, bin_counts, A, flags);
    if (!forked(hipGetLastError())) return DLIOM_ERR_HIP;
    if (!forked(hipcub::DeviceRadixSort::SortPairs(sort_temp, temp_bytes, A.key_in, A.key_out, A.val_in, A.val_out, sort_items, 0,
                                                   kBigKeyBits, big_stream)))
      return DLIOM_ERR_HIP;
    if ((ctx->func_attr_set & kFuncAttrHistogramBig) == 0u) {
      if (!forked(hipFuncSetAttribute(reinterpret_cast<const void*>(big_sort_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(kBigLdsBytes))) ||
          !forked(hipFuncSetAttribute(reinterpret_cast<const void*>(big_slice_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(kBigLdsBytes))))
        return DLIOM_ERR_HIP;
      ctx->func_attr_set |= kFuncAttrHistogramBig;
    }
    hipLaunchKernelGGL(big_slice_kernel, dim3(kMaxBig), dim3(kThreads), kBigLdsBytes, big_stream, bin_counts, histogram_size, squared_jump, A,
                       c_bucket, c_value, flags);
    if (!forked(hipGetLastError())) return DLIOM_ERR_HIP;
    if (!forked(hipEventRecord(ctx->hist_join, big_stream))) return DLIOM_ERR_HIP;
  }
  hipLaunchKernelGGL(slice_kernel, dim3(256), dim3(kThreads), lds, stream, rx, ry, rz, keys, n, bin_counts, histogram_size,
                     squared_jump, c_bucket, c_value, flags, with_big ? 1 : 0);
  if (with_big && hipStreamWaitEvent(stream, ctx->hist_join, 0) != hipSuccess) {
    (void)hipStreamSynchronize(ctx->hist_big_stream);  // joined the hard way
    return DLIOM_ERR_HIP;
  }
  hipLaunchKernelGGL(accumulate_kernel, dim3(static_cast<unsigned>(histogram_size)), dim3(64 * kAccWaves), acc_lds, stream, c_bucket,
                     c_value, static_cast<int>(n_padded), histogram_size, d_hist);
  DLIOM_HIP_TRY(hipGetLastError());
  // one read-back: [histogram | flags | had-big] through pinned memory
  const GatherJob back[2] = {{d_hist, static_cast<unsigned>(histogram_size)}, {flags, 2}};
  return gather_to_pinned(ctx, back, 2, pinned_dst, stream, done_word, done_seq);
}

// DLIOM_OK, DLIOM_ERR_CAPACITY (use dliom_rotational_histogram) or kRetryWithBigPath; *had_big: the cloud had slices
// above kMaxSlice
int read_histogram(const void* pinned_src, int histogram_size, float* histogram, bool* had_big) {
  const float* h = static_cast<const float*>(pinned_src);
  unsigned f[2];
  std::memcpy(f, h + histogram_size, 8);
  *had_big = f[1] != 0u;
  if ((f[0] & ~8u) != 0u) return DLIOM_ERR_CAPACITY;  // |z| >= 409.6 m, non-finite coordinates, more than 63 big slices, ...
  if ((f[0] & 8u) != 0u) return kRetryWithBigPath;
  std::memcpy(histogram, h, static_cast<size_t>(histogram_size) * 4);
  return DLIOM_OK;
}

// The histogram's completion word, polled for twice as long as the previous histogram took from enqueue to arrival (a
// scan with a floor: ~0.4 ms, a cube scan ~0.14 ms), at least the 150 us of every other read-back and at most 3 ms.
static int wait_histogram(dliom_ctx* ctx, hipStream_t stream, const unsigned* done_word, unsigned seq) {
  const int s = wait_done(ctx, stream, done_word, seq, ctx->hist_poll_us);
  const double took_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ctx->hist_enqueued_at).count();
  ctx->hist_poll_us = static_cast<int>(std::min(3000.0, std::max(150.0, 2.0 * took_us)));
  return s;
}

// One complete histogram on `stream`, waiting for it; runs the cloud a second time when it turns out to need the big
// path that was not enqueued (the context then expects big slices from the next cloud on).
int run_histogram(dliom_ctx* ctx, hipStream_t stream, dliom::DevBuf& scratch, void* pinned, unsigned* done_word, unsigned* seq_counter,
                  const dliom_cloud* cloud, const float rotation_wxyz[4], int histogram_size, float* histogram) {
  for (int attempt = 0; attempt < 2; ++attempt) {
    const bool with_big = ctx->hist_expect_big || attempt == 1;
    if (done_word != nullptr) {
      const unsigned seq = ++*seq_counter == 0u ? ++*seq_counter : *seq_counter;
      ctx->hist_enqueued_at = std::chrono::steady_clock::now();
      DLIOM_TRY(enqueue_histogram(ctx, stream, scratch, pinned, cloud, rotation_wxyz, histogram_size, with_big, done_word, seq));
      DLIOM_TRY(wait_histogram(ctx, stream, done_word, seq));
    } else {
      DLIOM_TRY(enqueue_histogram(ctx, stream, scratch, pinned, cloud, rotation_wxyz, histogram_size, with_big));
      DLIOM_HIP_TRY(hipStreamSynchronize(stream));
    }
    bool had_big = false;
    const int status = read_histogram(pinned, histogram_size, histogram, &had_big);
    ctx->hist_expect_big = had_big;
    if (status != kRetryWithBigPath) return status;
  }
  return DLIOM_ERR_CAPACITY;
}
}  // namespace

extern "C" int dliom_cloud_rotational_histogram(dliom_ctx* ctx, const dliom_cloud* cloud, const float rotation_wxyz[4],
                                                int histogram_size, float* histogram) {
  if (ctx == nullptr || cloud == nullptr || histogram == nullptr || histogram_size <= 0 || histogram_size > 255)
    return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  if (cloud->n == 0) {
    for (int i = 0; i < histogram_size; ++i) histogram[i] = 0.f;
    return DLIOM_OK;
  }
  if (cloud->n > (int64_t{1} << 26)) return DLIOM_ERR_CAPACITY;
  void* h = static_cast<char*>(ctx->pinned) + 2048;
  return run_histogram(ctx, ctx->stream, ctx->misc, h, ctx->done_word, &ctx->done_seq, cloud, rotation_wxyz, histogram_size, histogram);
}

// Test hook: the (bucket, value) pairs the histogram is summed from, in the order of the additions (slice order, then the
// order of the sorted points) -- the arrays accumulate_kernel reads, compacted.  A histogram of sums in the hundreds cannot
// tell whether a contribution of 1e-5 went into the right bucket; this can.  Blocking; up to `capacity` entries stored.
extern "C" int dliom_diag_histogram_contributions(dliom_ctx* ctx, const dliom_cloud* cloud, const float rotation_wxyz[4],
                                                  int histogram_size, int32_t* buckets, float* values, int64_t capacity,
                                                  int64_t* count) {
  if (ctx == nullptr || cloud == nullptr || count == nullptr || capacity < 0 || (capacity > 0 && (buckets == nullptr || values == nullptr)))
    return DLIOM_ERR_INVALID_ARGUMENT;
  *count = 0;
  if (cloud->n == 0) return DLIOM_OK;
  std::vector<float> histogram(static_cast<size_t>(histogram_size > 0 ? histogram_size : 1));
  DLIOM_TRY(dliom_cloud_rotational_histogram(ctx, cloud, rotation_wxyz, histogram_size, histogram.data()));
  // the layout of enqueue_histogram's scratch (ctx->misc, untouched since the call above returned)
  const size_t N = static_cast<size_t>(cloud->n);
  const size_t n_padded = (N + 1023) & ~static_cast<size_t>(1023);
  const char* base = static_cast<const char*>(ctx->misc.p);
  std::vector<float> v(n_padded);
  std::vector<unsigned char> b(n_padded);
  DLIOM_HIP_TRY(hipMemcpy(v.data(), base + 3 * n_padded * 4, n_padded * 4, hipMemcpyDeviceToHost));
  DLIOM_HIP_TRY(hipMemcpy(b.data(), base + 4 * n_padded * 4 + n_padded * 2, n_padded, hipMemcpyDeviceToHost));
  int64_t at = 0;
  for (size_t i = 0; i < n_padded; ++i) {
    if (b[i] == 0xFFu) continue;
    if (at < capacity) {
      buckets[at] = b[i];
      values[at] = v[i];
    }
    ++at;
  }
  *count = at;
  return DLIOM_OK;
}

// The same in two halves on the context's auxiliary stream: everything enqueued on the context so far is waited for
// (an event), then the histogram runs BESIDE whatever the caller puts on the context next -- the reference computes it
// right after InsertIntoSubmap from the same filtered cloud (local_trajectory_builder_3d.cc:590-610); neither writes it.
// The cloud must stay alive and unchanged until _finish returns (a cloud that turns out to have slices above 4096 points
// when the context did not expect any is run again there).
extern "C" int dliom_cloud_rotational_histogram_begin(dliom_ctx* ctx, const dliom_cloud* cloud, const float rotation_wxyz[4],
                                                      int histogram_size) {
  if (ctx == nullptr || cloud == nullptr || histogram_size <= 0 || histogram_size > 255) return DLIOM_ERR_INVALID_ARGUMENT;
  if (ctx->aux_histogram_size != 0) return DLIOM_ERR_INVALID_ARGUMENT;  // one pending histogram per context
  if (cloud->n > (int64_t{1} << 26)) return DLIOM_ERR_CAPACITY;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  if (ctx->aux_stream == nullptr) {  // all three or none
    hipStream_t st = nullptr;
    hipEvent_t ev = nullptr;
    void* pin = nullptr;
    const bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
                    hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess &&
                    hipHostMalloc(&pin, 4096, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess;
    if (!ok) {
      if (pin != nullptr) (void)hipHostFree(pin);
      if (ev != nullptr) (void)hipEventDestroy(ev);
      if (st != nullptr) (void)hipStreamDestroy(st);
      return DLIOM_ERR_HIP;
    }
    std::memset(pin, 0, 4096);
    ctx->aux_stream = st;
    ctx->aux_fork = ev;
    ctx->aux_pinned = pin;
  }
  if (cloud->n == 0) {
    std::memset(ctx->aux_pinned, 0, 4032);
    ctx->aux_histogram_size = histogram_size;
    ctx->aux_enqueued = false;
    return DLIOM_OK;
  }
  DLIOM_HIP_TRY(hipEventRecord(ctx->aux_fork, ctx->stream));
  DLIOM_HIP_TRY(hipStreamWaitEvent(ctx->aux_stream, ctx->aux_fork, 0));
  ctx->aux_seq = ++ctx->aux_seq == 0u ? 1u : ctx->aux_seq;
  ctx->hist_enqueued_at = std::chrono::steady_clock::now();
  DLIOM_TRY(enqueue_histogram(ctx, ctx->aux_stream, ctx->aux_scratch, ctx->aux_pinned, cloud, rotation_wxyz, histogram_size,
                              ctx->hist_expect_big, reinterpret_cast<unsigned*>(static_cast<char*>(ctx->aux_pinned) + 4032), ctx->aux_seq));
  ctx->aux_histogram_size = histogram_size;
  ctx->aux_enqueued = true;
  ctx->aux_cloud = cloud;
  ctx->aux_has_rotation = rotation_wxyz != nullptr;
  if (rotation_wxyz != nullptr) std::memcpy(ctx->aux_rotation, rotation_wxyz, sizeof(ctx->aux_rotation));
  return DLIOM_OK;
}

extern "C" int dliom_cloud_rotational_histogram_finish(dliom_ctx* ctx, float* histogram) {
  if (ctx == nullptr || histogram == nullptr || ctx->aux_histogram_size == 0) return DLIOM_ERR_INVALID_ARGUMENT;
  const int size = ctx->aux_histogram_size;
  ctx->aux_histogram_size = 0;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  unsigned* word = reinterpret_cast<unsigned*>(static_cast<char*>(ctx->aux_pinned) + 4032);
  if (ctx->aux_enqueued) DLIOM_TRY(wait_histogram(ctx, ctx->aux_stream, word, ctx->aux_seq));
  bool had_big = false;
  const int status = read_histogram(ctx->aux_pinned, size, histogram, &had_big);
  if (ctx->aux_enqueued) ctx->hist_expect_big = had_big;
  if (status != kRetryWithBigPath) return status;
  ctx->hist_expect_big = true;  // the first cloud with a floor after clouds without one: once more, with the big path
  return run_histogram(ctx, ctx->aux_stream, ctx->aux_scratch, ctx->aux_pinned, word, &ctx->aux_seq, ctx->aux_cloud,
                       ctx->aux_has_rotation ? ctx->aux_rotation : nullptr, size, histogram);
}
