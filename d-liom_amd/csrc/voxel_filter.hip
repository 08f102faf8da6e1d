// sensor::VoxelFilter / AdaptiveVoxelFilter on device-resident clouds.
//
//   sensor/internal/voxel_filter.cc:81-90   VoxelFilter::Filter: keep a point iff it is the FIRST of
//                                           its voxel in input order; output keeps the input order
//   sensor/internal/voxel_filter.cc:126-131 voxel index = RoundToInt(p / size) per axis
//   sensor/internal/voxel_filter.cc:28-77   FilterByMaxRange + AdaptivelyVoxelFiltered (the search
//                                           over edge lengths only needs survivor COUNTS)
//   sensor/internal/voxel_filter.cc:147-150 AdaptiveVoxelFilter::Filter
//
// "First point of its voxel" is order dependent, so it is computed as min-index-per-key:
//   1. insert: every point claims the slot of its 63-bit voxel key in an open-addressing table
//      (64-bit CAS) and atomicMin()s its index into the slot; the number of claimed slots is the
//      survivor count.  Several edge lengths run in ONE launch (grid.y), each with its own table:
//      the adaptive filter needs the counts of 8 halvings, then of the <= 16 nodes of its
//      bisection tree -- two launches and two 8-byte readbacks instead of ~10 host hash passes.
//   2. flag: point i survives iff table[slot_i].min_index == i; block survivor counts.
//   3. compact: exclusive scan of the flags (block counts + wave ballots), order-preserving
//      scatter of x, y, z; the survivors' max squared norm falls out of the same pass.
// Results are bit-identical to the host filter for every input whose voxel indices fit 21 bits
// per axis (|p / size| < 2^20: +-15 km at the adaptive filter's smallest edge of 1.5 cm);
// anything else returns DLIOM_ERR_INVALID_ARGUMENT.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "device_common.h"
#include "internal.h"

namespace dliom {

int read_max_norm(dliom_ctx* ctx, const unsigned* d_max_sq, float* max_norm);

constexpr int kVfBlock = 256;
constexpr int kMaxLengths = 32;   // edge lengths per insert launch (two filters x 16 bisection nodes)
constexpr int kMaxTreeNodes = 16;  // nodes of one filter's bisection tree
constexpr int kMaxFilters = 2;     // adaptive filters searched together (high and low resolution)
constexpr unsigned long long kEmptyKey = ~0ull;
constexpr unsigned kNoSlot = 0xFFFFFFFFu;

struct VfLengths {
  float size[kMaxLengths];
  float max_range[kMaxLengths];  // FilterByMaxRange bound of this length's filter; < 0: no crop
  int count;
};

// Scratch carved out of ctx->voxel for one insert launch of `num` lengths over `n` points.
struct VfTables {
  char* tables;              // [num] x { keys[capacity] (8 B), min_index[capacity] (4 B) }: one fill clears a launch's
                             // tables however many lengths it carries
  unsigned* slot;            // [num][n]
  unsigned* counters;        // [num] distinct voxels, [num] in-range points, one overflow flag; kVfCounterStride apart
  unsigned capacity;         // power of two >= 2 n
  int num;
  int packed;                // the launch's tables hold (key << 24 | min index) words (voxel_insert_kernel<true>)
};

__host__ __device__ __forceinline__ unsigned long long* table_keys(const VfTables& t, int l) {
  return reinterpret_cast<unsigned long long*>(t.tables + static_cast<size_t>(l) * t.capacity * 12);
}
__host__ __device__ __forceinline__ unsigned* table_min_index(const VfTables& t, int l) {
  return reinterpret_cast<unsigned*>(t.tables + static_cast<size_t>(l) * t.capacity * 12 + static_cast<size_t>(t.capacity) * 8);
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}

// One workgroup = 1024 consecutive points of one edge length.  The whole cloud is resident at once, so with 2 m
// voxels tens of thousands of threads would hit the same few hundred table slots in the same microsecond (and
// device-scope atomics are resolved at the memory side, one at a time per address).  The points of a workgroup are
// therefore folded in an LDS table first -- voxel key -> smallest index -- and only the owner of each LDS entry
// goes to the global table.
//
// What bounds the kernel is the chain of memory-side round trips behind the fold (round 5, timing experiments on a
// 64 x 1024 scan at 0.15 m, 47 000 voxels: 24.8 us as "look, CAS, look, atomicMin" on a 64-bit key and a 32-bit index
// word; 17.6 without the two looks; 14.7 with the CAS alone; 6.0 without the global table).  kPacked: key and index
// share ONE 64-bit word -- 13 bits per axis (|index| <= 4095: 614 m at 0.15 m), 24 bits of point index -- so that the
// first point of a voxel costs one compare-and-swap and nothing else; a later, smaller index of the same voxel one
// atomicMin without a return value on top.  A point outside 13 bits raises `unpackable` and the host repeats the launch
// with the 21-bit keys (kPacked = false: CAS on the key word, atomicMin on the index word).
constexpr int kVfInsertBlock = 1024;
constexpr int kVfCounterStride = 32;  // words between two counters
constexpr unsigned kVfLocalSlots = 2048;  // >= 2 x points: the LDS table is at most half full
constexpr unsigned kPackedIndexBits = 24, kPackedIndexMask = (1u << kPackedIndexBits) - 1u;
constexpr int kPackedAxisBits = 13;

template <bool kPacked>
__global__ __launch_bounds__(kVfInsertBlock) void voxel_insert_kernel(const float* __restrict__ x,
                                                                      const float* __restrict__ y,
                                                                      const float* __restrict__ z, unsigned n,
                                                                      VfLengths lengths, VfTables t) {
  __shared__ unsigned long long s_key[kVfLocalSlots];
  __shared__ unsigned s_min[kVfLocalSlots];   // smallest point index of the entry
  __shared__ unsigned s_slot[kVfLocalSlots];  // the entry's slot in the global table
  for (unsigned k = threadIdx.x; k < kVfLocalSlots; k += kVfInsertBlock) {
    s_key[k] = kEmptyKey;
    s_min[k] = 0xFFFFFFFFu;
  }
  const unsigned i = blockIdx.x * kVfInsertBlock + threadIdx.x;
  const int l = blockIdx.y;
  const float size = lengths.size[l];
  const float max_range = lengths.max_range[l];
  bool claimed = false, in_range = false, overflow = false, has_key = false;
  unsigned long long key = 0;
  if (i < n) {
    const float px = x[i], py = y[i], pz = z[i];
    // FilterByMaxRange: point.norm() <= max_range, Eigen's Vector3f reduction order
    in_range = max_range < 0.f || sqrtf(px * px + (py * py + pz * pz)) <= max_range;
    if (in_range) {
      const float qx = px / size, qy = py / size, qz = pz / size;
      // the rounded index stays inside 21 bits (2^20 - 1) / inside 13 bits
      const float lim = kPacked ? static_cast<float>((1 << (kPackedAxisBits - 1)) - 1) : 1048575.f;
      if (!(fabsf(qx) < lim && fabsf(qy) < lim && fabsf(qz) < lim)) {
        overflow = true;
      } else {
        constexpr int kOrigin = kPacked ? (1 << (kPackedAxisBits - 1)) : (1 << 20);
        constexpr int kShift = kPacked ? kPackedAxisBits : 21;
        const unsigned long long kx = static_cast<unsigned long long>(lround_away(qx) + kOrigin);
        const unsigned long long ky = static_cast<unsigned long long>(lround_away(qy) + kOrigin);
        const unsigned long long kz = static_cast<unsigned long long>(lround_away(qz) + kOrigin);
        key = (kx << (2 * kShift)) | (ky << kShift) | kz;
        has_key = true;
      }
    }
  }
  const unsigned long long hash = mix64(key);
  __syncthreads();
  unsigned local = 0;
  if (has_key) {
    local = static_cast<unsigned>(hash >> 32) & (kVfLocalSlots - 1);
    for (;;) {
      const unsigned long long prev = atomicCAS(&s_key[local], kEmptyKey, key);
      if (prev == kEmptyKey || prev == key) break;
      local = (local + 1) & (kVfLocalSlots - 1);
    }
    atomicMin(&s_min[local], i);
  }
  __syncthreads();
  if (has_key && s_min[local] == i) {  // this entry's first point: the only one the global table hears of
    const unsigned mask = t.capacity - 1;
    unsigned h = static_cast<unsigned>(hash) & mask;
    if (kPacked) {
      unsigned long long* words = table_keys(t, l);
      const unsigned long long mine = (key << kPackedIndexBits) | i;  // (39 key bits: never the empty word)
      for (;;) {
        const unsigned long long prev = atomicCAS(&words[h], kEmptyKey, mine);
        if (prev == kEmptyKey) {
          claimed = true;
          break;
        }
        if ((prev >> kPackedIndexBits) == key) {  // the slot's key never changes: the minimum runs over the index bits
          if (static_cast<unsigned>(prev & kPackedIndexMask) > i)
            __hip_atomic_fetch_min(&words[h], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        h = (h + 1) & mask;  // load factor <= 1/2: terminates
      }
    } else {
      unsigned long long* keys = table_keys(t, l);
      unsigned* min_index = table_min_index(t, l);
      for (;;) {
        const unsigned long long prev = atomicCAS(&keys[h], kEmptyKey, key);
        if (prev == kEmptyKey || prev == key) {
          claimed = prev == kEmptyKey;
          __hip_atomic_fetch_min(&min_index[h], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        h = (h + 1) & mask;
      }
    }
    s_slot[local] = h;
  }
  __syncthreads();
  if (i < n) t.slot[static_cast<size_t>(l) * n + i] = has_key ? s_slot[local] : kNoSlot;
  // One atomic per workgroup and counter, every counter on its own 128-byte line: these adds are resolved one at a
  // time per line, and with one add per wavefront on a shared line they WERE the kernel (49 us of 55 for six lengths).
  const int num_claimed = __syncthreads_count(claimed ? 1 : 0);
  const int num_in_range = __syncthreads_count(in_range ? 1 : 0);
  const int any_overflow = __syncthreads_or(overflow ? 1 : 0);
  if (threadIdx.x == 0) {
    if (num_claimed != 0) atomicAdd(&t.counters[kVfCounterStride * l], static_cast<unsigned>(num_claimed));
    if (max_range >= 0.f && num_in_range != 0)  // uncropped launches know the count: n
      atomicAdd(&t.counters[kVfCounterStride * (t.num + l)], static_cast<unsigned>(num_in_range));
    if (any_overflow) t.counters[kVfCounterStride * 2 * t.num] = 1u;  // kPacked: "unpackable", else: outside 21 bits
  }
}

// mode 0: survivors of the voxel filter of table `l`; mode 1: every in-range point (the adaptive
// filter's "already sparse enough" early return, voxel_filter.cc:42-45).
__global__ __launch_bounds__(kVfBlock) void voxel_flag_kernel(unsigned n, VfTables t, int l, int mode,
                                                              unsigned char* __restrict__ flags,
                                                              unsigned* __restrict__ block_counts) {
  const unsigned i = blockIdx.x * kVfBlock + threadIdx.x;
  bool keep = false;
  if (i < n) {
    const unsigned s = t.slot[static_cast<size_t>(l) * n + i];
    keep = s != kNoSlot && (mode == 1 || (t.packed ? static_cast<unsigned>(table_keys(t, l)[s] & kPackedIndexMask)
                                                               : table_min_index(t, l)[s]) == i);
    flags[i] = keep ? 1 : 0;
  }
  const int c = __syncthreads_count(keep ? 1 : 0);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = static_cast<unsigned>(c);
}

// Order-preserving scatter of the flagged points.  out_max_sq receives the bit pattern of the
// survivors' largest squared norm x*x + (y*y + z*z) (non-negative floats order like their bits).
__global__ __launch_bounds__(kVfBlock) void voxel_compact_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
    const float* __restrict__ w, unsigned n, const unsigned char* __restrict__ flags,
    const unsigned* __restrict__ block_counts, float* __restrict__ ox, float* __restrict__ oy,
    float* __restrict__ oz, float* __restrict__ ow, unsigned* __restrict__ out_index,
    unsigned* __restrict__ out_max_sq, unsigned* __restrict__ out_total) {
  __shared__ unsigned sh_part[kVfBlock];
  __shared__ unsigned sh_wave[kVfBlock / 64];
  // survivors in the blocks before this one
  unsigned part = 0;
  for (unsigned b = threadIdx.x; b < blockIdx.x; b += kVfBlock) part += block_counts[b];
  sh_part[threadIdx.x] = part;
  __syncthreads();
  for (unsigned s = kVfBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh_part[threadIdx.x] += sh_part[threadIdx.x + s];
    __syncthreads();
  }
  const unsigned base = sh_part[0];
  const unsigned i = blockIdx.x * kVfBlock + threadIdx.x;
  const bool keep = i < n && flags[i] != 0;
  const unsigned long long m = __ballot(keep);
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (lane == 0) sh_wave[wave] = static_cast<unsigned>(__popcll(m));
  __syncthreads();
  unsigned before = 0;
  for (unsigned k = 0; k < wave; ++k) before += sh_wave[k];
  // the number of survivors: the last workgroup knows it (no counter to zero, no atomic)
  if (out_total != nullptr && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    unsigned all = base;
    for (unsigned k = 0; k < kVfBlock / 64; ++k) all += sh_wave[k];
    *out_total = all;
  }
  unsigned sq_bits = 0u;
  if (keep) {
    const unsigned pos = base + before + static_cast<unsigned>(__popcll(m & ((1ull << lane) - 1ull)));
    const float px = x[i], py = y[i], pz = z[i];
    ox[pos] = px;
    oy[pos] = py;
    oz[pos] = pz;
    if (ow != nullptr) ow[pos] = w[i];
    if (out_index != nullptr) out_index[pos] = i;
    sq_bits = __float_as_uint(px * px + (py * py + pz * pz));
  }
  // the largest squared norm: one atomic per workgroup (one per survivor on the one word was 3 of this kernel's 9 us
  // on a 64 x 1024 scan: device-scope atomics on one address are resolved one at a time)
  if (out_max_sq != nullptr) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) sq_bits = max(sq_bits, static_cast<unsigned>(__shfl_xor(static_cast<int>(sq_bits), d, 64)));
    __syncthreads();  // (sh_wave's counts have been read)
    if (lane == 0) sh_wave[wave] = sq_bits;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned mx = 0u;
      for (unsigned k = 0; k < kVfBlock / 64; ++k) mx = max(mx, sh_wave[k]);
      if (mx != 0u) atomicMax(out_max_sq, mx);
    }
  }
}

static unsigned table_capacity(int64_t n) {
  unsigned c = 64;
  while (c < 2u * static_cast<unsigned>(n)) c <<= 1;
  return c;
}

struct VfScratch {
  VfTables tables[3];      // insert launches that can be alive together: first halvings, remaining halvings, bisection tree
  unsigned char* flags;
  unsigned* block_counts;
  unsigned* max_sq;        // kMaxFilters words
  int max_lengths;
};

static size_t align256(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

// `lengths`: the most edge lengths one insert launch of this call will carry.
static int carve_scratch(dliom_ctx* ctx, int64_t n, int lengths, VfScratch* s) {
  const unsigned cap = table_capacity(n);
  const size_t L = static_cast<size_t>(lengths);
  const size_t per_launch = align256(L * cap * 12) + align256(L * n * 4) + align256((2 * kMaxLengths + 1) * kVfCounterStride * 4);
  const unsigned blocks = static_cast<unsigned>((n + kVfBlock - 1) / kVfBlock);
  const size_t total = 3 * per_launch + align256(static_cast<size_t>(n)) + align256(static_cast<size_t>(blocks) * 4) + 256;
  s->max_lengths = lengths;
  DLIOM_TRY(ctx->voxel.reserve(total));
  char* p = static_cast<char*>(ctx->voxel.p);
  for (int k = 0; k < 3; ++k) {
    VfTables& t = s->tables[k];
    t.capacity = cap;
    t.num = 0;
    t.tables = p;
    p += align256(L * cap * 12);
    t.slot = reinterpret_cast<unsigned*>(p);
    p += align256(L * n * 4);
    t.counters = reinterpret_cast<unsigned*>(p);
    p += align256((2 * kMaxLengths + 1) * kVfCounterStride * 4);
  }
  s->flags = reinterpret_cast<unsigned char*>(p);
  p += align256(static_cast<size_t>(n));
  s->block_counts = reinterpret_cast<unsigned*>(p);
  p += align256(static_cast<size_t>(blocks) * 4);
  s->max_sq = reinterpret_cast<unsigned*>(p);
  return DLIOM_OK;
}

// Insert launch for `sizes`; counts[k] = survivors of VoxelFilter(sizes[k]) over the points within ranges[k]
// (< 0: every point), in_range[k] = how many those are.  Synchronises the stream (one readback of <= 260 bytes
// through pinned memory).
static int run_insert(dliom_ctx* ctx, const Soa& in, const std::vector<float>& sizes, const std::vector<float>& ranges,
                      VfTables* t, std::vector<unsigned>* counts, std::vector<unsigned>* in_range,
                      unsigned* also_zero = nullptr, int also_zero_words = 0) {
  const int num = static_cast<int>(sizes.size());
  if (num <= 0 || num > kMaxLengths || ranges.size() != sizes.size()) return DLIOM_ERR_INVALID_ARGUMENT;
  t->num = num;
  VfLengths lengths;
  lengths.count = num;
  for (int k = 0; k < num; ++k) {
    lengths.size[k] = sizes[k];
    lengths.max_range[k] = ranges[k];
  }
  const unsigned n = static_cast<unsigned>(in.n);
  unsigned* host = static_cast<unsigned*>(ctx->pinned);
  // packed words first (one atomic per voxel); a point outside their 13 bits per axis: once more with the 21-bit keys
  for (int packed = in.n <= (int64_t{1} << kPackedIndexBits) ? 1 : 0; packed >= 0; --packed) {
    t->packed = packed;
    // 0xFF bytes = empty keys, "infinite" min indices; counters start at zero
    // (one dispatch; `also_zero`: the words the emit step's compaction will atomicMax into)
    const size_t counter_bytes = static_cast<size_t>(2 * num + 1) * kVfCounterStride * 4;
    const FillJob fills[3] = {{t->tables, static_cast<size_t>(num) * t->capacity * 12, 0xFFFFFFFFu},
                              {t->counters, counter_bytes, 0u},
                              {also_zero, static_cast<size_t>(also_zero_words) * 4, 0u}};
    DLIOM_TRY(fill_multi(ctx, fills, also_zero != nullptr && also_zero_words > 0 ? 3 : 2));
    const dim3 grid((n + kVfInsertBlock - 1) / kVfInsertBlock, num);
    if (packed)
      hipLaunchKernelGGL(voxel_insert_kernel<true>, grid, dim3(kVfInsertBlock), 0, ctx->stream, in.x, in.y, in.z, n, lengths, *t);
    else
      hipLaunchKernelGGL(voxel_insert_kernel<false>, grid, dim3(kVfInsertBlock), 0, ctx->stream, in.x, in.y, in.z, n, lengths, *t);
    DLIOM_HIP_TRY(hipGetLastError());
    // the 2 num + 1 counters (kVfCounterStride words apart) packed into pinned memory by a kernel that ends in a
    // completion word: no memcpy, no full synchronise (internal.h, wait_done)
    const GatherJob job{t->counters, static_cast<unsigned>(2 * num + 1), static_cast<unsigned>(kVfCounterStride)};
    DLIOM_TRY(gather_and_wait(ctx, &job, 1, host));
    if (host[2 * num] == 0) break;
    if (!packed) return DLIOM_ERR_INVALID_ARGUMENT;  // voxel index outside 21 bits
    ++ctx->voxel_unpacked_reruns;
  }
  counts->resize(num);
  in_range->resize(num);
  for (int k = 0; k < num; ++k) {
    (*counts)[k] = host[k];
    (*in_range)[k] = host[num + k];
  }
  for (int k = 0; k < num; ++k)
    if (ranges[k] < 0.f) (*in_range)[k] = n;
  return DLIOM_OK;
}

// flag + compact of table (t, l) into caller-provided arrays (room for the survivor count the
// insert launch reported).  max_sq != nullptr: device word receiving the survivors' largest
// squared norm -- the CALLER has zeroed it (run_insert's also_zero).  No synchronisation.
static int emit_arrays(dliom_ctx* ctx, const Soa& in, const VfScratch& s, const VfTables& t, int l, int mode,
                       float* ox, float* oy, float* oz, float* ow, unsigned* max_sq) {
  const unsigned n = static_cast<unsigned>(in.n);
  const unsigned blocks = (n + kVfBlock - 1) / kVfBlock;
  hipLaunchKernelGGL(voxel_flag_kernel, dim3(blocks), dim3(kVfBlock), 0, ctx->stream, n, t, l, mode, s.flags,
                     s.block_counts);
  hipLaunchKernelGGL(voxel_compact_kernel, dim3(blocks), dim3(kVfBlock), 0, ctx->stream, in.x, in.y, in.z, in.w, n,
                     s.flags, s.block_counts, ox, oy, oz, ow, static_cast<unsigned*>(nullptr), max_sq, static_cast<unsigned*>(nullptr));
  DLIOM_HIP_TRY(hipGetLastError());
  return DLIOM_OK;
}

// ... into a new cloud of `n_out` points.
static int emit_cloud(dliom_ctx* ctx, const Soa& in, const VfScratch& s, const VfTables& t, int l, int mode,
                      int64_t n_out, dliom_cloud** out) {
  float *ox, *oy, *oz;
  DLIOM_TRY(alloc_device_cloud(ctx, n_out, out, &ox, &oy, &oz));
  float max_norm = 0.f;
  int st = DLIOM_OK;
  if (n_out > 0) {
    st = emit_arrays(ctx, Soa{in.x, in.y, in.z, nullptr, in.n}, s, t, l, mode, ox, oy, oz, nullptr, s.max_sq);
    if (st == DLIOM_OK) st = read_max_norm(ctx, s.max_sq, &max_norm);
  }
  if (st == DLIOM_OK) st = finish_device_cloud(ctx, *out, max_norm);
  if (st != DLIOM_OK) {
    dliom_cloud_destroy(*out);
    *out = nullptr;
  }
  return st;
}

// Compacts the points with flags[i] != 0 (order kept).  block_counts: scratch for n/256+1 words.
__global__ __launch_bounds__(kVfBlock) void count_flags_kernel(const unsigned char* __restrict__ flags, unsigned n,
                                                               unsigned char want,
                                                               unsigned char* __restrict__ out_flags,
                                                               unsigned* __restrict__ block_counts,
                                                               const unsigned* __restrict__ n_dev) {
  const unsigned i = blockIdx.x * kVfBlock + threadIdx.x;
  const unsigned live = n_dev != nullptr ? min(n, *n_dev) : n;  // entries behind `live` are not part of the input
  const bool keep = i < live && flags[i] == want;
  if (i < n) out_flags[i] = keep ? 1 : 0;
  const int c = __syncthreads_count(keep ? 1 : 0);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = static_cast<unsigned>(c);
}

int voxel_filter_arrays(dliom_ctx* ctx, const Soa& in, float size, float* ox, float* oy, float* oz, float* ow,
                        int64_t* n_out) {
  *n_out = 0;
  if (!(size > 0.f)) return DLIOM_ERR_INVALID_ARGUMENT;
  if (in.n == 0) return DLIOM_OK;
  VfScratch s;
  DLIOM_TRY(carve_scratch(ctx, in.n, 1, &s));
  std::vector<unsigned> counts, in_range;
  DLIOM_TRY(run_insert(ctx, in, {size}, {-1.f}, &s.tables[0], &counts, &in_range));
  *n_out = counts[0];
  return emit_arrays(ctx, in, s, s.tables[0], 0, 0, ox, oy, oz, ow, nullptr);
}

int voxel_filter_arrays_enqueue(dliom_ctx* ctx, const Soa& in, float size, float* ox, float* oy, float* oz, float* ow,
                                const unsigned** d_total, const unsigned** d_unpackable) {
  if (!(size > 0.f) || in.n <= 0) return DLIOM_ERR_INVALID_ARGUMENT;
  if (in.n > (int64_t{1} << kPackedIndexBits)) return DLIOM_ERR_CAPACITY;
  VfScratch s;
  DLIOM_TRY(carve_scratch(ctx, in.n, 1, &s));
  VfTables& t = s.tables[0];
  t.num = 1;
  t.packed = 1;
  VfLengths lengths;
  lengths.count = 1;
  lengths.size[0] = size;
  lengths.max_range[0] = -1.f;
  const size_t counter_bytes = static_cast<size_t>(2 * 1 + 1) * kVfCounterStride * 4;
  const FillJob fills[2] = {{t.tables, static_cast<size_t>(t.capacity) * 12, 0xFFFFFFFFu}, {t.counters, counter_bytes, 0u}};
  DLIOM_TRY(fill_multi(ctx, fills, 2));
  const unsigned n = static_cast<unsigned>(in.n);
  hipLaunchKernelGGL(voxel_insert_kernel<true>, dim3((n + kVfInsertBlock - 1) / kVfInsertBlock, 1), dim3(kVfInsertBlock), 0, ctx->stream,
                     in.x, in.y, in.z, n, lengths, t);
  const unsigned blocks = (n + kVfBlock - 1) / kVfBlock;
  hipLaunchKernelGGL(voxel_flag_kernel, dim3(blocks), dim3(kVfBlock), 0, ctx->stream, n, t, 0, 0, s.flags, s.block_counts);
  hipLaunchKernelGGL(voxel_compact_kernel, dim3(blocks), dim3(kVfBlock), 0, ctx->stream, in.x, in.y, in.z, in.w, n, s.flags,
                     s.block_counts, ox, oy, oz, ow, static_cast<unsigned*>(nullptr), static_cast<unsigned*>(nullptr), s.max_sq);
  DLIOM_HIP_TRY(hipGetLastError());
  *d_total = s.max_sq;
  *d_unpackable = t.counters + kVfCounterStride * 2;  // (2 * num)
  return DLIOM_OK;
}

int compact_equal_arrays(dliom_ctx* ctx, const Soa& in, const unsigned char* kinds, unsigned char want, float* ox,
                         float* oy, float* oz, int64_t* n_out, const void* also_src, unsigned also_words, void* also_dst) {
  *n_out = 0;
  if (in.n == 0) {
    if (also_src != nullptr && also_words > 0) {
      const GatherJob job{also_src, also_words};
      DLIOM_TRY(gather_and_wait(ctx, &job, 1, ctx->pinned));
      std::memcpy(also_dst, ctx->pinned, static_cast<size_t>(also_words) * 4);
    }
    return DLIOM_OK;
  }
  VfScratch s;
  DLIOM_TRY(carve_scratch(ctx, in.n, 1, &s));
  const unsigned n = static_cast<unsigned>(in.n);
  const unsigned blocks = (n + kVfBlock - 1) / kVfBlock;
  // (the survivor count comes out of the compaction's last workgroup: until round 5 a memset and one atomic per workgroup)
  hipLaunchKernelGGL(count_flags_kernel, dim3(blocks), dim3(kVfBlock), 0, ctx->stream, kinds, n, want, s.flags,
                     s.block_counts, static_cast<const unsigned*>(nullptr));
  hipLaunchKernelGGL(voxel_compact_kernel, dim3(blocks), dim3(kVfBlock), 0, ctx->stream, in.x, in.y, in.z,
                     static_cast<const float*>(nullptr), n, s.flags, s.block_counts, ox, oy, oz,
                     static_cast<float*>(nullptr), static_cast<unsigned*>(nullptr), static_cast<unsigned*>(nullptr), s.max_sq);
  DLIOM_HIP_TRY(hipGetLastError());
  unsigned* host = static_cast<unsigned*>(ctx->pinned);
  const GatherJob jobs[2] = {{s.max_sq, 1}, {also_src, also_words}};
  DLIOM_TRY(gather_and_wait(ctx, jobs, also_src != nullptr && also_words > 0 ? 2 : 1, host));
  *n_out = host[0];
  if (also_src != nullptr && also_words > 0) std::memcpy(also_dst, host + 1, static_cast<size_t>(also_words) * 4);
  return DLIOM_OK;
}

int compact_equal_arrays_enqueue(dliom_ctx* ctx, const Soa& in, const unsigned char* kinds, unsigned char want, float* ox,
                                 float* oy, float* oz, const unsigned* n_dev, const unsigned** d_total) {
  if (in.n <= 0 || n_dev == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  VfScratch s;
  DLIOM_TRY(carve_scratch(ctx, in.n, 1, &s));  // (the layout voxel_filter_arrays_enqueue had: its count is s.max_sq[0])
  if (n_dev != s.max_sq) return DLIOM_ERR_INVALID_ARGUMENT;
  const unsigned n = static_cast<unsigned>(in.n);
  const unsigned blocks = (n + kVfBlock - 1) / kVfBlock;
  hipLaunchKernelGGL(count_flags_kernel, dim3(blocks), dim3(kVfBlock), 0, ctx->stream, kinds, n, want, s.flags, s.block_counts, n_dev);
  hipLaunchKernelGGL(voxel_compact_kernel, dim3(blocks), dim3(kVfBlock), 0, ctx->stream, in.x, in.y, in.z,
                     static_cast<const float*>(nullptr), n, s.flags, s.block_counts, ox, oy, oz,
                     static_cast<float*>(nullptr), static_cast<unsigned*>(nullptr), static_cast<unsigned*>(nullptr), s.max_sq + 1);
  DLIOM_HIP_TRY(hipGetLastError());
  *d_total = s.max_sq + 1;
  return DLIOM_OK;
}

int read_max_norm(dliom_ctx* ctx, const unsigned* d_max_sq, float* max_norm) {
  unsigned* host = static_cast<unsigned*>(ctx->pinned);
  const GatherJob job{d_max_sq, 1};
  DLIOM_TRY(gather_and_wait(ctx, &job, 1, host));
  float sq;
  std::memcpy(&sq, host, 4);
  *max_norm = std::sqrt(sq);  // sqrt is monotone and correctly rounded: == max of the norms
  return DLIOM_OK;
}

int voxel_filter_cloud(dliom_ctx* ctx, const dliom_cloud& in, float size, dliom_cloud** out) {
  *out = nullptr;
  if (!(size > 0.f)) return DLIOM_ERR_INVALID_ARGUMENT;
  if (in.n == 0) {
    float *x, *y, *z;
    DLIOM_TRY(alloc_device_cloud(ctx, 0, out, &x, &y, &z));
    return finish_device_cloud(ctx, *out, 0.f);
  }
  VfScratch s;
  DLIOM_TRY(carve_scratch(ctx, in.n, 1, &s));
  std::vector<unsigned> counts, in_range;
  const Soa soa{in.d_x, in.d_y, in.d_z, nullptr, in.n};
  DLIOM_TRY(run_insert(ctx, soa, {size}, {-1.f}, &s.tables[0], &counts, &in_range, s.max_sq, 1));
  return emit_cloud(ctx, soa, s, s.tables[0], 0, 0, counts[0], out);
}

// AdaptiveVoxelFilter::Filter for `num_filters` option sets over the same cloud (the front end's high and low
// resolution filters, local_trajectory_builder_3d.cc:507-533), searched TOGETHER: every insert launch carries the
// pending edge lengths of all filters, so two filters cost the three launch + readback round trips of one.
//   round 1   max_length and its first two halvings, per filter            (voxel_filter.cc:46-58)
//   round 1b  the remaining halvings of the filters none of those decided  (rare)
//   round 2   every mid_length each filter's bisection (:63-73) can reach  (<= 16 tree nodes per filter)
//   emit      flag + compact per filter, one readback of the survivors' largest norms
int adaptive_voxel_filter_clouds(dliom_ctx* ctx, const dliom_cloud& in, const dliom_adaptive_voxel_filter_options* const* opts,
                                 int num_filters, dliom_cloud** outs) {
  if (num_filters <= 0 || num_filters > kMaxFilters) return DLIOM_ERR_INVALID_ARGUMENT;
  for (int f = 0; f < num_filters; ++f) {
    outs[f] = nullptr;
    if (!(opts[f]->max_length > 0.f)) return DLIOM_ERR_INVALID_ARGUMENT;
  }
  auto fail = [&](int st) {
    for (int f = 0; f < num_filters; ++f) {
      if (outs[f] != nullptr) dliom_cloud_destroy(outs[f]);
      outs[f] = nullptr;
    }
    return st;
  };
  if (in.n == 0) {
    for (int f = 0; f < num_filters; ++f) {
      float *x, *y, *z;
      int st = alloc_device_cloud(ctx, 0, &outs[f], &x, &y, &z);
      if (st == DLIOM_OK) st = finish_device_cloud(ctx, outs[f], 0.f);
      if (st != DLIOM_OK) return fail(st);
    }
    return DLIOM_OK;
  }
  VfScratch s;
  DLIOM_TRY(carve_scratch(ctx, in.n, kMaxTreeNodes * num_filters, &s));
  const Soa soa{in.d_x, in.d_y, in.d_z, nullptr, in.n};

  struct Node {
    float low, high, mid;
    int ok_child, fail_child;  // -1: the loop ends
  };
  struct Search {
    std::vector<float> sizes, highs;  // max_length, then every low_length of the halving loop (:55-58)
    size_t first_batch = 0;
    std::vector<unsigned> counts;     // survivors per entry of sizes (as far as counted)
    std::vector<int> table, slot;     // where entry i of sizes was counted
    unsigned in_range = 0;
    bool chosen = false;
    const VfTables* chosen_table = nullptr;
    int chosen_l = 0, chosen_mode = 0;
    unsigned chosen_count = 0;
    std::vector<Node> nodes;
    int tree_base = 0;  // first slot of this filter's nodes in the round-2 launch
  } search[kMaxFilters];

  // round 1.  The halving search almost always ends within its first two steps, and the small edge lengths are the
  // expensive ones (most distinct voxels): count max_length and two halvings first, the rest only if needed.
  std::vector<float> sizes, ranges;
  for (int f = 0; f < num_filters; ++f) {
    Search& q = search[f];
    const dliom_adaptive_voxel_filter_options& o = *opts[f];
    q.sizes.push_back(o.max_length);
    for (float high = o.max_length; high > 1e-2f * o.max_length; high /= 2.f) {
      q.highs.push_back(high);
      q.sizes.push_back(high / 2.f);
    }
    q.first_batch = std::min<size_t>(3, q.sizes.size());
    for (size_t i = 0; i < q.first_batch; ++i) {
      q.table.push_back(0);
      q.slot.push_back(static_cast<int>(sizes.size()));
      sizes.push_back(q.sizes[i]);
      ranges.push_back(o.max_range);
    }
  }
  std::vector<unsigned> counts, in_range;
  DLIOM_TRY(run_insert(ctx, soa, sizes, ranges, &s.tables[0], &counts, &in_range, s.max_sq, kMaxFilters));
  sizes.clear();
  ranges.clear();
  for (int f = 0; f < num_filters; ++f) {
    Search& q = search[f];
    const float min_points = opts[f]->min_num_points;
    q.in_range = in_range[q.slot[0]];
    for (size_t i = 0; i < q.first_batch; ++i) q.counts.push_back(counts[q.slot[i]]);
    if (static_cast<float>(q.in_range) <= min_points) {  // "already sparse enough" (:42-45)
      q.chosen = true;
      q.chosen_table = &s.tables[0];
      q.chosen_l = q.slot[0];
      q.chosen_mode = 1;
      q.chosen_count = q.in_range;
      continue;
    }
    if (static_cast<float>(q.counts[0]) >= min_points) {  // max_length is dense enough (:46-50)
      q.chosen = true;
      q.chosen_table = &s.tables[0];
      q.chosen_l = q.slot[0];
      q.chosen_count = q.counts[0];
      continue;
    }
    bool decided = false;
    for (size_t k = 1; k < q.first_batch; ++k) decided = decided || static_cast<float>(q.counts[k]) >= min_points;
    if (!decided) {
      for (size_t i = q.first_batch; i < q.sizes.size(); ++i) {
        q.table.push_back(2);
        q.slot.push_back(static_cast<int>(sizes.size()));
        sizes.push_back(q.sizes[i]);
        ranges.push_back(opts[f]->max_range);
      }
    }
  }
  if (!sizes.empty()) {  // round 1b
    if (static_cast<int>(sizes.size()) > s.max_lengths) return DLIOM_ERR_CAPACITY;
    DLIOM_TRY(run_insert(ctx, soa, sizes, ranges, &s.tables[2], &counts, &in_range));
    for (int f = 0; f < num_filters; ++f) {
      Search& q = search[f];
      for (size_t i = q.counts.size(); i < q.slot.size(); ++i) q.counts.push_back(counts[q.slot[i]]);
    }
    sizes.clear();
    ranges.clear();
  }
  // round 2: the bisection tree of every filter whose halving loop found a dense-enough low_length
  for (int f = 0; f < num_filters; ++f) {
    Search& q = search[f];
    if (q.chosen) continue;
    const float min_points = opts[f]->min_num_points;
    const VfTables* tables[3] = {&s.tables[0], &s.tables[1], &s.tables[2]};
    size_t k = 0;
    for (; k < q.highs.size() && k + 1 < q.counts.size(); ++k)
      if (static_cast<float>(q.counts[k + 1]) >= min_points) break;
    if (!(k < q.highs.size() && k + 1 < q.counts.size())) {
      // no edge length was dense enough: the last low_length's result stands (:56-57,76)
      const size_t last = q.counts.size() - 1;
      q.chosen = true;
      q.chosen_table = tables[q.table[last]];
      q.chosen_l = q.slot[last];
      q.chosen_count = q.counts[last];
      continue;
    }
    q.chosen_table = tables[q.table[k + 1]];
    q.chosen_l = q.slot[k + 1];
    q.chosen_count = q.counts[k + 1];
    // breadth-first expansion of the (low, high) states; a child is linked when it is popped
    std::vector<std::pair<float, float>> todo{{q.highs[k] / 2.f, q.highs[k]}};
    std::vector<std::pair<int, int>> origin{{-1, 0}};  // (parent node, 0 = ok branch / 1 = fail branch)
    size_t head = 0;
    while (head < todo.size()) {
      const float low = todo[head].first, high = todo[head].second;
      const std::pair<int, int> from = origin[head];
      ++head;
      if (!((high - low) / low > 1e-1f)) continue;
      if (static_cast<int>(q.nodes.size()) == kMaxTreeNodes) return DLIOM_ERR_CAPACITY;
      Node nd{low, high, (low + high) / 2.f, -1, -1};
      const int id = static_cast<int>(q.nodes.size());
      q.nodes.push_back(nd);
      if (from.first >= 0) (from.second == 0 ? q.nodes[from.first].ok_child : q.nodes[from.first].fail_child) = id;
      todo.push_back({nd.mid, high});  // candidate dense enough: low = mid
      origin.push_back({id, 0});
      todo.push_back({low, nd.mid});   // else: high = mid
      origin.push_back({id, 1});
    }
    q.tree_base = static_cast<int>(sizes.size());
    for (const Node& nd : q.nodes) {
      sizes.push_back(nd.mid);
      ranges.push_back(opts[f]->max_range);
    }
  }
  if (!sizes.empty()) {
    DLIOM_TRY(run_insert(ctx, soa, sizes, ranges, &s.tables[1], &counts, &in_range));
    for (int f = 0; f < num_filters; ++f) {
      Search& q = search[f];
      if (q.chosen || q.nodes.empty()) continue;
      const float min_points = opts[f]->min_num_points;
      for (int id = 0; id >= 0;) {
        if (static_cast<float>(counts[q.tree_base + id]) >= min_points) {
          q.chosen_table = &s.tables[1];
          q.chosen_l = q.tree_base + id;
          q.chosen_count = counts[q.tree_base + id];
          id = q.nodes[id].ok_child;
        } else {
          id = q.nodes[id].fail_child;
        }
      }
    }
  }
  // emit: both compactions are queued, then ONE readback of the largest squared norms
  bool any = false;
  for (int f = 0; f < num_filters; ++f) {
    Search& q = search[f];
    float *ox, *oy, *oz;
    int st = alloc_device_cloud(ctx, q.chosen_count, &outs[f], &ox, &oy, &oz);
    if (st == DLIOM_OK && q.chosen_count > 0) {
      st = emit_arrays(ctx, soa, s, *q.chosen_table, q.chosen_l, q.chosen_mode, ox, oy, oz, nullptr, s.max_sq + f);
      any = true;
    }
    if (st != DLIOM_OK) return fail(st);
  }
  float max_sq[kMaxFilters] = {0.f, 0.f};
  if (any) {
    unsigned* host = static_cast<unsigned*>(ctx->pinned);
    int st = DLIOM_OK;
    const GatherJob job{s.max_sq, static_cast<unsigned>(num_filters)};
    st = gather_and_wait(ctx, &job, 1, host);
    if (st != DLIOM_OK) return fail(st);
    std::memcpy(max_sq, host, 4 * num_filters);
  }
  for (int f = 0; f < num_filters; ++f) {
    // sqrt is monotone and correctly rounded: == max of the norms
    const int st = finish_device_cloud(ctx, outs[f], search[f].chosen_count > 0 ? std::sqrt(max_sq[f]) : 0.f);
    if (st != DLIOM_OK) return fail(st);
  }
  return DLIOM_OK;
}

int adaptive_voxel_filter_cloud(dliom_ctx* ctx, const dliom_cloud& in, const dliom_adaptive_voxel_filter_options& o,
                                dliom_cloud** out) {
  const dliom_adaptive_voxel_filter_options* opts[1] = {&o};
  return adaptive_voxel_filter_clouds(ctx, in, opts, 1, out);
}

}  // namespace dliom

using namespace dliom;

namespace dliom {
struct DownloadPose {
  Quat4 q;
  float t[3];
  int apply;
};
__global__ __launch_bounds__(256) void download_packed_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ z, int n, DownloadPose p,
                                                              float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float px = x[i], py = y[i], pz = z[i];
  if (p.apply) {
    float rx, ry, rz;
    rotate_point(p.q, px, py, pz, rx, ry, rz);
    px = rx + p.t[0];
    py = ry + p.t[1];
    pz = rz + p.t[2];
  }
  out[3 * i] = px;
  out[3 * i + 1] = py;
  out[3 * i + 2] = pz;
}
}  // namespace dliom

extern "C" {

int dliom_cloud_voxel_filter(dliom_ctx* ctx, const dliom_cloud* in, float size, dliom_cloud** out) {
  if (ctx == nullptr || in == nullptr || out == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  return voxel_filter_cloud(ctx, *in, size, out);
}

int dliom_cloud_adaptive_voxel_filter(dliom_ctx* ctx, const dliom_cloud* in,
                                      const dliom_adaptive_voxel_filter_options* options, dliom_cloud** out) {
  if (ctx == nullptr || in == nullptr || options == nullptr || out == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  return adaptive_voxel_filter_cloud(ctx, *in, *options, out);
}

int dliom_cloud_adaptive_voxel_filter_pair(dliom_ctx* ctx, const dliom_cloud* in,
                                           const dliom_adaptive_voxel_filter_options* first,
                                           const dliom_adaptive_voxel_filter_options* second, dliom_cloud** out_first,
                                           dliom_cloud** out_second) {
  if (ctx == nullptr || in == nullptr || first == nullptr || second == nullptr || out_first == nullptr ||
      out_second == nullptr)
    return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  const dliom_adaptive_voxel_filter_options* opts[2] = {first, second};
  dliom_cloud* outs[2] = {nullptr, nullptr};
  const int st = adaptive_voxel_filter_clouds(ctx, *in, opts, 2, outs);
  *out_first = outs[0];
  *out_second = outs[1];
  return st;
}

// dliom_cloud_download / _transformed: the points interleaved (and, for the second, moved by a float pose:
// sensor::TransformPointCloud's rotation * p + translation, point_cloud.cc:25-33, in Eigen's operation order) by ONE
// kernel that writes packed xyz straight into the context's pinned, device-visible staging block, in pieces of what that
// block holds.  Until round 6 a download was a stream synchronise, three pageable hipMemcpy and a host loop -- ~0.1 ms
// for the 300 points of a filtered cloud, and LocalTrajectoryBuilder3D's result needs three downloads a scan.
static int download_packed(const dliom_cloud* cloud, const float* pose7, float* points_xyz) {
  const size_t n = static_cast<size_t>(cloud->n);
  if (n == 0) return DLIOM_OK;
  dliom_ctx* ctx = cloud->ctx;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  const size_t room = (ctx->pinned_bytes - 8192) / 12;  // (the block's last 4 KB hold other calls' words)
  DownloadPose dp;
  dp.apply = pose7 != nullptr ? 1 : 0;
  if (pose7 != nullptr) {
    dp.q = Quat4{pose7[3], pose7[4], pose7[5], pose7[6]};
    dp.t[0] = pose7[0];
    dp.t[1] = pose7[1];
    dp.t[2] = pose7[2];
  } else {
    dp.q = Quat4{1.f, 0.f, 0.f, 0.f};
    dp.t[0] = dp.t[1] = dp.t[2] = 0.f;
  }
  float* staged = static_cast<float*>(ctx->pinned);
  for (size_t first = 0; first < n; first += room) {
    const size_t m = std::min(room, n - first);
    hipLaunchKernelGGL(download_packed_kernel, dim3(static_cast<unsigned>((m + 255) / 256)), dim3(256), 0, ctx->stream, cloud->d_x + first,
                       cloud->d_y + first, cloud->d_z + first, static_cast<int>(m), dp, staged);
    DLIOM_HIP_TRY(hipGetLastError());
    DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::memcpy(points_xyz + 3 * first, staged, m * 12);
  }
  return DLIOM_OK;
}

int dliom_cloud_download(const dliom_cloud* cloud, float* points_xyz) {
  if (cloud == nullptr || (cloud->n > 0 && points_xyz == nullptr)) return DLIOM_ERR_INVALID_ARGUMENT;
  return download_packed(cloud, nullptr, points_xyz);
}

int dliom_cloud_download_transformed(const dliom_cloud* cloud, const float pose[7], float* points_xyz) {
  if (cloud == nullptr || pose == nullptr || (cloud->n > 0 && points_xyz == nullptr)) return DLIOM_ERR_INVALID_ARGUMENT;
  return download_packed(cloud, pose, points_xyz);
}

}  // extern "C"
