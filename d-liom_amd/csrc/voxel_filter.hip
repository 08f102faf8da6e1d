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
#include <cstring>
#include <vector>

#include "device_common.h"
#include "internal.h"

namespace dliom {

int read_max_norm(dliom_ctx* ctx, const unsigned* d_max_sq, float* max_norm);

constexpr int kVfBlock = 256;
constexpr int kMaxLengths = 16;
constexpr unsigned long long kEmptyKey = ~0ull;
constexpr unsigned kNoSlot = 0xFFFFFFFFu;

struct VfLengths {
  float size[kMaxLengths];
  int count;
};

// Scratch carved out of ctx->voxel for one insert launch of `num` lengths over `n` points.
struct VfTables {
  unsigned long long* keys;  // [num][capacity]
  unsigned* min_index;       // [num][capacity]
  unsigned* slot;            // [num][n]
  unsigned* counters;        // [num] distinct voxels, then [num] = in-range points, [num+1] = overflow
  unsigned capacity;         // power of two >= 2 n
  int num;
};

__device__ __forceinline__ unsigned long long mix64(unsigned long long k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}

__global__ __launch_bounds__(kVfBlock) void voxel_insert_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ y,
                                                                const float* __restrict__ z, unsigned n,
                                                                int crop, float max_range, VfLengths lengths,
                                                                VfTables t) {
  const unsigned i = blockIdx.x * kVfBlock + threadIdx.x;
  const int l = blockIdx.y;
  const float size = lengths.size[l];
  bool claimed = false, in_range = false, overflow = false;
  if (i < n) {
    const float px = x[i], py = y[i], pz = z[i];
    // FilterByMaxRange: point.norm() <= max_range, Eigen's Vector3f reduction order
    in_range = !crop || sqrtf(px * px + (py * py + pz * pz)) <= max_range;
    unsigned my_slot = kNoSlot;
    if (in_range) {
      const float qx = px / size, qy = py / size, qz = pz / size;
      const float lim = 1048575.f;  // 2^20 - 1: the rounded index stays inside 21 bits
      if (!(fabsf(qx) < lim && fabsf(qy) < lim && fabsf(qz) < lim)) {
        overflow = true;
      } else {
        const unsigned long long kx = static_cast<unsigned long long>(lround_away(qx) + (1 << 20));
        const unsigned long long ky = static_cast<unsigned long long>(lround_away(qy) + (1 << 20));
        const unsigned long long kz = static_cast<unsigned long long>(lround_away(qz) + (1 << 20));
        const unsigned long long key = (kx << 42) | (ky << 21) | kz;
        unsigned long long* keys = t.keys + static_cast<size_t>(l) * t.capacity;
        unsigned* min_index = t.min_index + static_cast<size_t>(l) * t.capacity;
        const unsigned mask = t.capacity - 1;
        unsigned h = static_cast<unsigned>(mix64(key)) & mask;
        for (;;) {
          // look before locking: with large voxels thousands of points share a slot, and all but
          // the first few find their key in place and a smaller index already recorded
          unsigned long long prev = __builtin_nontemporal_load(&keys[h]);
          if (prev == kEmptyKey) prev = atomicCAS(&keys[h], kEmptyKey, key);
          if (prev == kEmptyKey || prev == key) {
            claimed = prev == kEmptyKey;
            if (__builtin_nontemporal_load(&min_index[h]) > i) atomicMin(&min_index[h], i);
            my_slot = h;
            break;
          }
          h = (h + 1) & mask;  // load factor <= 1/2: terminates
        }
      }
    }
    t.slot[static_cast<size_t>(l) * n + i] = my_slot;
  }
  // one atomic per wavefront and counter
  const unsigned long long mc = __ballot(claimed);
  const unsigned lane = threadIdx.x & 63u;
  if (mc != 0 && lane == static_cast<unsigned>(__ffsll(static_cast<long long>(mc)) - 1))
    atomicAdd(&t.counters[l], static_cast<unsigned>(__popcll(mc)));
  if (l == 0) {
    const unsigned long long mr = __ballot(in_range);
    if (mr != 0 && lane == static_cast<unsigned>(__ffsll(static_cast<long long>(mr)) - 1))
      atomicAdd(&t.counters[t.num], static_cast<unsigned>(__popcll(mr)));
  }
  if (overflow) t.counters[t.num + 1] = 1u;
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
    keep = s != kNoSlot && (mode == 1 || t.min_index[static_cast<size_t>(l) * t.capacity + s] == i);
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
    unsigned* __restrict__ out_max_sq) {
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
  if (keep) {
    const unsigned pos = base + before + static_cast<unsigned>(__popcll(m & ((1ull << lane) - 1ull)));
    const float px = x[i], py = y[i], pz = z[i];
    ox[pos] = px;
    oy[pos] = py;
    oz[pos] = pz;
    if (ow != nullptr) ow[pos] = w[i];
    if (out_index != nullptr) out_index[pos] = i;
    if (out_max_sq != nullptr) atomicMax(out_max_sq, __float_as_uint(px * px + (py * py + pz * pz)));
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
  unsigned* max_sq;
};

static size_t align256(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

static int carve_scratch(dliom_ctx* ctx, int64_t n, VfScratch* s) {
  const unsigned cap = table_capacity(n);
  const size_t per_launch = align256(static_cast<size_t>(kMaxLengths) * cap * 8) +
                            align256(static_cast<size_t>(kMaxLengths) * cap * 4) +
                            align256(static_cast<size_t>(kMaxLengths) * n * 4) + align256((kMaxLengths + 2) * 4);
  const unsigned blocks = static_cast<unsigned>((n + kVfBlock - 1) / kVfBlock);
  const size_t total = 3 * per_launch + align256(static_cast<size_t>(n)) + align256(static_cast<size_t>(blocks) * 4) + 256;
  DLIOM_TRY(ctx->voxel.reserve(total));
  char* p = static_cast<char*>(ctx->voxel.p);
  for (int k = 0; k < 3; ++k) {
    VfTables& t = s->tables[k];
    t.capacity = cap;
    t.num = 0;
    t.keys = reinterpret_cast<unsigned long long*>(p);
    p += align256(static_cast<size_t>(kMaxLengths) * cap * 8);
    t.min_index = reinterpret_cast<unsigned*>(p);
    p += align256(static_cast<size_t>(kMaxLengths) * cap * 4);
    t.slot = reinterpret_cast<unsigned*>(p);
    p += align256(static_cast<size_t>(kMaxLengths) * n * 4);
    t.counters = reinterpret_cast<unsigned*>(p);
    p += align256((kMaxLengths + 2) * 4);
  }
  s->flags = reinterpret_cast<unsigned char*>(p);
  p += align256(static_cast<size_t>(n));
  s->block_counts = reinterpret_cast<unsigned*>(p);
  p += align256(static_cast<size_t>(blocks) * 4);
  s->max_sq = reinterpret_cast<unsigned*>(p);
  return DLIOM_OK;
}

// Insert launch for `sizes`; counts[k] = survivors of VoxelFilter(sizes[k]); *in_range = points
// passing the crop.  Synchronises the stream (8..100-byte readback through pinned memory).
static int run_insert(dliom_ctx* ctx, const Soa& in, bool crop, float max_range,
                      const std::vector<float>& sizes, VfTables* t, std::vector<unsigned>* counts,
                      unsigned* in_range) {
  const int num = static_cast<int>(sizes.size());
  if (num <= 0 || num > kMaxLengths) return DLIOM_ERR_INVALID_ARGUMENT;
  t->num = num;
  VfLengths lengths;
  lengths.count = num;
  for (int k = 0; k < num; ++k) lengths.size[k] = sizes[k];
  // 0xFF bytes = empty keys, "infinite" min indices; counters start at zero
  DLIOM_HIP_TRY(hipMemsetAsync(t->keys, 0xFF, static_cast<size_t>(num) * t->capacity * 8, ctx->stream));
  DLIOM_HIP_TRY(hipMemsetAsync(t->min_index, 0xFF, static_cast<size_t>(num) * t->capacity * 4, ctx->stream));
  DLIOM_HIP_TRY(hipMemsetAsync(t->counters, 0, (num + 2) * 4, ctx->stream));
  const unsigned n = static_cast<unsigned>(in.n);
  const dim3 grid((n + kVfBlock - 1) / kVfBlock, num);
  hipLaunchKernelGGL(voxel_insert_kernel, grid, dim3(kVfBlock), 0, ctx->stream, in.x, in.y, in.z, n,
                     crop ? 1 : 0, max_range, lengths, *t);
  DLIOM_HIP_TRY(hipGetLastError());
  unsigned* host = static_cast<unsigned*>(ctx->pinned);
  DLIOM_HIP_TRY(hipMemcpyAsync(host, t->counters, (num + 2) * 4, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (host[num + 1] != 0) return DLIOM_ERR_INVALID_ARGUMENT;  // voxel index outside 21 bits
  counts->assign(host, host + num);
  *in_range = host[num];
  return DLIOM_OK;
}

// flag + compact of table (t, l) into caller-provided arrays (room for the survivor count the
// insert launch reported).  max_sq != nullptr: device word receiving the survivors' largest
// squared norm (zeroed here).  No synchronisation.
static int emit_arrays(dliom_ctx* ctx, const Soa& in, const VfScratch& s, const VfTables& t, int l, int mode,
                       float* ox, float* oy, float* oz, float* ow, unsigned* max_sq) {
  const unsigned n = static_cast<unsigned>(in.n);
  const unsigned blocks = (n + kVfBlock - 1) / kVfBlock;
  if (max_sq != nullptr) DLIOM_HIP_TRY(hipMemsetAsync(max_sq, 0, 4, ctx->stream));
  hipLaunchKernelGGL(voxel_flag_kernel, dim3(blocks), dim3(kVfBlock), 0, ctx->stream, n, t, l, mode, s.flags,
                     s.block_counts);
  hipLaunchKernelGGL(voxel_compact_kernel, dim3(blocks), dim3(kVfBlock), 0, ctx->stream, in.x, in.y, in.z, in.w, n,
                     s.flags, s.block_counts, ox, oy, oz, ow, static_cast<unsigned*>(nullptr), max_sq);
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
                                                               unsigned* __restrict__ total) {
  const unsigned i = blockIdx.x * kVfBlock + threadIdx.x;
  const bool keep = i < n && flags[i] == want;
  if (i < n) out_flags[i] = keep ? 1 : 0;
  const int c = __syncthreads_count(keep ? 1 : 0);
  if (threadIdx.x == 0) {
    block_counts[blockIdx.x] = static_cast<unsigned>(c);
    atomicAdd(total, static_cast<unsigned>(c));
  }
}

int voxel_filter_arrays(dliom_ctx* ctx, const Soa& in, float size, float* ox, float* oy, float* oz, float* ow,
                        int64_t* n_out) {
  *n_out = 0;
  if (!(size > 0.f)) return DLIOM_ERR_INVALID_ARGUMENT;
  if (in.n == 0) return DLIOM_OK;
  VfScratch s;
  DLIOM_TRY(carve_scratch(ctx, in.n, &s));
  std::vector<unsigned> counts;
  unsigned in_range = 0;
  DLIOM_TRY(run_insert(ctx, in, false, 0.f, {size}, &s.tables[0], &counts, &in_range));
  *n_out = counts[0];
  return emit_arrays(ctx, in, s, s.tables[0], 0, 0, ox, oy, oz, ow, nullptr);
}

int compact_equal_arrays(dliom_ctx* ctx, const Soa& in, const unsigned char* kinds, unsigned char want, float* ox,
                         float* oy, float* oz, int64_t* n_out) {
  *n_out = 0;
  if (in.n == 0) return DLIOM_OK;
  VfScratch s;
  DLIOM_TRY(carve_scratch(ctx, in.n, &s));
  const unsigned n = static_cast<unsigned>(in.n);
  const unsigned blocks = (n + kVfBlock - 1) / kVfBlock;
  DLIOM_HIP_TRY(hipMemsetAsync(s.max_sq, 0, 4, ctx->stream));
  hipLaunchKernelGGL(count_flags_kernel, dim3(blocks), dim3(kVfBlock), 0, ctx->stream, kinds, n, want, s.flags,
                     s.block_counts, s.max_sq);
  hipLaunchKernelGGL(voxel_compact_kernel, dim3(blocks), dim3(kVfBlock), 0, ctx->stream, in.x, in.y, in.z,
                     static_cast<const float*>(nullptr), n, s.flags, s.block_counts, ox, oy, oz,
                     static_cast<float*>(nullptr), static_cast<unsigned*>(nullptr), static_cast<unsigned*>(nullptr));
  DLIOM_HIP_TRY(hipGetLastError());
  unsigned* host = static_cast<unsigned*>(ctx->pinned);
  DLIOM_HIP_TRY(hipMemcpyAsync(host, s.max_sq, 4, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  *n_out = host[0];
  return DLIOM_OK;
}

int read_max_norm(dliom_ctx* ctx, const unsigned* d_max_sq, float* max_norm) {
  unsigned* host = static_cast<unsigned*>(ctx->pinned);
  DLIOM_HIP_TRY(hipMemcpyAsync(host, d_max_sq, 4, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
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
  DLIOM_TRY(carve_scratch(ctx, in.n, &s));
  std::vector<unsigned> counts;
  unsigned in_range = 0;
  const Soa soa{in.d_x, in.d_y, in.d_z, nullptr, in.n};
  DLIOM_TRY(run_insert(ctx, soa, false, 0.f, {size}, &s.tables[0], &counts, &in_range));
  return emit_cloud(ctx, soa, s, s.tables[0], 0, 0, counts[0], out);
}

int adaptive_voxel_filter_cloud(dliom_ctx* ctx, const dliom_cloud& in, const dliom_adaptive_voxel_filter_options& o,
                                dliom_cloud** out) {
  *out = nullptr;
  if (!(o.max_length > 0.f)) return DLIOM_ERR_INVALID_ARGUMENT;
  if (in.n == 0) {
    float *x, *y, *z;
    DLIOM_TRY(alloc_device_cloud(ctx, 0, out, &x, &y, &z));
    return finish_device_cloud(ctx, *out, 0.f);
  }
  VfScratch s;
  DLIOM_TRY(carve_scratch(ctx, in.n, &s));
  const Soa soa{in.d_x, in.d_y, in.d_z, nullptr, in.n};
  // launch 1: max_length and every low_length of the halving loop (voxel_filter.cc:55-58)
  std::vector<float> sizes{o.max_length};
  std::vector<float> highs;
  for (float high = o.max_length; high > 1e-2f * o.max_length; high /= 2.f) {
    highs.push_back(high);
    sizes.push_back(high / 2.f);
  }
  // The halving search almost always ends within its first two steps, and the small edge lengths
  // are the expensive ones (most distinct voxels): count max_length and the first two halvings
  // first, the rest only if none of them is dense enough.
  const size_t first_batch = std::min<size_t>(3, sizes.size());
  std::vector<unsigned> counts;
  unsigned in_range = 0;
  DLIOM_TRY(run_insert(ctx, soa, true, o.max_range, std::vector<float>(sizes.begin(), sizes.begin() + first_batch),
                       &s.tables[0], &counts, &in_range));
  const float min_points = o.min_num_points;
  if (static_cast<float>(in_range) <= min_points)  // "already sparse enough" (:42-45)
    return emit_cloud(ctx, soa, s, s.tables[0], 0, 1, in_range, out);
  if (static_cast<float>(counts[0]) >= min_points)  // max_length is dense enough (:46-50)
    return emit_cloud(ctx, soa, s, s.tables[0], 0, 0, counts[0], out);
  bool decided = false;
  for (size_t k = 1; k < first_batch; ++k) decided = decided || static_cast<float>(counts[k]) >= min_points;
  if (!decided && sizes.size() > first_batch) {
    std::vector<unsigned> more;
    unsigned dummy = 0;
    DLIOM_TRY(run_insert(ctx, soa, true, o.max_range, std::vector<float>(sizes.begin() + first_batch, sizes.end()),
                         &s.tables[2], &more, &dummy));
    counts.insert(counts.end(), more.begin(), more.end());
  }
  // table / slot of sizes[i]
  auto table_of = [&](size_t i) -> const VfTables& { return i < first_batch ? s.tables[0] : s.tables[2]; };
  auto slot_of = [&](size_t i) { return static_cast<int>(i < first_batch ? i : i - first_batch); };
  for (size_t k = 0; k < highs.size() && k + 1 < counts.size(); ++k) {
    if (!(static_cast<float>(counts[k + 1]) >= min_points)) continue;
    // launch 2: every mid_length the bisection (:63-73) can reach from (low, high)
    struct Node {
      float low, high, mid;
      int ok_child, fail_child;  // -1: the loop ends
    };
    std::vector<Node> nodes;
    // breadth-first expansion of the (low, high) states; a child is linked when it is popped
    std::vector<std::pair<float, float>> todo{{highs[k] / 2.f, highs[k]}};
    size_t head = 0;
    std::vector<std::pair<int, int>> origin{{-1, 0}};  // (parent node, 0 = ok branch / 1 = fail branch)
    while (head < todo.size()) {
      const float low = todo[head].first, high = todo[head].second;
      const std::pair<int, int> from = origin[head];
      ++head;
      if (!((high - low) / low > 1e-1f)) continue;
      if (static_cast<int>(nodes.size()) == kMaxLengths) return DLIOM_ERR_CAPACITY;
      Node nd{low, high, (low + high) / 2.f, -1, -1};
      const int id = static_cast<int>(nodes.size());
      nodes.push_back(nd);
      if (from.first >= 0) (from.second == 0 ? nodes[from.first].ok_child : nodes[from.first].fail_child) = id;
      todo.push_back({nd.mid, high});  // candidate dense enough: low = mid
      origin.push_back({id, 0});
      todo.push_back({low, nd.mid});   // else: high = mid
      origin.push_back({id, 1});
    }
    const VfTables* chosen_table = &table_of(k + 1);
    int chosen_l = slot_of(k + 1);
    unsigned chosen_count = counts[k + 1];
    if (!nodes.empty()) {
      std::vector<float> mids;
      for (const Node& nd : nodes) mids.push_back(nd.mid);
      std::vector<unsigned> mid_counts;
      unsigned dummy = 0;
      DLIOM_TRY(run_insert(ctx, soa, true, o.max_range, mids, &s.tables[1], &mid_counts, &dummy));
      for (int id = 0; id >= 0;) {
        if (static_cast<float>(mid_counts[id]) >= min_points) {
          chosen_table = &s.tables[1];
          chosen_l = id;
          chosen_count = mid_counts[id];
          id = nodes[id].ok_child;
        } else {
          id = nodes[id].fail_child;
        }
      }
    }
    return emit_cloud(ctx, soa, s, *chosen_table, chosen_l, 0, chosen_count, out);
  }
  // no edge length was dense enough: the last low_length's result stands (:56-57,76)
  const size_t last = sizes.size() - 1;
  return emit_cloud(ctx, soa, s, table_of(last), slot_of(last), 0, counts[last], out);
}

}  // namespace dliom

using namespace dliom;

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

int dliom_cloud_download(const dliom_cloud* cloud, float* points_xyz) {
  if (cloud == nullptr || (cloud->n > 0 && points_xyz == nullptr)) return DLIOM_ERR_INVALID_ARGUMENT;
  const size_t n = static_cast<size_t>(cloud->n);
  if (n == 0) return DLIOM_OK;
  std::vector<float> soa(3 * n);
  DLIOM_HIP_TRY(hipStreamSynchronize(cloud->ctx->stream));
  DLIOM_HIP_TRY(hipMemcpy(soa.data(), cloud->d_x, n * 4, hipMemcpyDeviceToHost));
  DLIOM_HIP_TRY(hipMemcpy(soa.data() + n, cloud->d_y, n * 4, hipMemcpyDeviceToHost));
  DLIOM_HIP_TRY(hipMemcpy(soa.data() + 2 * n, cloud->d_z, n * 4, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) {
    points_xyz[3 * i] = soa[i];
    points_xyz[3 * i + 1] = soa[n + i];
    points_xyz[3 * i + 2] = soa[2 * n + i];
  }
  return DLIOM_OK;
}

}  // extern "C"
