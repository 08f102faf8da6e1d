// Context, scratch memory, status strings, probability-value tables and the
// device-resident point cloud of libdliom.so.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "internal.h"

namespace dliom {

#ifdef DLIOM_EXPERIMENTS
int tuning_int(const char* name, int fallback) {
  const char* e = std::getenv(name);
  return e != nullptr ? std::atoi(e) : fallback;
}
#endif

static thread_local std::string g_last_error;

void set_last_error(const char* what, hipError_t e, const char* file, int line) {
  char buf[512];
  std::snprintf(buf, sizeof(buf), "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
  g_last_error = buf;
}

int DevBuf::reserve(size_t bytes) {
  if (bytes <= cap) return DLIOM_OK;
  size_t want = std::max(bytes, cap + cap / 2);
  want = (want + 255) & ~static_cast<size_t>(255);
  if (p != nullptr) {
    DLIOM_HIP_TRY(hipFree(p));
    p = nullptr;
    cap = 0;
  }
  DLIOM_HIP_TRY(hipMalloc(&p, want));
  cap = want;
  return DLIOM_OK;
}

void DevBuf::release() {
  if (p != nullptr) (void)hipFree(p);
  p = nullptr;
  cap = 0;
}

// max_i ||p_i|| with Eigen's Vector3f reduction order x*x + (y*y + z*z).
// sqrt is monotone and correctly rounded, so max of norms == sqrt of the max
// squared norm (real_time_correlative_scan_matcher_3d.cc:62-66).
float cloud_max_norm(const float* p, int64_t n) {
  float best = 0.f;
  for (int64_t i = 0; i < n; ++i) {
    const float x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
    const float s = x * x + (y * y + z * z);
    best = s > best ? s : best;
  }
  return std::sqrt(best);
}

// Smallest DynamicGrid bits (>= 1) whose index range [-32<<b, 32<<b) holds
// [min_index, max_index]; 9 when even bits == 8 is too small.
int needed_bits_for_cell_range(int min_index, int max_index) {
  for (int b = 1; b <= 8; ++b) {
    const int half = 32 << b;
    if (min_index >= -half && max_index < half) return b;
  }
  return 9;
}

__global__ void aos_to_soa_kernel(const float* __restrict__ aos, int64_t n, int64_t n_padded,
                                  float* __restrict__ x, float* __restrict__ y,
                                  float* __restrict__ z) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_padded) return;
  const bool in = i < n;
  x[i] = in ? aos[3 * i] : 0.f;
  y[i] = in ? aos[3 * i + 1] : 0.f;
  z[i] = in ? aos[3 * i + 2] : 0.f;
}

static int64_t pad_points(int64_t n) { return ((n + 4095) / 4096) * 4096; }

// 30-bit Morton code of the point quantised to 1/8 m (exact power of two), clamped to +-64 m.
// Only used to ORDER points so that neighbouring lanes of a wave look up neighbouring voxels;
// results never depend on it.
__device__ __forceinline__ unsigned spread3(unsigned v) {
  v &= 0x3FFu;
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
__global__ void morton_keys_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                   const float* __restrict__ z, int64_t n, unsigned* __restrict__ keys,
                                   unsigned* __restrict__ idx) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  auto q = [](float v) {
    const float c = fminf(fmaxf(v * 8.f + 512.f, 0.f), 1023.f);
    return static_cast<unsigned>(c);
  };
  keys[i] = spread3(q(x[i])) | (spread3(q(y[i])) << 1) | (spread3(q(z[i])) << 2);
  idx[i] = static_cast<unsigned>(i);
}
__global__ void gather_sorted_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                     const float* __restrict__ z, const unsigned* __restrict__ idx,
                                     int64_t n, int64_t n_padded, float* __restrict__ xs,
                                     float* __restrict__ ys, float* __restrict__ zs) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_padded) return;
  if (i < n) {
    const unsigned j = idx[i];
    xs[i] = x[j];
    ys[i] = y[j];
    zs[i] = z[j];
  } else {
    xs[i] = kPadCoordinate;  // outside every grid: reads value 0 (score kernel contract)
    ys[i] = kPadCoordinate;
    zs[i] = kPadCoordinate;
  }
}

// Cost class of every chunk of kCostChunk Morton-adjacent points (the unit of work of the LDS-box score kernel,
// score_box.h): the box a chunk needs grows with the extent of its points and, through the rotations of the search
// window, with their range.  Chunks that straddle a jump of the Morton curve need several boxes and take up to five
// times as long as a compact one; the kernel's dispenser hands chunks out in the order built here -- most expensive
// first -- so that the last tickets of a launch are short ones.  Only an ORDER of independent work items: results
// never depend on it.  CH = 32: one half-wave per chunk (the narrow box kernel's chunks); CH = 64: one wave per chunk
// (the big-box variants' chunks, round 6).
template <int CH>
__global__ void chunk_cost_kernel(const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
                                  int64_t n, int chunks, unsigned char* __restrict__ cls) {
  static_assert(CH == 32 || CH == 64, "one half-wave or one wave per chunk");
  const int lane = threadIdx.x & 63, l = lane & (CH - 1);
  const int wave_id = static_cast<int>((static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6);
  const int chunk = CH == 32 ? 2 * wave_id + (lane >> 5) : wave_id;
  const int64_t i = static_cast<int64_t>(chunk) * CH + l;
  const bool have = chunk < chunks && i < n;
  const float x = have ? xs[i] : 0.f, y = have ? ys[i] : 0.f, z = have ? zs[i] : 0.f;
  float lo[3] = {have ? x : 3.0e38f, have ? y : 3.0e38f, have ? z : 3.0e38f};
  float hi[3] = {have ? x : -3.0e38f, have ? y : -3.0e38f, have ? z : -3.0e38f};
  float r = fabsf(x) + fabsf(y) + fabsf(z);
#pragma unroll
  for (int m = CH / 2; m >= 1; m >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], m));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], m));
    }
    r = fmaxf(r, __shfl_xor(r, m));
  }
  if (l == 0 && chunk < chunks) {
    float v = 1.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) v *= fmaxf(hi[a] - lo[a], 0.f) + 0.05f * r + 0.6f;
    cls[chunk] = static_cast<unsigned char>(min(max(static_cast<int>(4.f * log2f(v)) + 24, 0), 63));
  }
}
// order[k] = the chunk handed out k-th: classes descending, chunk index ascending within a class (a stable counting
// sort, deterministic).  One workgroup of 256 threads, each owning a contiguous range of chunks.
__global__ __launch_bounds__(256) void chunk_order_kernel(const unsigned char* __restrict__ cls, int chunks,
                                                          unsigned* __restrict__ order) {
  __shared__ unsigned short hist[64][256];
  __shared__ unsigned base[64];
  const int t = threadIdx.x;
  for (int c = 0; c < 64; ++c) hist[c][t] = 0;
  const int per = (chunks + 255) / 256, first = t * per, last = min(first + per, chunks);
  for (int c = first; c < last; ++c) ++hist[cls[c]][t];
  __syncthreads();
  if (t < 64) {  // exclusive scan over the threads, per class
    unsigned run = 0;
    for (int k = 0; k < 256; ++k) {
      const unsigned v = hist[t][k];
      hist[t][k] = static_cast<unsigned short>(run);
      run += v;
    }
    base[t] = run;
  }
  __syncthreads();
  if (t == 0) {
    unsigned run = 0;
    for (int c = 63; c >= 0; --c) {
      const unsigned v = base[c];
      base[c] = run;
      run += v;
    }
  }
  __syncthreads();
  for (int c = first; c < last; ++c) {
    const int k = cls[c];
    order[base[k] + hist[k][t]++] = static_cast<unsigned>(c);
  }
}

// Device layout of a cloud inside one allocation:
//   [aos staging | x y z (input order) | xs ys zs (Morton order) | keys, idx (in/out) ]
static size_t cloud_bytes(int64_t n) {
  const size_t np = static_cast<size_t>(pad_points(n));
  return static_cast<size_t>(n) * 12 + 256 + np * 12 * 2 + np * 4 * 4 + 1024;
}

struct CloudLayout {
  float* aos;
  float *x, *y, *z, *xs, *ys, *zs;
  unsigned *keys_in, *keys_out, *idx_in, *idx_out;
};

static CloudLayout layout_cloud(char* base, int64_t n) {
  const int64_t np = pad_points(n);
  CloudLayout l;
  l.aos = reinterpret_cast<float*>(base);
  const size_t soa_off = (static_cast<size_t>(n) * 12 + 255) & ~static_cast<size_t>(255);
  l.x = reinterpret_cast<float*>(base + soa_off);
  l.y = l.x + np;
  l.z = l.y + np;
  l.xs = l.z + np;
  l.ys = l.xs + np;
  l.zs = l.ys + np;
  l.keys_in = reinterpret_cast<unsigned*>(l.zs + np);
  l.keys_out = l.keys_in + np;
  l.idx_in = l.keys_out + np;
  l.idx_out = l.idx_in + np;
  return l;
}

// Tail [n, n_padded) of the input-order arrays := 0.
__global__ void pad_tail_kernel(float* __restrict__ x, float* __restrict__ y, float* __restrict__ z, int64_t n,
                                int64_t n_padded) {
  const int64_t i = n + static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n_padded) x[i] = y[i] = z[i] = 0.f;
}
// Small clouds: the "Morton" arrays are a padded copy of the input order (a sort costs more than
// it saves).
__global__ void pad_copy_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                const float* __restrict__ z, int64_t n, int64_t n_padded,
                                float* __restrict__ xs, float* __restrict__ ys, float* __restrict__ zs) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_padded) return;
  xs[i] = i < n ? x[i] : kPadCoordinate;
  ys[i] = i < n ? y[i] : kPadCoordinate;
  zs[i] = i < n ? z[i] : kPadCoordinate;
}

// Clouds below this size keep their input order in the "Morton" arrays.
constexpr int64_t kMortonSortMinPoints = 4096;

// x, y, z [0, n) are in place on the device: pad them.  The Morton-ordered copies are built when a
// kernel that wants them first sees the cloud (ensure_morton).
static int finish_cloud(dliom_ctx* ctx, const CloudLayout& l, int64_t n, dliom_cloud* out) {
  const int64_t np = pad_points(n);
  if (np > n) {
    const int threads = 256;
    const unsigned blocks = static_cast<unsigned>((np - n + threads - 1) / threads);
    hipLaunchKernelGGL(pad_tail_kernel, dim3(blocks), dim3(threads), 0, ctx->stream, l.x, l.y, l.z, n, np);
    DLIOM_HIP_TRY(hipGetLastError());
  }
  out->ctx = ctx;
  out->device = ctx->device;
  out->n = n;
  out->n_padded = np;
  out->d_x = l.x;
  out->d_y = l.y;
  out->d_z = l.z;
  out->d_xs = l.xs;
  out->d_ys = l.ys;
  out->d_zs = l.zs;
  out->morton_ready = false;
  out->d_chunk_order = nullptr;
  out->d_chunk_order_big = nullptr;
  return DLIOM_OK;
}

int ensure_morton(dliom_ctx* ctx, const dliom_cloud* cloud) {
  dliom_cloud* c = const_cast<dliom_cloud*>(cloud);
  if (c->morton_ready) return DLIOM_OK;
  const int64_t n = c->n, np = c->n_padded;
  if (np > 0) {
    const CloudLayout l = layout_cloud(static_cast<char*>(c->base), n);
    const int threads = 256;
    const unsigned blocks = static_cast<unsigned>((np + threads - 1) / threads);
    if (n < kMortonSortMinPoints) {
      hipLaunchKernelGGL(pad_copy_kernel, dim3(blocks), dim3(threads), 0, ctx->stream, l.x, l.y, l.z, n, np, l.xs,
                         l.ys, l.zs);
    } else {
      hipLaunchKernelGGL(morton_keys_kernel, dim3(blocks), dim3(threads), 0, ctx->stream, l.x, l.y, l.z, n,
                         l.keys_in, l.idx_in);
      DLIOM_HIP_TRY(hipGetLastError());
      size_t temp_bytes = 0;
      DLIOM_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, l.keys_in, l.keys_out, l.idx_in,
                                                       l.idx_out, static_cast<int>(n), 0, 30, ctx->stream));
      DLIOM_TRY(ctx->sort_tmp.reserve(temp_bytes));
      DLIOM_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(ctx->sort_tmp.p, temp_bytes, l.keys_in, l.keys_out,
                                                       l.idx_in, l.idx_out, static_cast<int>(n), 0, 30,
                                                       ctx->stream));
      hipLaunchKernelGGL(gather_sorted_kernel, dim3(blocks), dim3(threads), 0, ctx->stream, l.x, l.y, l.z,
                         l.idx_out, n, np, l.xs, l.ys, l.zs);
      // the sort's key buffers are free again: chunk classes in keys_out, the chunk order in keys_in
      const int chunks = static_cast<int>((n + kCostChunk - 1) / kCostChunk);
      if (chunks <= 65535) {  // 16-bit counters in chunk_order_kernel (2 M points)
        unsigned char* cls = reinterpret_cast<unsigned char*>(l.keys_out);
        hipLaunchKernelGGL(chunk_cost_kernel<kCostChunk>, dim3((chunks + 7) / 8), dim3(256), 0, ctx->stream, l.xs, l.ys, l.zs, n, chunks, cls);
        DLIOM_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(chunk_order_kernel, dim3(1), dim3(256), 0, ctx->stream, cls, chunks, l.keys_in);
        c->d_chunk_order = l.keys_in;
        // ... and of the 64-point chunks of the big-box variants, behind the first table (the key buffers hold 4 bytes a
        // point; the two tables take 1/8 + 1/16 of that).  Only clouds those variants are chosen for: several passes
        // need more than 27 translations, a search that small on a small cloud never reaches the box kernel.
        const int chunks64 = static_cast<int>((n + kCostChunkBig - 1) / kCostChunkBig);
        if (n >= 16384) {
          unsigned char* cls64 = cls + ((static_cast<size_t>(chunks) + 255) & ~static_cast<size_t>(255));
          unsigned* order64 = l.keys_in + ((static_cast<size_t>(chunks) + 63) & ~static_cast<size_t>(63));
          hipLaunchKernelGGL(chunk_cost_kernel<kCostChunkBig>, dim3((chunks64 + 3) / 4), dim3(256), 0, ctx->stream, l.xs, l.ys, l.zs, n,
                             chunks64, cls64);
          DLIOM_HIP_TRY(hipGetLastError());
          hipLaunchKernelGGL(chunk_order_kernel, dim3(1), dim3(256), 0, ctx->stream, cls64, chunks64, order64);
          c->d_chunk_order_big = order64;
        }
      }
    }
    DLIOM_HIP_TRY(hipGetLastError());
  }
  c->morton_ready = true;
  return DLIOM_OK;
}

static int fill_cloud(dliom_ctx* ctx, char* base, const float* points_xyz, int64_t n,
                      dliom_cloud* out) {
  const int64_t np = pad_points(n);
  const CloudLayout l = layout_cloud(base, n);
  if (n > 0) {
    DLIOM_HIP_TRY(hipMemcpyAsync(l.aos, points_xyz, static_cast<size_t>(n) * 12,
                                 hipMemcpyHostToDevice, ctx->stream));
  }
  if (np > 0) {
    const int threads = 256;
    const unsigned blocks = static_cast<unsigned>((np + threads - 1) / threads);
    hipLaunchKernelGGL(aos_to_soa_kernel, dim3(blocks), dim3(threads), 0, ctx->stream, l.aos, n, np, l.x,
                       l.y, l.z);
  }
  DLIOM_TRY(finish_cloud(ctx, l, n, out));
  out->base = base;
  out->max_norm = cloud_max_norm(points_xyz, n);
  {
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int64_t i = 0; i < n; ++i) {
      ax = std::max(ax, std::fabs(points_xyz[3 * i]));
      ay = std::max(ay, std::fabs(points_xyz[3 * i + 1]));
      az = std::max(az, std::fabs(points_xyz[3 * i + 2]));
    }
    out->abs_max[0] = ax;
    out->abs_max[1] = ay;
    out->abs_max[2] = az;
  }
  return DLIOM_OK;
}

int stage_cloud(dliom_ctx* ctx, const float* points_xyz, int64_t n, dliom_cloud* out,
                size_t scratch_offset_bytes) {
  // All staged clouds of one call must be reserved up front by the caller
  // through a single reserve (offsets into ctx->points); here we only fill.
  char* base = static_cast<char*>(ctx->points.p) + scratch_offset_bytes;
  out->owned_by_ctx_scratch = true;
  return fill_cloud(ctx, base, points_xyz, n, out);
}

// ---- several fills / small read-backs per dispatch (internal.h) ------------------------------------------
namespace {
struct FillArgs {
  uint4* p[4];
  unsigned long long vec[4];  // 16-byte units
  unsigned* tail[4];          // the words after the last whole unit
  unsigned tail_words[4];
  unsigned value[4];
};
__global__ __launch_bounds__(256) void fill_multi_kernel(FillArgs a) {
  const int j = blockIdx.y;
  const unsigned v = a.value[j];
  const uint4 v4 = make_uint4(v, v, v, v);
  const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * 256u;
  for (unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * 256u + threadIdx.x; i < a.vec[j]; i += stride)
    a.p[j][i] = v4;
  if (blockIdx.x == 0 && threadIdx.x < a.tail_words[j]) a.tail[j][threadIdx.x] = v;
}
struct GatherArgs {
  const unsigned* src[6];
  unsigned words[6];
  unsigned offset[6];
  unsigned stride[6];
  unsigned* dst;
  unsigned* done_word;
  unsigned done_seq;
  int n;
};
__global__ __launch_bounds__(256) void gather_to_pinned_kernel(GatherArgs a) {
  for (int j = 0; j < a.n; ++j)
    for (unsigned i = threadIdx.x; i < a.words[j]; i += 256u) a.dst[a.offset[j] + i] = a.src[j][static_cast<size_t>(i) * a.stride[j]];
  if (a.done_word != nullptr) {  // every writer releases its own copies system-wide, then the word
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence_system();
      *reinterpret_cast<volatile unsigned*>(a.done_word) = a.done_seq;
    }
  }
}
}  // namespace

int fill_multi(dliom_ctx* ctx, const FillJob* jobs, int num_jobs, hipStream_t stream) {
  if (num_jobs <= 0) return DLIOM_OK;
  if (stream == nullptr) stream = ctx->stream;
  if (num_jobs > 4) return DLIOM_ERR_INVALID_ARGUMENT;
  FillArgs a;
  unsigned long long most = 1;
  for (int j = 0; j < num_jobs; ++j) {
    char* p = static_cast<char*>(jobs[j].p);
    size_t bytes = jobs[j].bytes;
    if ((reinterpret_cast<uintptr_t>(p) & 3u) != 0 || (bytes & 3u) != 0) return DLIOM_ERR_INVALID_ARGUMENT;
    // leading words up to 16-byte alignment are rare (every caller's buffers are 256-byte aligned): plain memset then
    if ((reinterpret_cast<uintptr_t>(p) & 15u) != 0) {
      const unsigned char b = static_cast<unsigned char>(jobs[j].value & 0xFFu);
      if (jobs[j].value != 0x01010101u * b) return DLIOM_ERR_INVALID_ARGUMENT;
      DLIOM_HIP_TRY(hipMemsetAsync(p, b, bytes, stream));
      bytes = 0;
    }
    a.p[j] = reinterpret_cast<uint4*>(p);
    a.vec[j] = bytes / 16;
    a.tail[j] = reinterpret_cast<unsigned*>(p + a.vec[j] * 16);
    a.tail_words[j] = static_cast<unsigned>((bytes % 16) / 4);
    a.value[j] = jobs[j].value;
    most = std::max(most, a.vec[j]);
  }
  const unsigned blocks = static_cast<unsigned>(std::min<unsigned long long>((most + 255) / 256, 2048));
  hipLaunchKernelGGL(fill_multi_kernel, dim3(blocks, num_jobs), dim3(256), 0, stream, a);
  DLIOM_HIP_TRY(hipGetLastError());
  return DLIOM_OK;
}

int gather_to_pinned(dliom_ctx* ctx, const GatherJob* jobs, int num_jobs, void* pinned_dst, hipStream_t stream, unsigned* done_word,
                     unsigned done_seq) {
  if (num_jobs <= 0) return DLIOM_OK;
  if (stream == nullptr) stream = ctx->stream;
  if (num_jobs > 6 || pinned_dst == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  GatherArgs a;
  unsigned off = 0;
  for (int j = 0; j < 6; ++j) {
    a.src[j] = nullptr;
    a.words[j] = a.offset[j] = 0;
    a.stride[j] = 1;
  }
  for (int j = 0; j < num_jobs; ++j) {
    if (jobs[j].words > 1024u) return DLIOM_ERR_INVALID_ARGUMENT;
    a.src[j] = static_cast<const unsigned*>(jobs[j].src);
    a.words[j] = jobs[j].words;
    a.stride[j] = jobs[j].stride == 0 ? 1 : jobs[j].stride;
    a.offset[j] = off;
    off += jobs[j].words;
  }
  a.dst = static_cast<unsigned*>(pinned_dst);
  a.done_word = done_word;
  a.done_seq = done_seq;
  a.n = num_jobs;
  hipLaunchKernelGGL(gather_to_pinned_kernel, dim3(1), dim3(256), 0, stream, a);
  DLIOM_HIP_TRY(hipGetLastError());
  return DLIOM_OK;
}

int wait_done(dliom_ctx* ctx, hipStream_t stream, const unsigned* done_word, unsigned done_seq, int max_poll_us) {
  if (ctx != nullptr) ++ctx->read_backs;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 1;; ++spins) {
    if (__atomic_load_n(done_word, __ATOMIC_ACQUIRE) == done_seq) return DLIOM_OK;
    if ((spins & 63u) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(max_poll_us)) break;
  }
  if (ctx != nullptr) ++ctx->poll_fallbacks;  // long kernels in front are expected; a count that grows with every call is not
  DLIOM_HIP_TRY(hipStreamSynchronize(stream));
  return __atomic_load_n(done_word, __ATOMIC_ACQUIRE) == done_seq ? DLIOM_OK : DLIOM_ERR_HIP;
}

int gather_and_wait(dliom_ctx* ctx, const GatherJob* jobs, int num_jobs, void* pinned_dst) {
  if (num_jobs <= 0) return DLIOM_OK;
  if (ctx->done_word == nullptr) {  // no completion word (allocation failed at creation): the plain way
    DLIOM_TRY(gather_to_pinned(ctx, jobs, num_jobs, pinned_dst));
    DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DLIOM_OK;
  }
  const unsigned seq = ++ctx->done_seq == 0u ? ++ctx->done_seq : ctx->done_seq;
  DLIOM_TRY(gather_to_pinned(ctx, jobs, num_jobs, pinned_dst, ctx->stream, ctx->done_word, seq));
  return wait_done(ctx, ctx->stream, ctx->done_word, seq);
}

int zero_words(dliom_ctx* ctx, unsigned** out) {
  if (!ctx->zero_words_ready) {
    DLIOM_TRY(ctx->zero_words.reserve(256));
    DLIOM_HIP_TRY(hipMemsetAsync(ctx->zero_words.p, 0, 256, ctx->stream));
    ctx->zero_words_ready = true;
  }
  *out = ctx->zero_words.as<unsigned>();
  return DLIOM_OK;
}

// Cloud allocations are pooled per device: a scan makes four clouds (raw, filtered, high, low) and
// hipMalloc/hipFree cost more than the kernels that fill them.  Blocks are power-of-two sized and
// handed back by dliom_cloud_destroy after a device synchronise (what hipFree would have done).
namespace {
struct PoolBlock {
  int device;
  size_t bytes;
  void* p;
};
std::mutex g_pool_mutex;
std::vector<PoolBlock> g_pool;
constexpr size_t kPoolMaxBlocks = 64;
}  // namespace

static int pool_alloc(int device, size_t need, void** p, size_t* bytes) {
  size_t cls = 64 * 1024;
  while (cls < need) cls <<= 1;
  {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (size_t i = 0; i < g_pool.size(); ++i)
      if (g_pool[i].device == device && g_pool[i].bytes == cls) {
        *p = g_pool[i].p;
        *bytes = cls;
        g_pool.erase(g_pool.begin() + i);
        return DLIOM_OK;
      }
  }
  DLIOM_HIP_TRY(hipMalloc(p, cls));
  *bytes = cls;
  return DLIOM_OK;
}

static void pool_free(int device, void* p, size_t bytes) {
  // nothing in flight ON THE OWNING DEVICE may still read the block (a process may drive several GPUs: the
  // current device is not necessarily the block's)
  int current = -1;
  (void)hipGetDevice(&current);
  if (current != device) (void)hipSetDevice(device);
  (void)hipDeviceSynchronize();
  if (current >= 0 && current != device) (void)hipSetDevice(current);
  {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    if (g_pool.size() < kPoolMaxBlocks) {
      g_pool.push_back(PoolBlock{device, bytes, p});
      return;
    }
  }
  (void)hipFree(p);
}

// A cloud whose points a kernel is about to write: allocates for `n` points and hands back the
// input-order arrays; finish_device_cloud() completes it once x, y, z [0, n) are written on
// ctx->stream.
int alloc_device_cloud(dliom_ctx* ctx, int64_t n, dliom_cloud** out, float** x, float** y, float** z) {
  *out = nullptr;
  void* base = nullptr;
  size_t bytes = 0;
  DLIOM_TRY(pool_alloc(ctx->device, cloud_bytes(n), &base, &bytes));
  dliom_cloud* c = new dliom_cloud;
  c->owned_by_ctx_scratch = false;
  c->base = base;
  c->base_bytes = bytes;
  c->device = ctx->device;
  c->ctx = ctx;
  c->n = n;
  const CloudLayout l = layout_cloud(static_cast<char*>(base), n);
  *x = l.x;
  *y = l.y;
  *z = l.z;
  *out = c;
  return DLIOM_OK;
}

// [0, n) := the source arrays, [n, n_padded) := 0: finish_cloud's pad_tail_kernel with the copy in front of it
__global__ void copy_pad_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz,
                                int64_t n, int64_t n_padded, float* __restrict__ x, float* __restrict__ y, float* __restrict__ z) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_padded) return;
  x[i] = i < n ? sx[i] : 0.f;
  y[i] = i < n ? sy[i] : 0.f;
  z[i] = i < n ? sz[i] : 0.f;
}

// A cloud whose points lie in arrays of the caller (written before their number was known on the host): copied in by the
// launch that pads the tail anyway.
int finish_device_cloud_from(dliom_ctx* ctx, dliom_cloud* c, float max_norm, const float* sx, const float* sy, const float* sz) {
  const CloudLayout l = layout_cloud(static_cast<char*>(c->base), c->n);
  const int64_t n = c->n, np = pad_points(n);
  if (np > 0) {
    hipLaunchKernelGGL(copy_pad_kernel, dim3(static_cast<unsigned>((np + 255) / 256)), dim3(256), 0, ctx->stream, sx, sy, sz, n, np,
                       l.x, l.y, l.z);
    DLIOM_HIP_TRY(hipGetLastError());
  }
  c->ctx = ctx;
  c->device = ctx->device;
  c->n_padded = np;
  c->d_x = l.x;
  c->d_y = l.y;
  c->d_z = l.z;
  c->d_xs = l.xs;
  c->d_ys = l.ys;
  c->d_zs = l.zs;
  c->morton_ready = false;
  c->d_chunk_order = nullptr;
  c->d_chunk_order_big = nullptr;
  c->max_norm = max_norm;
  return DLIOM_OK;
}

int finish_device_cloud(dliom_ctx* ctx, dliom_cloud* c, float max_norm) {
  const CloudLayout l = layout_cloud(static_cast<char*>(c->base), c->n);
  DLIOM_TRY(finish_cloud(ctx, l, c->n, c));
  c->max_norm = max_norm;
  return DLIOM_OK;
}

size_t staged_cloud_bytes(int64_t n) { return cloud_bytes(n); }

}  // namespace dliom

using namespace dliom;

int dliom_ctx::begin_span(int id) {
  if (!profiling || ((profiling_mask >> id) & 1u) == 0u) return -1;
  hipEvent_t ev[2];
  for (int k = 0; k < 2; ++k) {
    if (!event_pool.empty()) {
      ev[k] = event_pool.back();
      event_pool.pop_back();
    } else if (hipEventCreate(&ev[k]) != hipSuccess) {
      return -1;
    }
  }
  (void)hipEventRecord(ev[0], stream);
  spans.push_back(Span{ev[0], ev[1], id});
  return static_cast<int>(spans.size()) - 1;
}

void dliom_ctx::end_span(int span) {
  if (span < 0) return;
  (void)hipEventRecord(spans[span].b, stream);
}

int dliom_ctx::collect_spans() {
  DLIOM_HIP_TRY(hipStreamSynchronize(stream));
  for (const Span& s : spans) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      kernel_ms[s.id] += ms;
      kernel_launches[s.id] += 1;
    }
    event_pool.push_back(s.a);
    event_pool.push_back(s.b);
  }
  spans.clear();
  return DLIOM_OK;
}

extern "C" {

const char* dliom_status_string(int status) {
  switch (status) {
    case DLIOM_OK: return "ok";
    case DLIOM_ERR_INVALID_ARGUMENT: return "invalid argument (null pointer or bad size)";
    case DLIOM_ERR_HIP: return "HIP runtime error";
    case DLIOM_ERR_NO_DEVICE: return "no HIP device";
    case DLIOM_ERR_SCORE_NOT_POSITIVE: return "CHECK_GT(score, 0) failed";
    case DLIOM_ERR_WEIGHTS: return "occupied_space_weight count/positivity check failed";
    case DLIOM_ERR_GRID_EXTENT: return "grid would need more than 8 bits (CHECK_LE(new_bits, 8))";
    case DLIOM_ERR_RAY_TOO_LONG: return "ray longer than 1<<15 cells";
    case DLIOM_ERR_EMPTY_CLOUD: return "empty point cloud";
    case DLIOM_ERR_CAPACITY: return "output buffer too small";
    case DLIOM_ERR_SOLVER: return "solver failure";
    case DLIOM_ERR_DIVERGED: return "IMU window diverged (velocity or bias beyond the FailureDetection limits)";
    case DLIOM_ERR_PEER_FAILED: return "sharded match: another rank failed before the exchange";
    default: return "unknown status";
  }
}

const char* dliom_last_error(void) { return g_last_error.c_str(); }

int dliom_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static int ctx_create_common(int device_id, hipStream_t stream, bool owns, dliom_ctx** out) {
  if (out == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return DLIOM_ERR_NO_DEVICE;
  if (device_id < 0 || device_id >= n) return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipSetDevice(device_id));
  dliom_ctx* ctx = new dliom_ctx;
  ctx->device = device_id;
  if (owns) {
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      set_last_error("hipStreamCreateWithFlags", e, __FILE__, __LINE__);
      delete ctx;
      return DLIOM_ERR_HIP;
    }
    ctx->owns_stream = true;
  } else {
    ctx->stream = stream;
  }
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) ctx->num_cus = prop.multiProcessorCount;
  }
  ctx->pinned_bytes = 1 << 20;
  hipError_t e = hipHostMalloc(&ctx->pinned, ctx->pinned_bytes, hipHostMallocCoherent | hipHostMallocMapped);
  if (e != hipSuccess) {
    set_last_error("hipHostMalloc", e, __FILE__, __LINE__);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return DLIOM_ERR_HIP;
  }
  {
    void* w = nullptr;
    if (hipHostMalloc(&w, 64, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess) {  // optional: without it read-backs synchronise fully
      ctx->done_word = static_cast<unsigned*>(w);
      *ctx->done_word = 0u;
    }
  }
  *out = ctx;
  return DLIOM_OK;
}

int dliom_ctx_create(int device_id, dliom_ctx** out) {
  return ctx_create_common(device_id, nullptr, true, out);
}

int dliom_ctx_create_on_stream(int device_id, void* hip_stream, dliom_ctx** out) {
  return ctx_create_common(device_id, static_cast<hipStream_t>(hip_stream), false, out);
}

int dliom_ctx_destroy(dliom_ctx* ctx) {
  if (ctx == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& s : ctx->spans) {
    (void)hipEventDestroy(s.a);
    (void)hipEventDestroy(s.b);
  }
  for (auto& e : ctx->event_pool) (void)hipEventDestroy(e);
  if (ctx->rtcsm_state != nullptr && ctx->rtcsm_state_free != nullptr) ctx->rtcsm_state_free(ctx->rtcsm_state);
  ctx->points.release();
  ctx->cand.release();
  ctx->sums.release();
  ctx->bounds.release();
  ctx->rescore.release();
  ctx->partials.release();
  ctx->misc.release();
  ctx->sort_tmp.release();
  ctx->voxel.release();
  ctx->box_tables.release();
  ctx->box_error.release();
  ctx->zero_words.release();
  ctx->deskew_flags.release();
  ctx->box_counters.release();
  ctx->box_extents.release();
  ctx->csm_arrivals.release();
  ctx->aux_scratch.release();
  if (ctx->aux_pinned != nullptr) (void)hipHostFree(ctx->aux_pinned);
  if (ctx->aux_fork != nullptr) (void)hipEventDestroy(ctx->aux_fork);
  if (ctx->aux_stream != nullptr) (void)hipStreamDestroy(ctx->aux_stream);
  if (ctx->hist_fork != nullptr) (void)hipEventDestroy(ctx->hist_fork);
  if (ctx->hist_join != nullptr) (void)hipEventDestroy(ctx->hist_join);
  if (ctx->hist_big_stream != nullptr) (void)hipStreamDestroy(ctx->hist_big_stream);
  if (ctx->pinned != nullptr) (void)hipHostFree(ctx->pinned);
  if (ctx->done_word != nullptr) (void)hipHostFree(ctx->done_word);
  if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return DLIOM_OK;
}

int dliom_ctx_device(const dliom_ctx* ctx) { return ctx == nullptr ? -1 : ctx->device; }

int dliom_ctx_synchronize(dliom_ctx* ctx) {
  if (ctx == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return DLIOM_OK;
}

int dliom_ctx_memory_stats(const dliom_ctx* ctx, dliom_memory_stats* out) {
  if (ctx == nullptr || out == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  std::memset(out, 0, sizeof *out);
  const dliom::MemoryLedger& l = *ctx->ledger;
  out->grids = l.grids;
  out->leaf_table_bytes = l.leaf_table_bytes;
  out->leaf_pool_bytes = l.leaf_pool_bytes;
  out->mirror_bytes = l.mirror_bytes;
  out->mirror_budget_bytes = l.mirror_budget;
  out->mirrors_refused = l.mirrors_refused;
  const dliom::DevBuf* bufs[] = {&ctx->points, &ctx->cand, &ctx->sums, &ctx->bounds, &ctx->rescore, &ctx->partials, &ctx->misc,
                                 &ctx->sort_tmp, &ctx->voxel, &ctx->box_tables, &ctx->box_counters, &ctx->box_extents,
                                 &ctx->csm_arrivals, &ctx->box_error, &ctx->deskew_flags, &ctx->zero_words, &ctx->aux_scratch};
  for (const dliom::DevBuf* b : bufs) out->scratch_bytes += static_cast<int64_t>(b->cap);
  return DLIOM_OK;
}

int dliom_ctx_set_mirror_budget(dliom_ctx* ctx, int64_t bytes) {
  if (ctx == nullptr || bytes < 0) return DLIOM_ERR_INVALID_ARGUMENT;
  ctx->ledger->mirror_budget = bytes;
  return DLIOM_OK;
}

int dliom_ctx_read_backs(const dliom_ctx* ctx, int64_t* count) {
  if (ctx == nullptr || count == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  *count = ctx->read_backs;
  return DLIOM_OK;
}

int dliom_host_register(dliom_ctx* ctx, void* buffer, size_t bytes) {
  if (ctx == nullptr || buffer == nullptr || bytes == 0) return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  const hipError_t e = hipHostRegister(buffer, bytes, hipHostRegisterDefault);
  if (e != hipSuccess) {
    set_last_error("hipHostRegister", e, __FILE__, __LINE__);
    (void)hipGetLastError();  // the runtime keeps the error for the next hipGetLastError(): a later launch check must not find it
    return e == hipErrorHostMemoryAlreadyRegistered ? DLIOM_ERR_INVALID_ARGUMENT : DLIOM_ERR_HIP;
  }
  return DLIOM_OK;
}

int dliom_host_unregister(dliom_ctx* ctx, void* buffer) {
  if (ctx == nullptr || buffer == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));  // nothing of this context still reads it
  const hipError_t e = hipHostUnregister(buffer);
  if (e != hipSuccess) {
    set_last_error("hipHostUnregister", e, __FILE__, __LINE__);
    (void)hipGetLastError();
    return e == hipErrorHostMemoryNotRegistered ? DLIOM_ERR_INVALID_ARGUMENT : DLIOM_ERR_HIP;
  }
  return DLIOM_OK;
}

int dliom_ctx_poll_fallbacks(const dliom_ctx* ctx, int64_t* count) {
  if (ctx == nullptr || count == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  *count = ctx->poll_fallbacks;
  return DLIOM_OK;
}

int dliom_ctx_voxel_filter_reruns(const dliom_ctx* ctx, int64_t* count) {
  if (ctx == nullptr || count == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  *count = ctx->voxel_unpacked_reruns;
  return DLIOM_OK;
}

int dliom_ctx_set_tuning(dliom_ctx* ctx, int knob, int value) {
  if (ctx == nullptr || knob < 0 || knob >= DLIOM_TUNE_COUNT) return DLIOM_ERR_INVALID_ARGUMENT;
  switch (knob) {
    case DLIOM_TUNE_SCORE_KERNEL:
      if (value < 0 || value > 3) return DLIOM_ERR_INVALID_ARGUMENT;
      break;
    case DLIOM_TUNE_CSM_ONE_LAUNCH_MAX:
      if (value < 0 || value > 4096) return DLIOM_ERR_INVALID_ARGUMENT;
      break;
    case DLIOM_TUNE_RESERVED_TEST_HOOK:
#ifdef DLIOM_TEST_HOOKS  // libdliom_hooks.so (make hooks): 1 injects the box kernel's inconsistency word once; 2 / 3 make
      if (value < 0 || value > 3) return DLIOM_ERR_INVALID_ARGUMENT;  // the de-skew record every hit / also "fix" every record
      break;
#else
      return DLIOM_ERR_INVALID_ARGUMENT;  // the shipped library has no fault injection
#endif
    default:
      if (value != 0 && value != 1) return DLIOM_ERR_INVALID_ARGUMENT;
  }
  ctx->tuning[knob] = value;
  return DLIOM_OK;
}

int dliom_ctx_get_tuning(const dliom_ctx* ctx, int knob, int* value) {
  if (ctx == nullptr || value == nullptr || knob < 0 || knob >= DLIOM_TUNE_COUNT) return DLIOM_ERR_INVALID_ARGUMENT;
  *value = ctx->tuning[knob];
  return DLIOM_OK;
}

int dliom_ctx_set_profiling(dliom_ctx* ctx, int enabled) {
  if (ctx == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  ctx->profiling = enabled != 0;
  // enabled > 1: bit (k + 1) selects kernel id k, e.g. 2 = DLIOM_KERNEL_RTCSM_SCORE only
  ctx->profiling_mask = enabled > 1 ? static_cast<unsigned>(enabled) >> 1 : ~0u;
  return DLIOM_OK;
}

int dliom_ctx_reset_profiling(dliom_ctx* ctx) {
  if (ctx == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_TRY(ctx->collect_spans());
  for (int i = 0; i < DLIOM_KERNEL_COUNT; ++i) {
    ctx->kernel_ms[i] = 0;
    ctx->kernel_launches[i] = 0;
  }
  return DLIOM_OK;
}

int dliom_ctx_kernel_time(dliom_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches) {
  if (ctx == nullptr || kernel_id < 0 || kernel_id >= DLIOM_KERNEL_COUNT)
    return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_TRY(ctx->collect_spans());
  if (total_ms != nullptr) *total_ms = ctx->kernel_ms[kernel_id];
  if (launches != nullptr) *launches = ctx->kernel_launches[kernel_id];
  return DLIOM_OK;
}

// ---- probability value tables (host, float arithmetic as in the reference) -------
// mapping/probability_values.h:32-44,48-54 and probability_values.cc:27-36,73-83.
static inline float clampf(float v, float lo, float hi) { return v > hi ? hi : (v < lo ? lo : v); }
static const float kMinP = 0.1f;
static const float kMaxP = 1.f - 0.1f;

static inline uint16_t probability_to_value(float p) {
  const int v =
      static_cast<int>(std::lround((clampf(p, kMinP, kMaxP) - kMinP) * (32766.f / (kMaxP - kMinP)))) + 1;
  return static_cast<uint16_t>(v);
}
static inline float value_to_probability(int v) {
  if (v == 0) return kMinP;
  const float kScale = (kMaxP - kMinP) / 32766.f;
  return v * kScale + (kMinP - kScale);
}

float dliom_odds(float probability) { return probability / (1.f - probability); }

uint16_t dliom_probability_to_value(float probability) { return probability_to_value(probability); }

int dliom_compute_lookup_table_to_apply_odds(float odds, uint16_t* t) {
  if (t == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  t[0] = static_cast<uint16_t>(probability_to_value(odds / (odds + 1.f)) + 32768u);
  for (int cell = 1; cell != 32768; ++cell) {
    const float p = value_to_probability(cell);
    const float o = odds * (p / (1.f - p));
    t[cell] = static_cast<uint16_t>(probability_to_value(o / (o + 1.f)) + 32768u);
  }
  return DLIOM_OK;
}

int dliom_value_to_probability_table(float* t) {
  if (t == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  for (int v = 0; v != 32768; ++v) {
    t[v] = value_to_probability(v);
    t[v + 32768] = t[v];
  }
  return DLIOM_OK;
}

// ---- device-resident cloud --------------------------------------------------------
int dliom_cloud_create(dliom_ctx* ctx, const float* points_xyz, int64_t n, dliom_cloud** out) {
  if (ctx == nullptr || out == nullptr || n < 0 || (n > 0 && points_xyz == nullptr))
    return DLIOM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  void* base = nullptr;
  size_t bytes = 0;
  DLIOM_TRY(pool_alloc(ctx->device, staged_cloud_bytes(n), &base, &bytes));
  dliom_cloud* c = new dliom_cloud;
  c->owned_by_ctx_scratch = false;
  int s = fill_cloud(ctx, static_cast<char*>(base), points_xyz, n, c);
  c->base_bytes = bytes;
  // the host buffer may be reused by the caller as soon as we return
  if (s == DLIOM_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) s = DLIOM_ERR_HIP;
  if (s != DLIOM_OK) {
    pool_free(ctx->device, base, bytes);
    delete c;
    return s;
  }
  *out = c;
  return DLIOM_OK;
}

int dliom_cloud_destroy(dliom_cloud* cloud) {
  if (cloud == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  if (!cloud->owned_by_ctx_scratch && cloud->base != nullptr) pool_free(cloud->device, cloud->base, cloud->base_bytes);
  delete cloud;
  return DLIOM_OK;
}

int dliom_cloud_size(const dliom_cloud* cloud, int64_t* n) {
  if (cloud == nullptr || n == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  *n = cloud->n;
  return DLIOM_OK;
}

}  // extern "C"
