// RealTimeCorrelativeScanMatcher3D on gfx950.
//
// Replaces mapping/internal/3d/scan_matching/real_time_correlative_scan_matcher_3d.cc
// (:34-53 Match, :55-95 GenerateExhaustiveSearchTransforms, :97-113 ScoreCandidate).
//
// Pipeline per Match (DESIGN.md "RTCSM3D"):
//   host   window sizes, the (2A+1)^3 candidate rotations normalized(q_init * q_r) and the
//          (2L+1)^3 candidate translations q_init * t_j + t_init, in the reference's float
//          arithmetic (candidate c = j * R + r in generation order z,y,x,rz,ry,rx)
//   GPU A  score volume: exact integer sum_i max(v_i & 0x7fff, 1) per candidate
//          (rtcsm_score_dense_kernel over the grid's bricked dense mirror; the two leaf-table
//          kernels above it are the fallbacks for grids whose mirror would not fit)
//   GPU B  rigorous float-score interval per candidate from that sum; candidates whose upper
//          bound reaches the best lower bound survive
//   GPU C  survivors only: the reference's SEQUENTIAL float sum in point order, bit-identical,
//          evaluated in parallel (chunk functions + binade-wise scan; element scan and serial
//          replay kept as cross-checks)
//   host   score = sum / N * exp(-(|t| wt + angle wr)^2) as the reference computes it; first
//          strictly greater score in generation order wins.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only: the library is resolved at run time (dlopen), see rccl_api()

#include "device_common.h"
#include "host_math.h"
#include "score_box.h"

namespace dliom {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------- kernel A
// grid = (point tiles, rotation tiles), 256 threads.  Every thread keeps PPT points of its tile
// in registers (Morton order: the 64 lanes of a wave sit in one small patch of space).  A block
// walks its rotations; for each one it rotates its points once, then walks the T translations
// (uniform -> three scalar loads per translation) and looks up PPT voxels per lane in two
// batches of independent gathers (PPT leaf-table loads, then PPT voxel loads).  The per-lane
// integer sum is reduced across the wave with six DPP adds, combined across the block's four
// waves in LDS, and leaves the block as ONE atomic per candidate per rotation.
// WIDE: 64-bit leaf-table indices, for DynamicGrid bits = 8 ((8 << 8)^3 = 2^33 table entries).
template <int PPT, bool WIDE = false>
__global__ __launch_bounds__(kBlock) void rtcsm_score_kernel(
    GridView g, const float* __restrict__ px, const float* __restrict__ py,
    const float* __restrict__ pz, const float4* __restrict__ rot, int R, int r_first, int r_last,
    const float* __restrict__ trans, int T, int rots_per_block,
    unsigned long long* __restrict__ sums, int debug_no_atomic) {
  // Clouds are padded to a multiple of 4096 points with far-away points (kPadCoordinate): those
  // fall outside every grid, read 0 and add exactly 1 each -- the host subtracts the pad count.
  extern __shared__ unsigned block_sum[];  // T entries
  float x[PPT], y[PPT], z[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int i = (blockIdx.x * PPT + k) * kBlock + threadIdx.x;
    x[k] = px[i];
    y[k] = py[i];
    z[k] = pz[i];
  }
  const int r_begin = r_first + blockIdx.y * rots_per_block;  // this shard's rotations only
  const int r_end = min(r_begin + rots_per_block, r_last);
  const int lane = threadIdx.x & 63;
  const float inv = g.inv_resolution;
  typedef typename std::conditional<WIDE, unsigned long long, unsigned>::type idx_t;
  const idx_t sentinel = static_cast<idx_t>(1) << (3 * g.log2_leaves);  // table[L^3] is always 0 (null leaf)
  const unsigned lb = static_cast<unsigned>(g.log2_leaves);
  for (int r = r_begin; r < r_end; ++r) {
    for (int j = threadIdx.x; j < T; j += kBlock) block_sum[j] = 0u;
    __syncthreads();
    const float4 qq = rot[r];
    const Quat4 q{qq.x, qq.y, qq.z, qq.w};  // stored (w,x,y,z)
    float rx[PPT], ry[PPT], rz[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) rotate_point(q, x[k], y[k], z[k], rx[k], ry[k], rz[k]);
#pragma unroll 1
    for (int j = 0; j < T; ++j) {
      const float tx = trans[3 * j], ty = trans[3 * j + 1], tz = trans[3 * j + 2];
      idx_t tix[PPT];
      unsigned cel[PPT];
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const float cx = rx[k] + tx, cy = ry[k] + ty, cz = rz[k] + tz;
        bool near = false;
        int ix = cell_fast(cx, inv, &near);
        int iy = cell_fast(cy, inv, &near);
        int iz = cell_fast(cz, inv, &near);
        if (__builtin_expect(near, 0)) {  // within rounding reach of a cell boundary: exact path
          ix = cell_of(cx, g.resolution);
          iy = cell_of(cy, g.resolution);
          iz = cell_of(cz, g.resolution);
        }
        const unsigned sx = static_cast<unsigned>(ix + g.half);
        const unsigned sy = static_cast<unsigned>(iy + g.half);
        const unsigned sz = static_cast<unsigned>(iz + g.half);
        const bool inside = (sx | sy | sz) < g.grid_size;  // grid_size is a power of two
        const idx_t t = (((static_cast<idx_t>(sz >> 3) << lb) | (sy >> 3)) << lb) | (sx >> 3);
        tix[k] = inside ? t : sentinel;
        cel[k] = ((sz & 7u) << 6) | ((sy & 7u) << 3) | (sx & 7u);
      }
#pragma unroll
      for (int k = 0; k < PPT; ++k) tix[k] = g.table[tix[k]];
#pragma unroll
      for (int k = 0; k < PPT; ++k) cel[k] = g.pool[(static_cast<size_t>(tix[k]) << 9) | cel[k]];
      unsigned a = 0;
#pragma unroll
      for (int k = 0; k < PPT; ++k) a += max(cel[k] & 0x7FFFu, 1u);
      const unsigned total = wave_sum_lane63(a);
      if (lane == 63) atomicAdd(&block_sum[j], total);
    }
    __syncthreads();
    if (!debug_no_atomic) {
      for (int j = threadIdx.x; j < T; j += kBlock)
        atomicAdd(&sums[static_cast<size_t>(j) * R + r], static_cast<unsigned long long>(block_sum[j]));
    }
  }
}

// ---------------------------------------------------------------------------------- kernel A'
// Lanes = candidate ROTATIONS, points are wave-uniform.  grid = (rotation groups of 256, point
// chunks).  Each lane owns one rotation (its quaternion stays in 4 VGPRs) and TC integer
// accumulators, one per translation; the block streams its chunk of points through scalar
// registers (P at a time), rotates each point per lane and looks up the TC translated positions.
// Why: the 64 lanes of a gather now differ only by a few milliradians of rotation of the SAME
// point, so they fall into a handful of neighbouring voxels -- a gather touches ~4 cache lines
// instead of ~40, which removes the texture-addresser bound of the points-per-lane mapping
// (profiles/).  There is no cross-lane reduction at all: a lane's accumulators ARE its
// candidates' partial sums, added to the score volume with 64-lane contiguous atomics.
template <int TC, int P>
__global__ __launch_bounds__(kBlock) void rtcsm_score_rot_kernel(
    GridView g, const float* __restrict__ px, const float* __restrict__ py,
    const float* __restrict__ pz, int points_per_chunk, const float4* __restrict__ rot, int R,
    int r_first, int r_last, const float4* __restrict__ trans4, int T,
    unsigned long long* __restrict__ sums) {
  // LDS: the T translations (read as wave-wide broadcasts) and one accumulator column per lane
  // (TC rows): `lds_acc[jj][lane]` is touched by that lane only -- no conflicts, no barriers.
  extern __shared__ float4 lds_trans[];
  __shared__ unsigned lds_acc[TC][kBlock];
  for (int j = threadIdx.x; j < T; j += kBlock) lds_trans[j] = trans4[j];
  __syncthreads();
  const int r = r_first + blockIdx.x * kBlock + threadIdx.x;
  const bool active = r < r_last;
  const float4 qq = rot[active ? r : r_first];
  const Quat4 q{qq.x, qq.y, qq.z, qq.w};  // stored (w,x,y,z)
  const float inv = g.inv_resolution;
  const unsigned sentinel = 1u << (3 * g.log2_leaves);  // table[L^3] is always 0 (null leaf)
  const unsigned lb = static_cast<unsigned>(g.log2_leaves);
  const int p_begin = blockIdx.y * points_per_chunk;
  const int p_end = p_begin + points_per_chunk;  // the cloud is padded: no tail handling
  for (int jc = 0; jc < T; jc += TC) {
    const int tc = min(TC, T - jc);
    for (int jj = 0; jj < tc; ++jj) lds_acc[jj][threadIdx.x] = 0u;
#pragma unroll 1
    for (int i = p_begin; i < p_end; i += P) {
      float rx[P], ry[P], rz[P];
#pragma unroll
      for (int k = 0; k < P; ++k) rotate_point(q, px[i + k], py[i + k], pz[i + k], rx[k], ry[k], rz[k]);
#pragma unroll 1
      for (int jj = 0; jj < tc; ++jj) {
        const float4 t = lds_trans[jc + jj];  // same address in every lane: LDS broadcast
        unsigned tix[P], cel[P];
#pragma unroll
        for (int k = 0; k < P; ++k) {
          const float cx = rx[k] + t.x, cy = ry[k] + t.y, cz = rz[k] + t.z;
          bool near = false;
          int ix = cell_fast(cx, inv, &near);
          int iy = cell_fast(cy, inv, &near);
          int iz = cell_fast(cz, inv, &near);
          if (__builtin_expect(near, 0)) {  // within rounding reach of a cell boundary: exact path
            ix = cell_of(cx, g.resolution);
            iy = cell_of(cy, g.resolution);
            iz = cell_of(cz, g.resolution);
          }
          const unsigned sx = static_cast<unsigned>(ix + g.half);
          const unsigned sy = static_cast<unsigned>(iy + g.half);
          const unsigned sz = static_cast<unsigned>(iz + g.half);
          const bool inside = (sx | sy | sz) < g.grid_size;  // grid_size is a power of two
          const unsigned tt = (((sz >> 3) << lb | (sy >> 3)) << lb) | (sx >> 3);
          tix[k] = inside ? tt : sentinel;
          cel[k] = ((sz & 7u) << 6) | ((sy & 7u) << 3) | (sx & 7u);
        }
#pragma unroll
        for (int k = 0; k < P; ++k) tix[k] = g.table[tix[k]];
#pragma unroll
        for (int k = 0; k < P; ++k) cel[k] = g.pool[(tix[k] << 9) | cel[k]];
        unsigned a = 0;
#pragma unroll
        for (int k = 0; k < P; ++k) a += max(cel[k] & 0x7FFFu, 1u);
        atomicAdd(&lds_acc[jj][threadIdx.x], a);  // ds_add_u32, own column
      }
    }
    if (active) {
      for (int jj = 0; jj < tc; ++jj)
        atomicAdd(&sums[static_cast<size_t>(jc + jj) * R + r],
                  static_cast<unsigned long long>(lds_acc[jj][threadIdx.x]));
    }
  }
}

// ---------------------------------------------------------------------------------- kernel A''
// The same rotation-per-lane walk over the DENSE MIRROR of the grid (grid.hip::ensure_dense):
// one load per lookup, no leaf table, no leaf/cell bit surgery, and the mirror already holds
// max(value, 1), the quantity that is summed.
//   z  = fma(c, 1/res, K)    K = half + 1 + 1/2: shift into the mirror's [0, S) range AND the
//                            half of "floor(x + 1/2)" in the fma's single rounding
//   i' = v_cvt_flr_i32_f32(clamp(z, 0, S - 1/2))   guard cells 0 and S-1 read 1 = "outside/unknown"
//   near <=> some frac(z_c) <= band or >= 1 - band, band = 2.2 S 2^-24 (then: exact path)
//   byte offset = X[ix'] + Y[iy'] + Z[iz']  (LDS tables of the bricked layout, see the kernel)
// Error budget, in cells, with q = c/res the real quotient (|q| <= S/2 inside the grid, z <= S):
//   fl(1/res) relative error 2^-24        -> |c fl(1/res) - q| <= |q| 2^-24     <= (S/2) 2^-24
//   one rounding of the fma               -> <= ulp(z)/2 <= z 2^-24              <=  S    2^-24
//   the reference's lround(fl(q)) can only differ from floor(q + 1/2) when q is within |q| 2^-24
//   of a half-integer (its own rounding of the quotient, and the half-away tie rule)
//                                                                                <= (S/2) 2^-24
// so when z is farther than (2 S + 1) 2^-24 from every integer, floor(z) and the reference agree;
// 2.2 S 2^-24 is used (~6.7e-5 cells at S = 514).  Points far outside the grid break the bounds
// but land in a guard cell on both paths.  The exact path is taken per point (a wave-level branch):
// ~10 % of the 4-point iterations have a near lookup in some lane, usually in one point only.
__device__ __forceinline__ int cvt_flr(float y) {  // floor to int in one instruction
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(y));
  return r;
}

constexpr int kDenseMaxBlock = 256;

// Domain of the per-axis offset tables.  CLAMP = true: [0, S), every lookup is clamped into it
// (v_med3_f32).  CLAMP = false: the host proved that every lookup of this launch -- real points
// within max ||p|| + max |t| of the origin, padding replaced by `pad`, a point that stays beyond the
// grid's edge under every candidate -- lands in [lo, lo + size); entries outside [0, S) alias the
// guard cells, so the three clamps per lookup disappear.
struct DenseDomain {
  int lo;     // mirror coordinate of table entry 0 (<= 0)
  int size;   // entries per axis
  int n_real; // points below this index are real, the rest of the padded arrays is replaced by pad
  float pad[3];
};

// DEBUG != 0 are TIMING-ONLY variants (wrong sums) that isolate one pipe each (DLIOM_SCORE_DEBUG,
// profiles/r2_score_pipe_experiment.json): 1 = all index math and LDS table reads, the gather replaced
// by a register value; 2 = the gather with the index math done once per point instead of once per
// (point, translation); 3 = index math only (no LDS table reads, no gather).
template <int P, bool CLAMP, int DEBUG = 0>
__global__ __launch_bounds__(kDenseMaxBlock) void rtcsm_score_dense_kernel(
    GridView g, DenseDomain dom, const float* __restrict__ px, const float* __restrict__ py,
    const float* __restrict__ pz, int points_per_chunk, int point_chunks,
    int rot_groups, const float4* __restrict__ rot, int R, int r_first, int r_last,
    const float4* __restrict__ trans4, int T, int t_chunk, unsigned long long* __restrict__ sums) {
  // XCD-aware block -> (point chunk, rotation group) map.  Workgroup b lands on XCD b % 8 (observed,
  // used for speed only): the rotation groups of one point chunk run back to back on ONE XCD and
  // share its L2 lines; chunks are dealt to the XCDs round-robin -- giving every XCD one contiguous
  // Morton range of the cloud instead measured 8 % slower (all CUs of an XCD then hammer the few L2
  // channels that hold one compact region of the mirror).
  const int bs = blockDim.x;  // 64..256, picked by the host to waste the fewest lanes on R % bs
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rot_group = slot % rot_groups;
  const int chunk_id = (slot / rot_groups) * 8 + xcd;
  if (chunk_id >= point_chunks) return;
  extern __shared__ float4 lds_dyn[];  // [T translations | 3 TS byte offsets | t_chunk x blockDim accumulators]
  float4* lds_trans = lds_dyn;
  unsigned* lds_off = reinterpret_cast<unsigned*>(lds_dyn + T);
  const int S = g.dense_stride;
  const int TS = dom.size;  // table entries per axis (S when clamping)
  unsigned* lds_acc = lds_off + 3 * TS;
  for (int j = threadIdx.x; j < T; j += bs) lds_trans[j] = trans4[j];
  // The mirror is stored in 4 x 4 x 4 bricks of 128 B (one cache line): lanes of a wave look up
  // cells a few voxels apart in EVERY direction, and the texture addresser works per distinct line.
  // Per-axis byte offsets come from LDS tables, so the brick swizzle costs no VALU:
  //   offset(x, y, z) = X[x] + Y[y] + Z[z],  X[x] = (x>>2) 128 + (x&3) 2,
  //   Y[y] = (y>>2) B 128 + (y&3) 8,  Z[z] = (z>>2) B^2 128 + (z&3) 32,  B = bricks per axis
  {
    const unsigned B = static_cast<unsigned>(g.dense_bricks);
    for (int c = threadIdx.x; c < TS; c += bs) {
      const unsigned m = static_cast<unsigned>(min(max(c + dom.lo, 0), S - 1));  // outside -> guard cell
      const unsigned h = m >> 2, l = m & 3u;
      lds_off[c] = h * 128u + l * 2u;
      lds_off[TS + c] = h * B * 128u + l * 8u;
      lds_off[2 * TS + c] = h * B * B * 128u + l * 32u;
    }
  }
  __syncthreads();
  const int r = r_first + rot_group * bs + threadIdx.x;
  const bool active = r < r_last;
  const float4 qq = rot[active ? r : r_first];
  const Quat4 q{qq.x, qq.y, qq.z, qq.w};  // stored (w,x,y,z)
  const float inv = g.inv_resolution;
  const float K = static_cast<float>(g.half + 1 - dom.lo) + 0.5f;  // exact: a small integer + 1/2
  const float band = 2.2f * static_cast<float>(TS) * 5.9604645e-8f;
  const float lim = static_cast<float>(TS - 1) + 0.5f;
  const int to_table = g.half + 1 - dom.lo;
  const int p_begin = chunk_id * points_per_chunk;
  const int p_end = p_begin + points_per_chunk;  // the cloud is padded: no tail handling
  // gridDim.y > 1 (small searches): the translation slices are spread over workgroups as well -- a search of a few
  // hundred points fills a fraction of the chip, and each of a workgroup's steps is one exposed gather latency
  for (int jc = blockIdx.y * t_chunk; jc < T; jc += t_chunk * gridDim.y) {
    const int tc = min(t_chunk, T - jc);
    for (int jj = 0; jj < tc; ++jj) lds_acc[jj * bs + threadIdx.x] = 0u;
#pragma unroll 1
    for (int i = p_begin; i < p_end; i += P) {
      float rx[P], ry[P], rz[P];
#pragma unroll
      for (int k = 0; k < P; ++k) {
        // wave-uniform point (scalar loads); without clamps the padding is swapped for dom.pad
        const bool real = CLAMP || i + k < dom.n_real;
        rotate_point(q, real ? px[i + k] : dom.pad[0], real ? py[i + k] : dom.pad[1], real ? pz[i + k] : dom.pad[2],
                     rx[k], ry[k], rz[k]);
      }
      if (DEBUG == 2) {
        unsigned o[P];
#pragma unroll
        for (int k = 0; k < P; ++k) {
          const float4 t = lds_trans[jc];
          o[k] = lds_off[cvt_flr(__builtin_fmaf(rx[k] + t.x, inv, K))] +
                 lds_off[TS + cvt_flr(__builtin_fmaf(ry[k] + t.y, inv, K))] +
                 lds_off[2 * TS + cvt_flr(__builtin_fmaf(rz[k] + t.z, inv, K))];
        }
#pragma unroll 1
        for (int jj = 0; jj < tc; ++jj) {
          const unsigned d = (jj % 3) * 2u + ((jj / 3) % 3) * 8u + ((jj / 9) % 3) * 32u;  // uniform, stays inside a brick or the next
          unsigned a = 0;
#pragma unroll
          for (int k = 0; k < P; ++k)
            a += *reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(g.dense) + (o[k] + d));
          atomicAdd(&lds_acc[jj * bs + threadIdx.x], a);
        }
        continue;
      }
#pragma unroll 1
      for (int jj = 0; jj < tc; ++jj) {
        const float4 t = lds_trans[jc + jj];  // same address in every lane: LDS broadcast
        unsigned ox[P], oy[P], oz[P];  // byte offsets of the three clamped mirror coordinates
        float lo[P], hi[P];            // per point: min / max over the axes of frac(z)
#pragma unroll
        for (int k = 0; k < P; ++k) {
          // z = c / res + (half + 1) + 1/2 in ONE rounding; floor(z) is the mirror coordinate
          const float zx = __builtin_fmaf(rx[k] + t.x, inv, K), zy = __builtin_fmaf(ry[k] + t.y, inv, K),
                      zz = __builtin_fmaf(rz[k] + t.z, inv, K);
          const float fx = __builtin_amdgcn_fractf(zx), fy = __builtin_amdgcn_fractf(zy),
                      fz = __builtin_amdgcn_fractf(zz);
          lo[k] = fminf(fminf(fx, fy), fz);
          hi[k] = fmaxf(fmaxf(fx, fy), fz);
          if (CLAMP) {
            // clamp in float (one v_med3_f32): floor(clamp(z, 0, TS - 1/2)) == clamp(floor(z), 0, TS - 1)
            ox[k] = lds_off[cvt_flr(__builtin_amdgcn_fmed3f(zx, 0.f, lim))];
            oy[k] = lds_off[TS + cvt_flr(__builtin_amdgcn_fmed3f(zy, 0.f, lim))];
            oz[k] = lds_off[2 * TS + cvt_flr(__builtin_amdgcn_fmed3f(zz, 0.f, lim))];
          } else if (DEBUG == 3) {
            ox[k] = cvt_flr(zx);
            oy[k] = cvt_flr(zy) << 3;
            oz[k] = cvt_flr(zz) << 6;
          } else {
            ox[k] = lds_off[cvt_flr(zx)];
            oy[k] = lds_off[TS + cvt_flr(zy)];
            oz[k] = lds_off[2 * TS + cvt_flr(zz)];
          }
        }
        float m = lo[0], M = hi[0];
#pragma unroll
        for (int k = 1; k < P; ++k) {
          m = fminf(m, lo[k]);
          M = fmaxf(M, hi[k]);
        }
        if (__builtin_expect(m <= band || M >= 1.f - band, 0)) {
          // some lookup of this lane sits within rounding reach of a cell boundary: the exact
          // division path, but only for the points that need it (one wave-level branch per point)
#pragma unroll
          for (int k = 0; k < P; ++k) {
            if (lo[k] <= band || hi[k] >= 1.f - band) {
              ox[k] = lds_off[min(max(cell_of(rx[k] + t.x, g.resolution) + to_table, 0), TS - 1)];
              oy[k] = lds_off[TS + min(max(cell_of(ry[k] + t.y, g.resolution) + to_table, 0), TS - 1)];
              oz[k] = lds_off[2 * TS + min(max(cell_of(rz[k] + t.z, g.resolution) + to_table, 0), TS - 1)];
            }
          }
        }
        unsigned v[P];
#pragma unroll
        for (int k = 0; k < P; ++k)
          v[k] = DEBUG == 1 || DEBUG == 3
                     ? ((ox[k] + oy[k] + oz[k]) & 0x7fffu)
                     : *reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(g.dense) +
                                                                (ox[k] + oy[k] + oz[k]));
        unsigned a = 0;
#pragma unroll
        for (int k = 0; k < P; ++k) a += v[k];  // the mirror stores max(value, 1) already
        atomicAdd(&lds_acc[jj * bs + threadIdx.x], a);  // ds_add_u32, own column
      }
    }
    if (active) {
      for (int jj = 0; jj < tc; ++jj)
        atomicAdd(&sums[static_cast<size_t>(jc + jj) * R + r],
                  static_cast<unsigned long long>(lds_acc[jj * bs + threadIdx.x]));
    }
  }
}

// ---------------------------------------------------------------------------------- kernel B
struct BoundParams {
  double a;        // (double)kScale
  double b;        // (double)(kMin - kScale)
  double delta;    // max |LUT[v] - (a v + b)|, v' convention (0 -> 1)
  double wt, wr;   // penalty weights
  int n;           // points
  int n_pad;       // padding points, each of which added exactly 1 to every sum
  int R, T;
  int r_first, r_last;  // rotations owned by this shard
};

// Upper bound of the accumulated rounding error of the reference's sequential float sum.
// Each addition is off by at most half an ulp of its result (round to nearest) and the partial
// sums s_m are monotone, so the error after m additions is <= m 2^-24 s_m; with every addend
// <= 0.9000001 this gives s_m <= 0.9000001 m / (1 - m 2^-24) <= 0.93 m for m <= 2^19.  Together
// with s_m <= U (the final sum's upper bound) the (i+1)-th result is <= min(0.93 (i+1), U).
__device__ double seq_sum_error_bound(int n, double U) {
  double err = 0.0;
  long long i = 0;  // indices [0, i) are already accounted for
  // binade k: results in [2^k, 2^(k+1)) have ulp 2^(k-23).  Index i (the (i+1)-th addition) has
  // its result bounded by B_i = min(0.93 (i+1), U).
  for (int k = -4; k < 64 && i < n; ++k) {
    const double top = ldexp(1.0, k + 1);
    long long last;  // indices [i, last) are charged this binade's ulp
    if (U < top) {
      last = n;  // the cap keeps every remaining result below `top`
    } else {
      // 0.93 (idx+1) < top  <=>  idx + 1 < top / 0.93 ; one index fewer guards the division's rounding
      last = static_cast<long long>(floor(top / 0.93)) - 1;
      if (last > n) last = n;
    }
    if (last > i) {
      err += 0.5 * ldexp(1.0, k - 23) * static_cast<double>(last - i);
      i = last;
    }
  }
  return err;
}

__device__ __forceinline__ float next_up(float f) { return __uint_as_float(__float_as_uint(f) + 1u); }
__device__ __forceinline__ float next_down(float f) {
  const unsigned u = __float_as_uint(f);
  return u == 0u ? 0.f : __uint_as_float(u - 1u);
}

__global__ void rtcsm_bounds_kernel(const unsigned long long* __restrict__ sums, long long C,
                                    const float* __restrict__ t_norm, const float* __restrict__ r_angle,
                                    BoundParams p, float* __restrict__ lo, float* __restrict__ hi,
                                    unsigned* __restrict__ best_lo_bits) {
  const long long c = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool in_range = c < C;
  const int j = in_range ? static_cast<int>(c / p.R) : 0, r = in_range ? static_cast<int>(c % p.R) : 0;
  const bool mine = in_range && r >= p.r_first && r < p.r_last;
  float flo = 0.f, fhi = -1.f;  // another shard's candidate: never a survivor, no lower bound
  if (mine) {
  const double s = static_cast<double>(sums[c] - static_cast<unsigned long long>(p.n_pad));
  const double mid = p.a * s + p.b * p.n;
  const double slack = p.delta * p.n + 1e-9 * mid;
  double sum_lo = mid - slack, sum_hi = mid + slack;
  const double U = sum_hi * (1.0 + 1.0 / 32.0);
  const double e = seq_sum_error_bound(p.n, U);
  if (sum_hi + e > U || p.n > (1 << 19)) {  // bound not applicable: keep the candidate
    flo = 0.f;
    fhi = 3.0e38f;
  } else {
    sum_lo -= e;
    sum_hi += e;
    // score = float(double(float(sum)/float(N)) * exp(-(|t| wt + angle wr)^2)); every step is
    // monotone in sum, so outward-rounded endpoints bracket the reference's value.
    const double arg = static_cast<double>(t_norm[j]) * p.wt + static_cast<double>(r_angle[r]) * p.wr;
    const double pen = exp(-(arg * arg));
    const float nf = static_cast<float>(p.n);
    float qlo = next_down(next_down(static_cast<float>(sum_lo))) / nf;
    float qhi = next_up(next_up(static_cast<float>(sum_hi))) / nf;
    qlo = next_down(qlo);
    qhi = next_up(qhi);
    flo = next_down(next_down(static_cast<float>(static_cast<double>(qlo) * pen * (1.0 - 1e-9))));
    fhi = next_up(next_up(static_cast<float>(static_cast<double>(qhi) * pen * (1.0 + 1e-9))));
    if (!(flo > 0.f)) flo = 0.f;
  }
  }
  if (in_range) {
    lo[c] = flo;
    hi[c] = fhi;
  }
  // best lower bound: the wavefront's maximum first (non-negative floats order like their bits), ONE atomic per
  // wavefront -- one per candidate on this one word was most of the kernel (13 us for 35 937 candidates)
  unsigned best = __float_as_uint(flo);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) best = max(best, static_cast<unsigned>(__shfl_xor(static_cast<int>(best), off, 64)));
  if ((threadIdx.x & 63) == 0 && best != 0u) atomicMax(best_lo_bits, best);
}

__global__ void rtcsm_select_kernel(const float* __restrict__ hi, long long C,
                                    const unsigned* __restrict__ best_lo_bits,
                                    unsigned* __restrict__ count, unsigned* __restrict__ list) {
  const long long c = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float best_lo = __uint_as_float(*best_lo_bits);
  if (hi[c] >= best_lo) {
    const unsigned k = atomicAdd(count, 1u);
    list[k] = static_cast<unsigned>(c);
  }
}

// ---------------------------------------------------------------------------------- kernel C
// One workgroup per surviving candidate.  Waves 1-3 turn tiles of points (INPUT order) into
// probabilities in LDS -- ValueToProbability(value) as probability_values.cc:27-36 computes it:
// value * kScale + (kMin - kScale), 0 -> kMin -- while lane 0 of wave 0 replays the reference's
// `score += probability` loop (rtcsm_3d.cc:101-104) over the previous tile: strictly sequential
// float additions in point order, so the sum is bit-identical to the reference's.
constexpr int kChainTile = 2048;
__global__ __launch_bounds__(256) void rtcsm_rescore_kernel(
    GridView g, const float* __restrict__ px, const float* __restrict__ py,
    const float* __restrict__ pz, int n, const float4* __restrict__ rot, int R,
    const float* __restrict__ trans, const unsigned* __restrict__ list, const unsigned* __restrict__ count,
    float k_scale, float k_offset, float k_unknown, float* __restrict__ sums) {
  __shared__ float4 tile[2][kChainTile / 4];
  if (count != nullptr && blockIdx.x >= *count) return;
  const unsigned c = list[blockIdx.x];
  const int j = static_cast<int>(c / static_cast<unsigned>(R));
  const int r = static_cast<int>(c % static_cast<unsigned>(R));
  const float4 qq = rot[r];
  const Quat4 q{qq.x, qq.y, qq.z, qq.w};
  const float tx = trans[3 * j], ty = trans[3 * j + 1], tz = trans[3 * j + 2];
  const int num_tiles = (n + kChainTile - 1) / kChainTile;
  const int producer = static_cast<int>(threadIdx.x) - 64;  // waves 1..3
  auto produce = [&](int t) {
    if (producer < 0) return;
    float* dst = reinterpret_cast<float*>(tile[t & 1]);
    for (int k = producer; k < kChainTile; k += 192) {
      const int i = t * kChainTile + k;
      float prob = 0.f;  // +0.f padding leaves a float running sum unchanged
      if (i < n) {
        float rx, ry, rz;
        rotate_point(q, px[i], py[i], pz[i], rx, ry, rz);
        const unsigned v = grid_value(g, cell_of(rx + tx, g.resolution), cell_of(ry + ty, g.resolution),
                                      cell_of(rz + tz, g.resolution)) & 0x7FFFu;
        prob = v == 0u ? k_unknown : static_cast<float>(static_cast<int>(v)) * k_scale + k_offset;
      }
      dst[k] = prob;
    }
  };
  float s = 0.f;
  produce(0);
  __syncthreads();
  for (int t = 0; t < num_tiles; ++t) {
    if (t + 1 < num_tiles) produce(t + 1);
    if (threadIdx.x == 0) {
      const float4* src = tile[t & 1];
#pragma unroll 8
      for (int k = 0; k < kChainTile / 4; ++k) {
        const float4 v = src[k];
        s += v.x;
        s += v.y;
        s += v.z;
        s += v.w;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = s;
}

// Parallel, bit-exact evaluation of the SAME sequential float sum (one workgroup of 1024 threads
// per surviving candidate).  Float addition is not associative, but inside one binade it is
// integer arithmetic: with the running sum s = m * U (U = ulp of the binade, m a 24-bit mantissa)
// and every addend an exact multiple of 2^-27,
//     fl(s + a) = (m + c) * U,   c = floor(a/U) + [frac(a/U) > 1/2]   (+ ties-to-even),
// so a stretch of additions that stays in one binade is an exact integer prefix sum.  Ties
// (frac == 1/2) depend on the parity of m, which a tie itself resets to even; that makes every
// segment of addends a function {parity in} -> {increment, parity out}, and functions compose
// associatively -> a block-wide scan.  The sum is therefore computed binade by binade: each pass
// scans a window that must contain the next binade crossing, finds the first addition whose
// result reaches 2^24 U, rounds that one exact sum to the new ulp 2U, and continues behind it.
// The first 256 additions are simply replayed in float by one lane.
struct ParityFn {
  unsigned s0, s1;  // total increment for parity-in 0 / 1
  unsigned p0, p1;  // parity out
};
__device__ __forceinline__ ParityFn compose(const ParityFn& a, const ParityFn& b) {  // a first, then b
  ParityFn r;
  r.s0 = a.s0 + (a.p0 ? b.s1 : b.s0);
  r.p0 = a.p0 ? b.p1 : b.p0;
  r.s1 = a.s1 + (a.p1 ? b.s1 : b.s0);
  r.p1 = a.p1 ? b.p1 : b.p0;
  return r;
}
__device__ __forceinline__ ParityFn shfl_up_fn(const ParityFn& f, int off) {
  ParityFn r;
  r.s0 = __shfl_up(f.s0, off, 64);
  r.s1 = __shfl_up(f.s1, off, 64);
  r.p0 = __shfl_up(f.p0, off, 64);
  r.p1 = __shfl_up(f.p1, off, 64);
  return r;
}

constexpr int kScanThreads = 1024;
constexpr int kSerialPrefix = 256;

// Per survivor k and point i (input order): the 15-bit grid value, for the scan kernel below.
__global__ void rtcsm_rescore_values_kernel(GridView g, const float* __restrict__ px,
                                            const float* __restrict__ py, const float* __restrict__ pz,
                                            int n, int n_stride, const float4* __restrict__ rot, int R,
                                            const float* __restrict__ trans, const unsigned* __restrict__ list,
                                            const unsigned* __restrict__ count,
                                            unsigned short* __restrict__ values, float k_scale, float k_offset,
                                            float k_unknown, double* __restrict__ chunk_sums, int num_chunks) {
  // launched for an upper bound of survivors when the host has not read the count yet
  if (count != nullptr && blockIdx.y >= *count) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned short v = 0;
  if (i < n) {
    const unsigned c = list[blockIdx.y];
    const int j = static_cast<int>(c / static_cast<unsigned>(R));
    const int r = static_cast<int>(c % static_cast<unsigned>(R));
    const float4 qq = rot[r];
    const Quat4 q{qq.x, qq.y, qq.z, qq.w};
    float rx, ry, rz;
    rotate_point(q, px[i], py[i], pz[i], rx, ry, rz);
    v = static_cast<unsigned short>(grid_value(g, cell_of(rx + trans[3 * j], g.resolution),
                                               cell_of(ry + trans[3 * j + 1], g.resolution),
                                               cell_of(rz + trans[3 * j + 2], g.resolution)) & 0x7FFFu);
  }
  if (i < n_stride) values[static_cast<size_t>(blockIdx.y) * n_stride + i] = v;
  if (chunk_sums != nullptr) {
    // real (double) sum of the probabilities of this wavefront's 64-point chunk: locates the binade
    // of the running sum for rtcsm_rescore_chunk_fns_kernel
    double p = 0.;
    if (i < n) p = v == 0 ? static_cast<double>(k_unknown)
                          : static_cast<double>(static_cast<float>(static_cast<int>(v)) * k_scale + k_offset);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
    const int chunk = i >> 6;
    if ((threadIdx.x & 63) == 0 && chunk < num_chunks) chunk_sums[static_cast<size_t>(blockIdx.y) * num_chunks + chunk] = p;
  }
}

__global__ __launch_bounds__(kScanThreads) void rtcsm_rescore_scan_kernel(
    const unsigned short* __restrict__ values, int n, int n_stride, float k_scale, float k_offset,
    float k_unknown, const unsigned* __restrict__ count, float* __restrict__ sums) {
  extern __shared__ unsigned short lds_value[];  // n_stride grid values (15 bit), input order
  if (count != nullptr && blockIdx.x >= *count) return;
  __shared__ ParityFn wave_total[kScanThreads / 64];
  __shared__ unsigned sh_m, sh_e, sh_i0, sh_cross, sh_m_before, sh_total;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
    // coalesced 16-byte loads of this survivor's row (n_stride is a multiple of 8)
    const uint4* src = reinterpret_cast<const uint4*>(values + static_cast<size_t>(blockIdx.x) * n_stride);
    uint4* dst = reinterpret_cast<uint4*>(lds_value);
    for (int i = tid; i < n_stride / 8; i += kScanThreads) dst[i] = src[i];
  }
  __syncthreads();
  // probability of point i as the reference's float (probability_values.cc:27-36) ...
  auto prob = [&](int i) -> float {
    const unsigned v = lds_value[i];
    return v == 0u ? k_unknown : static_cast<float>(static_cast<int>(v)) * k_scale + k_offset;
  };
  // ... and as an exact integer in units of 2^-27 (every probability is in [2^-4, 1))
  auto fixed = [&](int i) -> unsigned {
    const unsigned b = __float_as_uint(prob(i));
    return ((b & 0x7FFFFFu) | 0x800000u) << ((b >> 23) - 123u);
  };
  if (tid == 0) {
    float s = 0.f;
    const int n0 = min(n, kSerialPrefix);
    for (int i = 0; i < n0; ++i) s += prob(i);
    const unsigned b = __float_as_uint(s);
    sh_m = (b & 0x7FFFFFu) | 0x800000u;  // s = m * 2^(e - 27), e = biased exponent - 123
    sh_e = (b >> 23) - 123u;
    sh_i0 = static_cast<unsigned>(n0);
  }
  __syncthreads();
  while (sh_i0 < static_cast<unsigned>(n)) {  // uniform: one pass per binade
    const unsigned m = sh_m, e = sh_e, i0 = sh_i0;
    const unsigned U = 1u << e, half = U >> 1, fmask = U - 1u;
    // window that must contain the crossing: every addend is >= 0.1 > 13421772 * 2^-27
    const unsigned c_min = max(13421772u >> e, 1u);
    const unsigned remaining = static_cast<unsigned>(n) - i0;
    const unsigned window = min(remaining, ((1u << 24) - m) / c_min + 2u);
    const unsigned seg = ((window + kScanThreads - 1) / kScanThreads) | 1u;  // odd: LDS banks
    const unsigned begin = i0 + static_cast<unsigned>(tid) * seg;
    const unsigned end = min(i0 + window, begin + seg);
    // phase A: this segment as a function of the incoming parity
    ParityFn f{0u, 0u, 0u, 1u};
    for (unsigned i = begin; i < end; ++i) {
      const unsigned a = fixed(static_cast<int>(i));
      const unsigned q = a >> e, fr = a & fmask;
      if (e != 0u && fr == half) {  // tie: round to even mantissa
        f.s0 += q + ((f.p0 + q) & 1u);
        f.s1 += q + ((f.p1 + q) & 1u);
        f.p0 = 0u;
        f.p1 = 0u;
      } else {
        const unsigned c = q + (fr > half ? 1u : 0u);
        f.s0 += c;
        f.s1 += c;
        f.p0 = (f.p0 + c) & 1u;
        f.p1 = (f.p1 + c) & 1u;
      }
    }
    // block-wide inclusive scan of the composition
    ParityFn inc = f;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const ParityFn o = shfl_up_fn(inc, off);
      if (lane >= off) inc = compose(o, inc);
    }
    if (lane == 63) wave_total[wave] = inc;
    if (tid == 0) {
      sh_cross = 0xFFFFFFFFu;
    }
    __syncthreads();
    ParityFn before{0u, 0u, 0u, 1u};  // composition of all earlier waves
    for (int w = 0; w < wave; ++w) before = compose(before, wave_total[w]);
    ParityFn excl = shfl_up_fn(inc, 1);  // earlier lanes of this wave
    if (lane == 0) excl = ParityFn{0u, 0u, 0u, 1u};
    excl = compose(before, excl);
    const unsigned p_start = m & 1u;
    unsigned mt = m + (p_start ? excl.s1 : excl.s0);
    if (tid == kScanThreads - 1) {
      const ParityFn all = compose(before, inc);
      sh_total = p_start ? all.s1 : all.s0;
    }
    // phase B: replay the segment with the real mantissa, look for the first result >= 2^24
    unsigned my_cross = 0xFFFFFFFFu, my_before = 0u;
    for (unsigned i = begin; i < end; ++i) {
      const unsigned a = fixed(static_cast<int>(i));
      const unsigned q = a >> e, fr = a & fmask;
      unsigned c;
      if (e != 0u && fr == half) {
        c = q + ((mt + q) & 1u);
      } else {
        c = q + (fr > half ? 1u : 0u);
      }
      if (mt + c >= (1u << 24)) {
        my_cross = i;
        my_before = mt;
        break;
      }
      mt += c;
    }
    if (my_cross != 0xFFFFFFFFu) atomicMin(&sh_cross, my_cross);
    __syncthreads();
    if (my_cross != 0xFFFFFFFFu && my_cross == sh_cross) sh_m_before = my_before;
    __syncthreads();
    if (tid == 0) {
      if (sh_cross == 0xFFFFFFFFu) {  // the window ended inside this binade
        sh_m = m + sh_total;
        sh_i0 = i0 + window;
      } else {
        // exact sum of the crossing addition, rounded once to the next binade's ulp (2U)
        const unsigned long long X =
            (static_cast<unsigned long long>(sh_m_before) << e) + fixed(static_cast<int>(sh_cross));
        const unsigned e2 = e + 1u;
        const unsigned long long q2 = X >> e2, f2 = X & ((1ull << e2) - 1ull), h2 = 1ull << e;
        const unsigned long long up = (f2 > h2 || (f2 == h2 && (q2 & 1ull))) ? 1ull : 0ull;
        sh_m = static_cast<unsigned>(q2 + up);
        sh_e = e2;
        sh_i0 = sh_cross + 1u;
      }
    }
    __syncthreads();
  }
  if (tid == 0) sums[blockIdx.x] = __uint_as_float(((sh_e + 123u) << 23) | (sh_m & 0x7FFFFFu));
}

// ---- chunked variant of the exact scan (method 2) -----------------------------------------------
// The element scan above spends its time in one CU walking 64 elements per thread twice per
// binade.  Here the composition of every aligned 64-point chunk is computed ONCE, in parallel over
// the whole chip: the real (double) prefix sum tells, within a proven error bound, which binade(s)
// the reference's running float sum can be in while it crosses the chunk -- at most two -- and the
// chunk's ParityFn is stored for each.  The scan then works on chunk functions (one per thread) and
// only opens the two chunks that matter per binade: the one it resumes in and the one the sum
// leaves the binade in (both handled by a wavefront, lane per element).
constexpr int kChunk = 64;
struct ChunkFns {  // the chunk's ParityFn for up to two binades; e == 0xFFFFFFFF: absent
  unsigned e0, s00, s01, pp0;  // pp = p0 | p1 << 1
  unsigned e1, s10, s11, pp1;
};

__device__ __forceinline__ ParityFn element_fn(unsigned a, unsigned e) {
  const unsigned U = 1u << e, half = U >> 1, fmask = U - 1u;
  const unsigned q = a >> e, fr = a & fmask;
  if (e != 0u && fr == half) return ParityFn{q + (q & 1u), q + ((1u + q) & 1u), 0u, 0u};  // tie -> even
  const unsigned c = q + (fr > half ? 1u : 0u);
  return ParityFn{c, c, c & 1u, (1u + c) & 1u};
}
__device__ __forceinline__ ParityFn wave_inclusive_scan(ParityFn f, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const ParityFn o = shfl_up_fn(f, off);
    if (lane >= off) f = compose(o, f);
  }
  return f;
}
__device__ __forceinline__ unsigned fixed_of_value(unsigned v, float k_scale, float k_offset, float k_unknown) {
  const float p = v == 0u ? k_unknown : static_cast<float>(static_cast<int>(v)) * k_scale + k_offset;
  const unsigned b = __float_as_uint(p);
  return ((b & 0x7FFFFFu) | 0x800000u) << ((b >> 23) - 123u);
}
// binade index e (ulp = 2^(e-27)) of a positive real sum: floor(log2 x) + 4, never below 0
__device__ __forceinline__ int binade_of(double x) { return max(ilogb(fmax(x, 0.0625)) + 4, 0); }

__global__ __launch_bounds__(256) void rtcsm_rescore_chunk_fns_kernel(
    const unsigned short* __restrict__ values, int n, int n_stride, float k_scale, float k_offset, float k_unknown,
    const unsigned* __restrict__ count, const double* __restrict__ chunk_sums, int num_chunks,
    ChunkFns* __restrict__ fns) {
  if (count != nullptr && blockIdx.y >= *count) return;
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= num_chunks) return;
  const double* sums = chunk_sums + static_cast<size_t>(blockIdx.y) * num_chunks;
  double before = 0.;
  for (int k = lane; k < c; k += 64) before += sums[k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
  const double after = before + sums[c];
  const int i = c * kChunk + lane;
  const int i_end = min(n, (c + 1) * kChunk);
  // |sequential float sum - real sum| <= sum_i 2^-24 s_i <= i 2^-24 s_i (partial sums are monotone)
  const double delta = 1.05 * static_cast<double>(i_end) * 5.9604644775390625e-8 * after + 1e-6;
  const int e_lo = binade_of(before - delta), e_hi = binade_of(after + delta);
  ChunkFns out{0xFFFFFFFFu, 0u, 0u, 0u, 0xFFFFFFFFu, 0u, 0u, 0u};
  if (e_hi - e_lo <= 1) {
    const unsigned a = i < n ? fixed_of_value(values[static_cast<size_t>(blockIdx.y) * n_stride + i], k_scale, k_offset, k_unknown) : 0u;
    const ParityFn id{0u, 0u, 0u, 1u};
    const ParityFn t0 = wave_inclusive_scan(i < n ? element_fn(a, static_cast<unsigned>(e_lo)) : id, lane);
    out.e0 = static_cast<unsigned>(e_lo);
    out.s00 = t0.s0;
    out.s01 = t0.s1;
    out.pp0 = t0.p0 | (t0.p1 << 1);
    if (e_hi != e_lo) {
      const ParityFn t1 = wave_inclusive_scan(i < n ? element_fn(a, static_cast<unsigned>(e_hi)) : id, lane);
      out.e1 = static_cast<unsigned>(e_hi);
      out.s10 = t1.s0;
      out.s11 = t1.s1;
      out.pp1 = t1.p0 | (t1.p1 << 1);
    }
  }
  if (lane == 63) fns[static_cast<size_t>(blockIdx.y) * num_chunks + c] = out;
}

// ParityFn in two words (increment in bits 0..30, parity out in bit 31): a composition is and / select / add per word,
// and the scans below move it with DPP (row shifts and the two row broadcasts) instead of ds_bpermute -- the scan
// kernel is ONE workgroup per survivor and nothing but dependent latency: ~45 wave scans and ~60 barriers a survivor
// made it 53 us for the one survivor a match usually has (round 4 profile).  Increments inside a pass's window stay
// below 2^28 (at most 2^24 / c_min elements of at most 10 c_min each), so bit 31 is free.
struct PFn2 {
  unsigned a, b;  // a: parity in 0, b: parity in 1
};
__device__ __forceinline__ PFn2 pfn2_identity() { return PFn2{0u, 0x80000000u}; }
__device__ __forceinline__ PFn2 pack_fn(const ParityFn& f) { return PFn2{f.s0 | (f.p0 << 31), f.s1 | (f.p1 << 31)}; }
__device__ __forceinline__ PFn2 compose2(const PFn2& x, const PFn2& y) {  // x first, then y
  PFn2 r;
  r.a = (x.a & 0x7FFFFFFFu) + (static_cast<int>(x.a) < 0 ? y.b : y.a);
  r.b = (x.b & 0x7FFFFFFFu) + (static_cast<int>(x.b) < 0 ? y.b : y.a);
  return r;
}
template <int kCtrl, int kRowMask>
__device__ __forceinline__ PFn2 dpp_fn(const PFn2& f) {  // lanes without a source (or outside the row mask) get the identity
  PFn2 r;
  r.a = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(f.a), kCtrl, kRowMask, 0xf, false));
  r.b = static_cast<unsigned>(__builtin_amdgcn_update_dpp(static_cast<int>(0x80000000u), static_cast<int>(f.b), kCtrl, kRowMask, 0xf, false));
  return r;
}
__device__ __forceinline__ PFn2 wave_inclusive_scan2(PFn2 f) {
  f = compose2(dpp_fn<0x111, 0xf>(f), f);  // row_shr:1
  f = compose2(dpp_fn<0x112, 0xf>(f), f);  // row_shr:2
  f = compose2(dpp_fn<0x114, 0xf>(f), f);  // row_shr:4
  f = compose2(dpp_fn<0x118, 0xf>(f), f);  // row_shr:8
  f = compose2(dpp_fn<0x142, 0xa>(f), f);  // row_bcast:15 into rows 1 and 3
  f = compose2(dpp_fn<0x143, 0xc>(f), f);  // row_bcast:31 into rows 2 and 3
  return f;
}
__device__ __forceinline__ PFn2 wave_shift_right1(const PFn2& f) { return dpp_fn<0x138, 0xf>(f); }  // wave_shr:1, lane 0: identity
__device__ __forceinline__ PFn2 lane_of(const PFn2& f, int l) {  // l uniform
  return PFn2{static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(f.a), l)),
              static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(f.b), l))};
}
__device__ __forceinline__ unsigned apply2(const PFn2& f, unsigned parity) { return (parity ? f.b : f.a) & 0x7FFFFFFFu; }

// Round 5 form: thread t <-> chunk t for the whole kernel (its ChunkFns are loaded ONCE, beside the copy of the values
// into LDS, instead of a dependent 32-byte global load per pass), the first 256 additions replayed by wave 0 out of
// registers with v_readlane (one thread chasing 256 dependent LDS reads was a fifth of the kernel), the partial chunk a
// pass resumes in and the chunk the sum leaves the binade in opened by the WAVE that owns them (no hand-over through
// LDS), three barriers a pass.  Same arithmetic, same result bits as before (test_sequential_sum_kernels_bit_exact
// compares it with the element scan and the serial replay).
__global__ __launch_bounds__(kScanThreads) void rtcsm_rescore_chunk_scan_kernel(
    const unsigned short* __restrict__ values, int n, int n_stride, float k_scale, float k_offset, float k_unknown,
    const unsigned* __restrict__ count, const ChunkFns* __restrict__ fns, int num_chunks, float* __restrict__ sums) {
  extern __shared__ unsigned short lds_value[];  // n_stride grid values (15 bit), input order
  if (count != nullptr && blockIdx.x >= *count) return;
  __shared__ PFn2 wave_total[kScanThreads / 64];
  __shared__ unsigned sh_m, sh_e, sh_i0, sh_cross_t, sh_total, sh_mismatch_t, sh_force_serial;
  constexpr int kEarlyChunks = 128;            // the functions of the first chunks, for wave 0's own passes (below)
  __shared__ ChunkFns early_fns[kEarlyChunks];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const ChunkFns none{0xFFFFFFFFu, 0u, 0u, 0u, 0xFFFFFFFFu, 0u, 0u, 0u};
  const ChunkFns cf = tid < num_chunks ? fns[static_cast<size_t>(blockIdx.x) * num_chunks + tid] : none;  // in flight during the copy
  if (tid < kEarlyChunks) early_fns[tid] = cf;
  {
    const uint4* src = reinterpret_cast<const uint4*>(values + static_cast<size_t>(blockIdx.x) * n_stride);
    uint4* dst = reinterpret_cast<uint4*>(lds_value);
    for (int i = tid; i < n_stride / 8; i += kScanThreads) dst[i] = src[i];
  }
  __syncthreads();
  auto prob = [&](unsigned v) { return v == 0u ? k_unknown : static_cast<float>(static_cast<int>(v)) * k_scale + k_offset; };
  auto fixed = [&](unsigned i) { return fixed_of_value(lds_value[i], k_scale, k_offset, k_unknown); };
  const int n0 = min(n, kSerialPrefix);
  if (wave == 0) {
    float s = 0.f;
    if (n0 == kSerialPrefix) {
      // element k * 64 + l sits in lane l's register k: the 256 sequential additions read them with v_readlane
      float pv[kSerialPrefix / 64];
#pragma unroll
      for (int k = 0; k < kSerialPrefix / 64; ++k) pv[k] = prob(lds_value[k * 64 + lane]);
#pragma unroll
      for (int k = 0; k < kSerialPrefix / 64; ++k)
#pragma unroll
        for (int l = 0; l < 64; ++l) s += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pv[k]), l));
    } else {
      for (int i = 0; i < n0; ++i) s += prob(lds_value[i]);
    }
    // ---- the first binades by wave 0 ALONE (round 5).  The sum doubles from binade to binade, so the early passes
    // cover a few dozen chunks each -- and used to cost what the late ones cost, three barriers of sixteen waves (4.4 us
    // a pass, nine passes).  While a pass's window fits 64 chunks (lane l <-> chunk c0 + l, functions out of LDS) it
    // needs no other wave: the same steps as the block's pass below on wave-uniform state, no barrier.  Anything
    // unusual (a chunk without a function for this binade in front of the crossing, a window beyond the early chunks)
    // leaves the pass to the block.
    const unsigned b0 = __float_as_uint(s);
    unsigned m = (b0 & 0x7FFFFFu) | 0x800000u, e = (b0 >> 23) - 123u, i0 = static_cast<unsigned>(n0);
    const PFn2 idw = pfn2_identity();
    while (i0 < static_cast<unsigned>(n)) {
      const unsigned c_min = max(13421772u >> e, 1u);
      const unsigned window = min(static_cast<unsigned>(n) - i0, ((1u << 24) - m) / c_min + 2u);
      const unsigned c0 = i0 / kChunk;
      const unsigned i_lim = min(static_cast<unsigned>(n), ((i0 + window + kChunk - 1u) / kChunk) * kChunk);
      const unsigned last_chunk = (i_lim - 1u) / kChunk;
      if (last_chunk - c0 >= 64u || last_chunk >= static_cast<unsigned>(kEarlyChunks)) break;  // the block's job
      const unsigned c = c0 + static_cast<unsigned>(lane);
      const unsigned begin = max(i0, c * kChunk), end = min(i_lim, (c + 1u) * kChunk);
      const bool in_window = begin < end;
      PFn2 f = idw;
      bool mismatch = false;
      if (in_window && lane != 0) {
        const ChunkFns cw = early_fns[c];
        if (cw.e0 == e) f = PFn2{cw.s00 | ((cw.pp0 & 1u) << 31), cw.s01 | ((cw.pp0 >> 1) << 31)};
        else if (cw.e1 == e) f = PFn2{cw.s10 | ((cw.pp1 & 1u) << 31), cw.s11 | ((cw.pp1 >> 1) << 31)};
        else mismatch = true;
      }
      {  // the chunk the walk resumes in is partial: lane per element
        const unsigned he = min(i_lim, (c0 + 1u) * kChunk);
        const unsigned i = i0 + static_cast<unsigned>(lane);
        const PFn2 h = wave_inclusive_scan2(i < he ? pack_fn(element_fn(fixed(i), e)) : idw);
        const PFn2 whole = lane_of(h, 63);
        if (lane == 0) f = whole;
      }
      const PFn2 inc = wave_inclusive_scan2(f);
      const PFn2 excl = wave_shift_right1(inc);
      const unsigned p_start = m & 1u;
      const unsigned mt_start = m + apply2(excl, p_start), mt_end = m + apply2(inc, p_start);
      const unsigned long long cross_mask = __ballot(in_window && mt_end >= (1u << 24) && mt_start < (1u << 24));
      const unsigned long long mism_mask = __ballot(mismatch);
      const int cross_l = cross_mask != 0ull ? __ffsll(static_cast<long long>(cross_mask)) - 1 : 64;
      const int mism_l = mism_mask != 0ull ? __ffsll(static_cast<long long>(mism_mask)) - 1 : 64;
      if (mism_l < cross_l || (cross_l == 64 && mism_l != 64)) break;  // the block's pass knows how to open such chunks
      if (cross_l == 64) {  // the cloud ended inside this binade
        m = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(mt_end), 63));
        i0 = i_lim;
        continue;
      }
      const unsigned cb = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(begin), cross_l));
      const unsigned ce = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(end), cross_l));
      const unsigned ms = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(mt_start), cross_l));
      const unsigned i = cb + static_cast<unsigned>(lane);
      const unsigned a_i = i < ce ? fixed(i) : 0u;
      const PFn2 sc = wave_inclusive_scan2(i < ce ? pack_fn(element_fn(a_i, e)) : idw);
      const PFn2 ex = wave_shift_right1(sc);
      const unsigned ps = ms & 1u;
      const unsigned m_before = ms + apply2(ex, ps), m_after = ms + apply2(sc, ps);
      const unsigned long long crossed = __ballot(i < ce && m_after >= (1u << 24));
      if (crossed == 0ull) break;  // (cannot happen: the chunk's function said it crosses) -- the block decides
      const int first = __ffsll(static_cast<long long>(crossed)) - 1;
      const unsigned mb = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(m_before), first));
      const unsigned af = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(a_i), first));
      // exact sum of the crossing addition, rounded once to the next binade's ulp (2U)
      const unsigned long long X = (static_cast<unsigned long long>(mb) << e) + af;
      const unsigned e2 = e + 1u;
      const unsigned long long q2 = X >> e2, f2 = X & ((1ull << e2) - 1ull), h2 = 1ull << e;
      const unsigned long long up = (f2 > h2 || (f2 == h2 && (q2 & 1ull))) ? 1ull : 0ull;
      m = static_cast<unsigned>(q2 + up);
      e = e2;
      i0 = cb + static_cast<unsigned>(first) + 1u;
    }
    if (lane == 0) {
      sh_m = m;
      sh_e = e;
      sh_i0 = i0;
      sh_force_serial = 0u;
    }
  }
  __syncthreads();
  const PFn2 id = pfn2_identity();
  while (sh_i0 < static_cast<unsigned>(n)) {  // uniform: one pass per binade
    const unsigned m = sh_m, e = sh_e, i0 = sh_i0, force_serial = sh_force_serial;
    // window that must contain the crossing (every addend >= 0.1 > 13421772 * 2^-27), rounded up to
    // a chunk boundary: elements behind the crossing are never looked at
    const unsigned c_min = max(13421772u >> e, 1u);
    const unsigned remaining = static_cast<unsigned>(n) - i0;
    const unsigned window = min(remaining, ((1u << 24) - m) / c_min + 2u);
    const unsigned c0 = i0 / kChunk;
    const unsigned i_lim = min(static_cast<unsigned>(n), ((i0 + window + kChunk - 1u) / kChunk) * kChunk);
    // thread t <-> chunk t, clipped to [i0, i_lim): chunks before the one the walk resumes in and behind the window
    // are the identity
    const unsigned c = static_cast<unsigned>(tid);
    const unsigned begin = max(i0, c * kChunk), end = min(i_lim, (c + 1u) * kChunk);
    const bool in_window = begin < end && c >= c0;
    PFn2 f = id;
    bool mismatch = false;
    if (in_window && c != c0) {
      if (cf.e0 == e) f = PFn2{cf.s00 | ((cf.pp0 & 1u) << 31), cf.s01 | ((cf.pp0 >> 1) << 31)};
      else if (cf.e1 == e) f = PFn2{cf.s10 | ((cf.pp1 & 1u) << 31), cf.s11 | ((cf.pp1 >> 1) << 31)};
      else mismatch = true;
    }
    // A chunk without a function for this binade lies BEHIND the crossing (the window bound is loose
    // by up to 9x, and the real prefix proves the sum has left the binade by then): it stays the
    // identity and is never selected.  Should one ever sit before the crossing -- a violated bound --
    // the pass is repeated with such chunks opened element by element (sh_force_serial).
    if (mismatch && force_serial) {
      ParityFn g = ParityFn{0u, 0u, 0u, 1u};
      for (unsigned i = begin; i < end; ++i) g = compose(g, element_fn(fixed(i), e));
      f = pack_fn(g);
      mismatch = false;
    }
    if (static_cast<unsigned>(wave) == (c0 >> 6)) {  // the chunk the walk resumes in is partial: lane per element, by its wave
      const unsigned he = min(i_lim, (c0 + 1u) * kChunk);
      const unsigned i = i0 + static_cast<unsigned>(lane);
      const PFn2 h = wave_inclusive_scan2(i < he ? pack_fn(element_fn(fixed(i), e)) : id);
      const PFn2 whole = lane_of(h, 63);
      if (static_cast<unsigned>(lane) == (c0 & 63u)) f = whole;
    }
    // block-wide inclusive scan over the chunk functions
    const PFn2 inc = wave_inclusive_scan2(f);
    if (lane == 63) wave_total[wave] = inc;
    if (tid == 0) {
      sh_cross_t = 0xFFFFFFFFu;
      sh_mismatch_t = 0xFFFFFFFFu;
    }
    __syncthreads();  // (1) wave totals, reset words
    if (mismatch) atomicMin(&sh_mismatch_t, static_cast<unsigned>(tid));
    PFn2 before = id;
    for (int w = 0; w < wave; ++w) before = compose2(before, wave_total[w]);
    const PFn2 excl = compose2(before, wave_shift_right1(inc));
    const PFn2 incl = compose2(before, inc);
    const unsigned p_start = m & 1u;
    const unsigned mt_start = m + apply2(excl, p_start);
    const unsigned mt_end = m + apply2(incl, p_start);
    if (tid == kScanThreads - 1) sh_total = mt_end - m;
    const bool crosses = in_window && mt_end >= (1u << 24) && mt_start < (1u << 24);
    if (crosses) atomicMin(&sh_cross_t, static_cast<unsigned>(tid));
    __syncthreads();  // (2) first crossing chunk, first chunk without a function, total
    const unsigned cross_t = sh_cross_t, mismatch_t = sh_mismatch_t;
    if (mismatch_t < cross_t || (cross_t == 0xFFFFFFFFu && mismatch_t != 0xFFFFFFFFu)) {
      __syncthreads();  // everyone has read the verdict
      if (tid == 0) sh_force_serial = 1u;
      __syncthreads();
      continue;  // same (m, e, i0), chunks without a function opened serially
    }
    if (cross_t == 0xFFFFFFFFu) {
      if (tid == 0) {  // the cloud ended inside this binade
        sh_m = m + sh_total;
        sh_i0 = i_lim;
      }
    } else if (static_cast<unsigned>(wave) == (cross_t >> 6)) {  // open the crossing chunk: lane per element, by its wave
      const int cl = static_cast<int>(cross_t & 63u);
      const unsigned cb = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(begin), cl));
      const unsigned ce = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(end), cl));
      const unsigned ms = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(mt_start), cl));
      const unsigned i = cb + static_cast<unsigned>(lane);
      const PFn2 sc = wave_inclusive_scan2(i < ce ? pack_fn(element_fn(fixed(i), e)) : id);
      const PFn2 ex = wave_shift_right1(sc);
      const unsigned ps = ms & 1u;
      const unsigned m_before = ms + apply2(ex, ps);
      const unsigned m_after = ms + apply2(sc, ps);
      const unsigned long long crossed = __ballot(i < ce && m_after >= (1u << 24));
      const int first = __ffsll(static_cast<long long>(crossed)) - 1;
      if (lane == first) {
        // exact sum of the crossing addition, rounded once to the next binade's ulp (2U)
        const unsigned long long X = (static_cast<unsigned long long>(m_before) << e) + fixed(i);
        const unsigned e2 = e + 1u;
        const unsigned long long q2 = X >> e2, f2 = X & ((1ull << e2) - 1ull), h2 = 1ull << e;
        const unsigned long long up = (f2 > h2 || (f2 == h2 && (q2 & 1ull))) ? 1ull : 0ull;
        sh_m = static_cast<unsigned>(q2 + up);
        sh_e = e2;
        sh_i0 = i + 1u;
      }
    }
    __syncthreads();  // (3) the next pass's state
  }
  if (tid == 0) sums[blockIdx.x] = __uint_as_float(((sh_e + 123u) << 23) | (sh_m & 0x7FFFFFu));
}

// ---------------------------------------------------------------------------------- probes
__global__ void probe_cells_kernel(Quat4 q, float tx, float ty, float tz, const float* __restrict__ px,
                                   const float* __restrict__ py, const float* __restrict__ pz, int n,
                                   float resolution, int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float rx, ry, rz;
  rotate_point(q, px[i], py[i], pz[i], rx, ry, rz);
  out[3 * i] = cell_of(rx + tx, resolution);
  out[3 * i + 1] = cell_of(ry + ty, resolution);
  out[3 * i + 2] = cell_of(rz + tz, resolution);
}

// ---------------------------------------------------------------------------------- host side
struct Candidates {
  dliom_rtcsm_window w;
  PoseF init;
  std::vector<QF> rot;        // candidate rotation  normalized(init.q * q_r)
  std::vector<QF> rot_raw;    // q_r (transform rotation)
  std::vector<float> r_angle; // GetAngle(transform) per rotation
  std::vector<F3> trans;      // candidate translation init.q * t_j + init.t
  std::vector<float> t_norm;  // ||t_j||
};

// rtcsm_3d.cc:58-70
static void compute_window(const dliom_rtcsm_options& o, float resolution, float cloud_max_norm_value,
                           dliom_rtcsm_window* w) {
  w->linear_window_size = static_cast<int>(std::lround(o.linear_search_window / resolution));
  float max_scan_range = 3.f * resolution;
  max_scan_range = std::max(cloud_max_norm_value, max_scan_range);
  const float kSafetyMargin = 1.f - 1e-3f;
  const float res2 = resolution * (resolution * 1.f);
  const float range2 = max_scan_range * (max_scan_range * 1.f);
  w->angular_step_size = kSafetyMargin * std::acos(1.f - res2 / (2.f * range2));
  w->angular_window_size =
      static_cast<int>(std::lround(o.angular_search_window / w->angular_step_size));
  w->max_scan_range = max_scan_range;
  const int64_t tl = 2 * static_cast<int64_t>(w->linear_window_size) + 1;
  const int64_t ta = 2 * static_cast<int64_t>(w->angular_window_size) + 1;
  w->num_translations = tl * tl * tl;
  w->num_rotations = ta * ta * ta;
  w->num_candidates = w->num_translations * w->num_rotations;
}

static void generate_candidates(const dliom_rtcsm_options& o, float resolution, float max_norm,
                                const double init7[7], Candidates* c) {
  compute_window(o, resolution, max_norm, &c->w);
  c->init.t = F3{static_cast<float>(init7[0]), static_cast<float>(init7[1]), static_cast<float>(init7[2])};
  c->init.q = QF{static_cast<float>(init7[3]), static_cast<float>(init7[4]),
                 static_cast<float>(init7[5]), static_cast<float>(init7[6])};
  const int L = c->w.linear_window_size, A = c->w.angular_window_size;
  const float step = c->w.angular_step_size;
  // The transform rotations q_r and their angles depend on the window only: cached per thread
  // across matches (sin/cos/atan2 per rotation are the expensive part of this function).
  struct RotCache {
    int A = -1;
    float step = 0.f;
    std::vector<QF> rot_raw;
    std::vector<float> r_angle;
  };
  static thread_local RotCache cache;
  if (cache.A != A || cache.step != step) {
    cache.A = A;
    cache.step = step;
    cache.rot_raw.clear();
    cache.r_angle.clear();
    cache.rot_raw.reserve(c->w.num_rotations);
    cache.r_angle.reserve(c->w.num_rotations);
    for (int rz = -A; rz <= A; ++rz)
      for (int ry = -A; ry <= A; ++ry)
        for (int rx = -A; rx <= A; ++rx) {
          const QF q = angle_axis_to_quaternion(F3{rx * step, ry * step, rz * step});
          cache.rot_raw.push_back(q);
          cache.r_angle.push_back(rotation_angle(q));
        }
  }
  c->rot_raw = cache.rot_raw;
  c->r_angle = cache.r_angle;
  c->rot.resize(cache.rot_raw.size());
  for (size_t i = 0; i < cache.rot_raw.size(); ++i) c->rot[i] = qnormalized(qmul(c->init.q, cache.rot_raw[i]));
  c->trans.clear();
  c->t_norm.clear();
  for (int z = -L; z <= L; ++z)
    for (int y = -L; y <= L; ++y)
      for (int x = -L; x <= L; ++x) {
        const F3 t{x * resolution, y * resolution, z * resolution};
        c->t_norm.push_back(norm3(t));
        c->trans.push_back(add3(qrot(c->init.q, t), c->init.t));
      }
}

// Device copies of the candidate tables inside ctx->cand.
struct DeviceCandidates {
  float4* rot;
  float* trans;    // T x 3
  float4* trans4;  // T x (x,y,z,0): 16-byte rows for the rotation-per-lane kernel
  float* t_norm;
  float* r_angle;
  const float4* rot_src;  // where a kernel that runs BEFORE the pending copy (PrepArgs) finds the rotations: the pinned
                          // staging copy while the upload is pending, `rot` itself otherwise
};

// ---- pending copies / fills of a match (device_common.h: PrepArgs) --------------------------------------------------
static bool prep_add(PrepArgs* prep, void* dst, const void* src, size_t bytes, unsigned value) {
  if (prep == nullptr || prep->n >= kMaxPrepJobs || (bytes & 3u) != 0 || (reinterpret_cast<uintptr_t>(dst) & 15u) != 0 ||
      (reinterpret_cast<uintptr_t>(src) & 15u) != 0 || bytes / 16 > 0xFFFFFFFFull)
    return false;
  PrepJob& j = prep->job[prep->n++];
  j.dst = static_cast<uint4*>(dst);
  j.src = static_cast<const uint4*>(src);
  j.vec = static_cast<unsigned>(bytes / 16);
  j.tail_words = static_cast<unsigned>((bytes % 16) / 4);
  j.value = value;
  return true;
}
__global__ __launch_bounds__(256) void rtcsm_prep_kernel(PrepArgs a) { prep_block(a, blockIdx.x, gridDim.x, threadIdx.x, blockDim.x); }
// The pending jobs as a launch of their own (score kernels other than the LDS-box one, whose pre-pass carries them).
static int prep_flush(dliom_ctx* ctx, PrepArgs* prep) {
  if (prep == nullptr || prep->n == 0) return DLIOM_OK;
  unsigned most = 1;
  for (int j = 0; j < prep->n; ++j) most = std::max(most, prep->job[j].vec);
  hipLaunchKernelGGL(rtcsm_prep_kernel, dim3(std::min(512u, (most + 255u) / 256u)), dim3(256), 0, ctx->stream, *prep);
  DLIOM_HIP_TRY(hipGetLastError());
  prep->n = 0;
  return DLIOM_OK;
}

// prep != null: the H2D copy becomes a pending job (the staged tables stay in the pinned block until the kernel that
// carries the jobs has run -- every entry point that gets here synchronises the stream before it returns).
static int upload_candidates(dliom_ctx* ctx, const Candidates& c, DeviceCandidates* d, PrepArgs* prep = nullptr) {
  const size_t R = c.rot.size(), T = c.trans.size();
  const size_t Tpad = T;
  const size_t bytes_rot = (R * 16 + 255) & ~static_cast<size_t>(255);
  const size_t bytes_trans = (Tpad * 12 + 255) & ~static_cast<size_t>(255);
  const size_t bytes_tn = (T * 4 + 255) & ~static_cast<size_t>(255);
  const size_t bytes_ra = (R * 4 + 255) & ~static_cast<size_t>(255);
  const size_t bytes_t4 = (T * 16 + 255) & ~static_cast<size_t>(255);
  const size_t total = bytes_rot + bytes_trans + bytes_tn + bytes_ra + bytes_t4;
  DLIOM_TRY(ctx->cand.reserve(total));
  std::vector<char> host(total, 0);
  float* hr = reinterpret_cast<float*>(host.data());
  for (size_t i = 0; i < R; ++i) {
    hr[4 * i] = c.rot[i].w;
    hr[4 * i + 1] = c.rot[i].x;
    hr[4 * i + 2] = c.rot[i].y;
    hr[4 * i + 3] = c.rot[i].z;
  }
  float* ht = reinterpret_cast<float*>(host.data() + bytes_rot);
  for (size_t i = 0; i < Tpad; ++i) {
    const F3& t = c.trans[std::min(i, T - 1)];  // padding repeats the last translation (unused)
    ht[3 * i] = t.x;
    ht[3 * i + 1] = t.y;
    ht[3 * i + 2] = t.z;
  }
  std::memcpy(host.data() + bytes_rot + bytes_trans, c.t_norm.data(), T * 4);
  std::memcpy(host.data() + bytes_rot + bytes_trans + bytes_tn, c.r_angle.data(), R * 4);
  float* h4 = reinterpret_cast<float*>(host.data() + bytes_rot + bytes_trans + bytes_tn + bytes_ra);
  for (size_t i = 0; i < T; ++i) {
    h4[4 * i] = c.trans[i].x;
    h4[4 * i + 1] = c.trans[i].y;
    h4[4 * i + 2] = c.trans[i].z;
    h4[4 * i + 3] = 0.f;
  }
  char* base = static_cast<char*>(ctx->cand.p);
  d->rot_src = reinterpret_cast<const float4*>(base);
  if (total + 81920 <= ctx->pinned_bytes) {  // the last 80 KB stage the box-kernel tables and the readbacks
    // Pinned staging: truly asynchronous.  Every entry point that gets here synchronises the
    // stream before it returns, so the block is free again at the next call.
    std::memcpy(ctx->pinned, host.data(), total);
    if (prep_add(prep, base, ctx->pinned, total, 0u))
      d->rot_src = static_cast<const float4*>(ctx->pinned);
    else
      DLIOM_HIP_TRY(hipMemcpyAsync(base, ctx->pinned, total, hipMemcpyHostToDevice, ctx->stream));
  } else {
    DLIOM_HIP_TRY(hipMemcpyAsync(base, host.data(), total, hipMemcpyHostToDevice, ctx->stream));
    DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));  // `host` dies at scope exit
  }
  d->rot = reinterpret_cast<float4*>(base);
  d->trans = reinterpret_cast<float*>(base + bytes_rot);
  d->t_norm = reinterpret_cast<float*>(base + bytes_rot + bytes_trans);
  d->r_angle = reinterpret_cast<float*>(base + bytes_rot + bytes_trans + bytes_tn);
  d->trans4 = reinterpret_cast<float4*>(base + bytes_rot + bytes_trans + bytes_tn + bytes_ra);
  return DLIOM_OK;
}

static int env_int(const char* name, int fallback) { return tuning_int(name, fallback); }  // experiments builds only

// LDS-box score kernel (score_box.h): builds the per-pass constants in double and launches it.
// Returns DLIOM_ERR_CAPACITY when the search does not suit the kernel (the caller then uses the dense
// kernel): boxes that cannot hold a typical point's lookups, or coordinates beyond the scaled range.
static int launch_score_box(dliom_ctx* ctx, const dliom_cloud& cloud, const GridView& g, const Candidates& c,
                            const DeviceCandidates& d, int r_first, int r_last, unsigned long long* d_sums,
                            unsigned* d_error, PrepArgs* prep) {
  using namespace box;
  const int R = static_cast<int>(c.w.num_rotations), T = static_cast<int>(c.w.num_translations);
  const int n = static_cast<int>(cloud.n);
  // Which instantiation (score_box.h; round 6, A/B on BASELINE configs 2 and 5 in one process each):
  //   0  27 translations per pass, four waves per SIMD, 14 336-cell boxes: a window of ONE pass (config 2: 0.69 ms; the
  //      same kernel at three waves per SIMD 0.71, with 21 000-cell boxes 0.70, with 64-point chunks 0.70-0.72)
  //   1  the same at three waves per SIMD (168 registers) with 21 000-cell boxes: windows of several passes, whose
  //      translations reach further and make the boxes larger (config 5: 221 -> 195 ms)
  //   2  49 translations per pass at three waves per SIMD, 21 000-cell boxes: one staged box and one rotation per point
  //      serve 49 translations (a whole z-plane of config 5's 7^3 window: 159 ms; with 54 per pass 180) -- where its
  //      padding costs less than the 13 % of vector instructions per lookup it saves, by a margin (the wide kernel is
  //      measured at 343 translations only)
  static const int forced_variant = env_int("DLIOM_BOX_VARIANT", -1);
  const auto padded = [T](int tc) { return ((T + tc - 1) / tc) * tc; };
  const int variant = forced_variant >= 0 && forced_variant <= 2
                          ? forced_variant
                          : (T <= kTC ? 0 : (0.87 * padded(kTCWide) < 0.9 * padded(kTC) ? 2 : 1));
  const int TC = variant == 2 ? kTCWide : kTC;
  static const int forced_cells = env_int("DLIOM_BOX_CELLS", 0);
  const int cells = forced_cells > 0 ? forced_cells : (variant == 0 ? 14336 : 21000);
  // 64-point chunks where the boxes are large (config 5, same runs: 195 -> 189 ms at 27 translations per pass, 187 -> 181
  // at 54); on config 2's narrow kernel they cost 1-3 % (0.70-0.72 against 0.69 ms)
  static const int forced_chunk = env_int("DLIOM_BOX_CHUNK", 0);
  const int chunk_pts = forced_chunk > 0 ? forced_chunk : (variant == 0 ? kCostChunk : kCostChunkBig);
  static const int target_waves = env_int("DLIOM_BOX_WAVES", 0);
  const double res = static_cast<double>(g.resolution);
  const int passes = (T + TC - 1) / TC;
  // ---- error budget (score_box.h): E / u = 1 + (2 qmax + rmax + taumax) / 256
  double tmax = 0.0;
  for (const F3& t : c.trans) tmax = std::max(tmax, static_cast<double>(std::max(std::fabs(t.x), std::max(std::fabs(t.y), std::fabs(t.z)))));
  const double rmax = static_cast<double>(cloud.max_norm) / res + 1.0;
  const double qmax = rmax + tmax / res + 1.0;
  // Two different limits (round 5 separated them; until then the absolute coordinate was held below 880 cells, i.e. the
  // kernel only ran within 88 m of a 10 cm grid's origin): (i) Kb = (gi - lo) + f must be exact in a float with its 14
  // fraction bits, |gi - lo| < 1024 -- gi is the pass's translation and lo the box's origin, both ABSOLUTE cell
  // coordinates, and their difference is 128 minus the rotated point's cell: a limit on the scan's RANGE (checked below
  // with taumax, and per box in the kernel); (ii) the absolute coordinate q = (r + t) / res enters only the error budget
  // (the reference's own float rounding of r + t and of the quotient, 2 |q| 2^-24): the band widens with it, nothing
  // breaks -- any cell DynamicGrid can address (|q| < 8192 + range) is fine.
  if (!std::isfinite(qmax) || qmax > 16500.0) return (ctx->last_box_refusal = DLIOM_BOX_REFUSED_RANGE, DLIOM_ERR_CAPACITY);
  std::vector<Pass> pass(static_cast<size_t>(passes));
  std::vector<float> tau(static_cast<size_t>(passes) * TC * 4, 0.f);
  double taumax = 0.0;
  for (int tp = 0; tp < passes; ++tp) {  // pass centres first: taumax enters the band
    const int j0 = tp * TC, tc = std::min(TC, T - j0);
    for (int a = 0; a < 3; ++a) {
      double lo = 1e300, hi = -1e300;
      for (int j = j0; j < j0 + tc; ++j) {
        const double v = a == 0 ? c.trans[j].x : (a == 1 ? c.trans[j].y : c.trans[j].z);
        lo = std::min(lo, v);
        hi = std::max(hi, v);
      }
      taumax = std::max(taumax, (hi - lo) * 0.5 / res + 1.0);
    }
  }
  if (rmax + taumax + 130.0 > 1000.0) return (ctx->last_box_refusal = DLIOM_BOX_REFUSED_RANGE, DLIOM_ERR_CAPACITY);  // (i)
  const double e16 = 1.0 + (2.0 * qmax + rmax + taumax) / 256.0;
  const int B = static_cast<int>(std::ceil(e16 + 0.25));
  const int s_units = B;
  const double u = 1.0 / 65536.0;
  for (int tp = 0; tp < passes; ++tp) {
    Pass& ps = pass[static_cast<size_t>(tp)];
    ps.j0 = tp * TC;
    ps.tc = std::min(TC, T - ps.j0);
    ps.pad[0] = ps.pad[1] = 0;
    for (int a = 0; a < 3; ++a) {
      auto comp = [&](int j) { return static_cast<double>(a == 0 ? c.trans[j].x : (a == 1 ? c.trans[j].y : c.trans[j].z)); };
      double lo = 1e300, hi = -1e300;
      for (int j = ps.j0; j < ps.j0 + ps.tc; ++j) {
        lo = std::min(lo, comp(j));
        hi = std::max(hi, comp(j));
      }
      const double tcen = 0.5 * (lo + hi);
      const double G = tcen / res + 128.5 + s_units * u;
      const double gi = std::floor(G);
      const double f = std::nearbyint((G - gi) * 16384.0) / 16384.0;  // 14 fractional bits (may be 1.0)
      ps.gi[a] = static_cast<int>(gi);
      ps.f[a] = static_cast<float>(f);
      ps.uc[a] = static_cast<float>(tcen / res + 0.5);
      double reach = 0.0;
      for (int jj = 0; jj < TC; ++jj) {
        const int j = ps.j0 + std::min(jj, ps.tc - 1);  // short pass: padding repeats the last translation
        const double tv = comp(j) / res + 128.5 + s_units * u - (gi + f);
        const float tf = static_cast<float>(tv);
        tau[(static_cast<size_t>(tp) * TC + jj) * 4 + a] = tf;
        reach = std::max(reach, std::fabs(static_cast<double>(tf)));
      }
      // + the float roundings of the bounding box's own arithmetic at this distance from the origin (interval end +
      // pass centre: values up to qmax, four roundings of half an ulp each; 0.0002 of the kernel's slack covers < 1024)
      ps.reach[a] = static_cast<float>(reach + 0.02 + qmax * std::ldexp(1.0, -22));
    }
  }
  // ---- band bitmap: fractions phi of a scaled coordinate for which SOME translation of the pass gives
  // frac16(fl(phi + tau_j)) <= thr, i.e. frac(phi + tau_j) in [-u/2, (thr + 1/2) u]; one u of margin each side
  std::vector<unsigned> bitmap(static_cast<size_t>(passes) * kBitmapWords, 0u);
  const unsigned thr_units = static_cast<unsigned>(s_units + B);
  for (int tp = 0; tp < passes; ++tp) {
    const Pass& ps = pass[static_cast<size_t>(tp)];
    for (int a = 0; a < 3; ++a) {
      unsigned* words = bitmap.data() + static_cast<size_t>(tp) * kBitmapWords + static_cast<size_t>(a) * (kBuckets / 32);
      for (int jj = 0; jj < ps.tc; ++jj) {
        const double tv = static_cast<double>(tau[(static_cast<size_t>(tp) * TC + jj) * 4 + a]);
        const double af = tv - std::floor(tv);
        const double L = -af - 1.5 * u, H = -af + (thr_units + 1.5) * u;
        const long long b0 = static_cast<long long>(std::floor(L * kBuckets)), b1 = static_cast<long long>(std::floor(H * kBuckets));
        for (long long b = b0; b <= b1; ++b) {
          const unsigned m = static_cast<unsigned>(((b % kBuckets) + kBuckets) % kBuckets);
          words[m >> 5] |= 1u << (m & 31u);
        }
      }
    }
  }
  // ---- does a typical point fit?  spread of one point over 64 consecutive rotations <= 2 * 11 steps * rho
  {
    const double step = c.w.angular_step_size;
    const double a_lanes = (2.0 * c.w.angular_window_size + 1.0) * step;  // rotation-vector span of a workgroup (per axis, at most)
    const double spread = a_lanes * 0.6 * cloud.max_norm / res;                        // cells, at 60 % of the maximum range
    const double dim = spread + 2.0 * taumax + 8.0;
    if (dim > kMaxDim || dim * dim * (dim * 0.25) > cells) return (ctx->last_box_refusal = DLIOM_BOX_REFUSED_WINDOW, DLIOM_ERR_CAPACITY);
  }
  // ---- spread of every wave's 64 rotations around its centre lane (window only: q_init cancels)
  const int rot_groups_all = (r_last - r_first + 63) / 64;
  // waves per workgroup
  static const int forced_nw = env_int("DLIOM_BOX_NW", 0);  // tuning knob
  const int nw = forced_nw >= 1 && forced_nw <= kWaves
                     ? std::min(forced_nw, rot_groups_all)
                     : std::min(rot_groups_all, kWaves);  // four waves = one per SIMD: measured 2-3 % faster than three even
                                                          // when that leaves idle waves in the last rotation block
  const int rot_blocks = (rot_groups_all + nw - 1) / nw;
  if (TC * sizeof(float4) + kBitmapWords * 4 + 16 + static_cast<size_t>(nw) * kListWords * 4 + static_cast<size_t>(cells) * 2 >
      160 * 1024)
    return (ctx->last_box_refusal = DLIOM_BOX_REFUSED_LDS, DLIOM_ERR_CAPACITY);  // LDS budget (checked again where the launch is sized)
  struct GroupCache {  // depends on the window and the shard only: cached per thread across matches
    int A = -1, r_first = -1, r_last = -1, nw = 0;
    float step = 0.f;
    std::vector<Group> groups;
  };
  static thread_local GroupCache gcache;
  const bool gcache_hit = gcache.A == c.w.angular_window_size && gcache.step == c.w.angular_step_size &&
                          gcache.r_first == r_first && gcache.r_last == r_last && gcache.nw == nw;
  std::vector<Group>& groups = gcache.groups;
  if (!gcache_hit) {
    gcache.A = -1;
    groups.assign(static_cast<size_t>(rot_blocks), Group());
  }
  for (size_t gi = 0; !gcache_hit && gi < groups.size(); ++gi) {
    Group& gr = groups[gi];
    const int r0 = r_first + static_cast<int>(gi) * nw * 64;
    const int cnt = std::max(0, std::min(nw * 64, r_last - r0));
    gr.c_lane = cnt / 2;
    gr.theta2 = 0.f;
    for (int a = 0; a < 3; ++a) gr.dc[a] = gr.hd[a] = 0.f;
    if (cnt <= 0) {
      gr.c_lane = 0;
      continue;
    }
    auto qd = [&](int r, double q[4]) {
      const QF& f = c.rot_raw[static_cast<size_t>(r)];
      const double nn = std::sqrt(double(f.w) * f.w + double(f.x) * f.x + double(f.y) * f.y + double(f.z) * f.z);
      q[0] = f.w / nn; q[1] = f.x / nn; q[2] = f.y / nn; q[3] = f.z / nn;
    };
    double qc[4];
    qd(r0 + gr.c_lane, qc);
    double lo3[3] = {1e300, 1e300, 1e300}, hi3[3] = {-1e300, -1e300, -1e300}, th = 0.0;
    for (int l = 0; l < cnt; ++l) {
      double q[4];
      qd(r0 + l, q);
      // d = conj(qc) * q
      const double a0 = qc[0], a1 = -qc[1], a2 = -qc[2], a3 = -qc[3];
      double dq[4] = {a0 * q[0] - a1 * q[1] - a2 * q[2] - a3 * q[3], a0 * q[1] + a1 * q[0] + a2 * q[3] - a3 * q[2],
                      a0 * q[2] - a1 * q[3] + a2 * q[0] + a3 * q[1], a0 * q[3] + a1 * q[2] - a2 * q[1] + a3 * q[0]};
      if (dq[0] < 0) for (double& v : dq) v = -v;
      const double vn = std::sqrt(dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3]);
      const double ang = 2.0 * std::atan2(vn, dq[0]);
      const double k = vn > 1e-300 ? ang / vn : 2.0;
      const double dv[3] = {k * dq[1], k * dq[2], k * dq[3]};
      th = std::max(th, ang);
      for (int a = 0; a < 3; ++a) {
        lo3[a] = std::min(lo3[a], dv[a]);
        hi3[a] = std::max(hi3[a], dv[a]);
      }
    }
    if (th > 0.3) return (ctx->last_box_refusal = DLIOM_BOX_REFUSED_WINDOW, DLIOM_ERR_CAPACITY);  // the second-order bound below assumes small relative rotations
    for (int a = 0; a < 3; ++a) {
      gr.dc[a] = static_cast<float>(0.5 * (lo3[a] + hi3[a]));
      gr.hd[a] = static_cast<float>(0.5 * (hi3[a] - lo3[a]) * 1.0001 + 1e-7);
    }
    gr.theta2 = static_cast<float>(0.51 * th * th * 1.01 + 1e-9);
  }
  gcache.A = c.w.angular_window_size;
  gcache.step = c.w.angular_step_size;
  gcache.r_first = r_first;
  gcache.r_last = r_last;
  gcache.nw = nw;
  // every refusal (DLIOM_ERR_CAPACITY) is behind us: the launch below is what DLIOM_KERNEL_RTCSM_SCORE times
  struct SpanGuard {
    dliom_ctx* c;
    int span;
    ~SpanGuard() { c->end_span(span); }
  } span_guard{ctx, ctx->begin_span(DLIOM_KERNEL_RTCSM_SCORE)};
  // ---- device tables (after the candidate tables inside ctx->cand would alias uploads in flight: own buffer)
  const size_t tau_bytes = (tau.size() * 4 + 255) & ~static_cast<size_t>(255);
  const size_t pass_only_bytes = (pass.size() * sizeof(Pass) + 255) & ~static_cast<size_t>(255);
  const size_t group_bytes = (groups.size() * sizeof(Group) + 255) & ~static_cast<size_t>(255);
  const size_t bitmap_bytes = (bitmap.size() * 4 + 255) & ~static_cast<size_t>(255);
  const size_t pass_bytes = pass_only_bytes + group_bytes + bitmap_bytes;  // [passes | groups | bitmaps]
  DLIOM_TRY(ctx->box_tables.reserve(tau_bytes + pass_bytes));
  char* base = static_cast<char*>(ctx->box_tables.p);
  const Group* group_src = reinterpret_cast<const Group*>(base + tau_bytes + pass_only_bytes);  // for the pre-pass (below)
  // small (a few KB): staged through the pinned block when it fits, else a synchronous copy
  if (tau_bytes + pass_bytes <= 65536 && ctx->pinned_bytes >= 81920) {
    char* h = static_cast<char*>(ctx->pinned) + ctx->pinned_bytes - 81920;
    std::memcpy(h, tau.data(), tau.size() * 4);
    std::memcpy(h + tau_bytes, pass.data(), pass.size() * sizeof(Pass));
    std::memcpy(h + tau_bytes + pass_only_bytes, groups.data(), groups.size() * sizeof(Group));
    std::memcpy(h + tau_bytes + pass_only_bytes + group_bytes, bitmap.data(), bitmap.size() * 4);
    if (prep_add(prep, base, h, tau_bytes + pass_bytes, 0u))  // rides in the pre-pass, which reads its groups from the staging copy
      group_src = reinterpret_cast<const Group*>(h + tau_bytes + pass_only_bytes);
    else
      DLIOM_HIP_TRY(hipMemcpyAsync(base, h, tau_bytes + pass_bytes, hipMemcpyHostToDevice, ctx->stream));
  } else {
    DLIOM_HIP_TRY(hipMemcpyAsync(base, tau.data(), tau.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    DLIOM_HIP_TRY(hipMemcpyAsync(base + tau_bytes, pass.data(), pass.size() * sizeof(Pass), hipMemcpyHostToDevice, ctx->stream));
    DLIOM_HIP_TRY(hipMemcpyAsync(base + tau_bytes + pass_only_bytes, groups.data(), groups.size() * sizeof(Group),
                                 hipMemcpyHostToDevice, ctx->stream));
    DLIOM_HIP_TRY(hipMemcpyAsync(base + tau_bytes + pass_only_bytes + group_bytes, bitmap.data(), bitmap.size() * 4,
                                 hipMemcpyHostToDevice, ctx->stream));
    DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  Params p;
  p.tau = reinterpret_cast<const float4*>(base);
  p.pass = reinterpret_cast<const Pass*>(base + tau_bytes);
  p.group = reinterpret_cast<const Group*>(base + tau_bytes + pass_only_bytes);
  p.bitmap = reinterpret_cast<const unsigned*>(base + tau_bytes + pass_only_bytes + group_bytes);
  p.trans = d.trans;
  p.rot = d.rot;
  p.sums = d_sums;
  p.error = d_error;
  p.R = R;
  p.r_first = r_first;
  p.r_last = r_last;
  p.T = T;
  p.passes = passes;
  p.n = n;
  p.chunk = std::min(kMaxChunk, std::max(4, chunk_pts & ~3));
  p.order = p.chunk == kCostChunk ? cloud.d_chunk_order : (p.chunk == kCostChunkBig ? cloud.d_chunk_order_big : nullptr);
  p.point_chunks = (n + p.chunk - 1) / p.chunk;
  const int rot_groups = (r_last - r_first + 63) / 64;
  p.rot_groups = rot_groups;
  p.rot_blocks = rot_blocks;
  p.nw = nw;
  p.thr = thr_units;
  p.cells = cells;
  static const int split_points = env_int("DLIOM_BOX_SPLIT", 1);  // experiments builds: 0 = idle waves in a short rotation block
  p.split_points = split_points;
#ifdef DLIOM_EXPERIMENTS
  static const int box_debug = env_int("DLIOM_BOX_DEBUG", 0);
  p.debug = box_debug;
  if (box_debug & 512) p.order = nullptr;
#else
  p.debug = 0;
#endif
  const size_t lds = TC * sizeof(float4) + kBitmapWords * 4 + 16 + static_cast<size_t>(nw) * kListWords * 4 +
                     static_cast<size_t>(cells) * 2;
  if (lds > 160 * 1024) return (ctx->last_box_refusal = DLIOM_BOX_REFUSED_LDS, DLIOM_ERR_CAPACITY);
  typedef void (*BoxKernel)(GridView, Params, const float*, const float*, const float*);
  const BoxKernel kernel = variant == 0 ? rtcsm_score_box_kernel : (variant == 1 ? rtcsm_score_box_kernel_w3 : rtcsm_score_box_kernel_wide);
  const unsigned attr_bit = variant == 0 ? kFuncAttrScoreBox : (variant == 1 ? kFuncAttrScoreBoxW3 : kFuncAttrScoreBoxWide);
  if ((ctx->func_attr_set & attr_bit) == 0u) {
    DLIOM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ctx->func_attr_set |= attr_bit;
  }
  ctx->last_box_variant = variant;
  // workgroups: one round of residents (every wave walks an equal share of the point chunks, so a second,
  // partly filled round would only idle); `target_waves` overrides
  static thread_local size_t resident_lds = 0;
  static thread_local int resident = 0, resident_nw = 0, resident_variant = -1;
  if (resident_lds != lds || resident_nw != nw || resident_variant != variant) {
    resident_nw = nw;
    resident_variant = variant;
    DLIOM_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, reinterpret_cast<const void*>(kernel), 64 * nw, lds));
    resident_lds = lds;
  }
  const int num_cus = ctx->num_cus;
  const int want_blocks = target_waves > 0 ? target_waves / nw : std::max(1, resident) * num_cus;
  int slot_quads = std::max(1, want_blocks / std::max(1, rot_blocks * passes));
  slot_quads = std::min(slot_quads, p.point_chunks);
  // (32-bit register accumulators: the kernel adds them to the 64-bit volume every box::kFlushPoints points)
  p.slots = slot_quads;
  p.units = passes * rot_blocks;
  DLIOM_TRY(ctx->box_counters.reserve(static_cast<size_t>(passes) * rot_blocks * 4 + 256));
  if (!prep_add(prep, ctx->box_counters.p, nullptr, static_cast<size_t>(passes) * rot_blocks * 4, 0u))
    DLIOM_HIP_TRY(hipMemsetAsync(ctx->box_counters.p, 0, static_cast<size_t>(passes) * rot_blocks * 4, ctx->stream));
  p.counters = ctx->box_counters.as<unsigned>();
  // per-(rotation block, point) extents of the lookups: the boxes' bounding boxes are reductions over these
  p.ext_stride = static_cast<int>(cloud.n_padded);
  DLIOM_TRY(ctx->box_extents.reserve(static_cast<size_t>(rot_blocks) * 6 * static_cast<size_t>(cloud.n_padded) * 4));
  p.ext = ctx->box_extents.as<float>();
  {
    // the pre-pass is the first kernel of the match: one more row of workgroups carries the pending copies and fills
    PrepArgs none;
    none.n = 0;
    const PrepArgs& jobs = prep != nullptr ? *prep : none;
    const unsigned prep_rows = jobs.n > 0 ? 1u : 0u;
    hipLaunchKernelGGL(rtcsm_box_extent_kernel, dim3(static_cast<unsigned>((n + 255) / 256), static_cast<unsigned>(rot_blocks) + prep_rows),
                       dim3(256), 0, ctx->stream, p, g.inv_resolution, cloud.d_xs, cloud.d_ys, cloud.d_zs, ctx->box_extents.as<float>(),
                       group_src, d.rot_src, jobs);
    if (prep != nullptr) prep->n = 0;
  }
  const unsigned blocks = static_cast<unsigned>(slot_quads) * passes * rot_blocks;
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64 * nw), lds, ctx->stream, g, p, cloud.d_xs, cloud.d_ys, cloud.d_zs);
  DLIOM_HIP_TRY(hipGetLastError());
#ifdef DLIOM_TEST_HOOKS  // libdliom_hooks.so only (make hooks): as if the kernel had flagged an inconsistency
  if (ctx->tuning[DLIOM_TUNE_RESERVED_TEST_HOOK] == 1) {  // 1 only: 2 and 3 are the de-skew check's hooks (preprocess.hip)
    ctx->tuning[DLIOM_TUNE_RESERVED_TEST_HOOK] = 0;
    DLIOM_HIP_TRY(hipMemsetAsync(static_cast<char*>(ctx->box_error.p) + 4, 1, 1, ctx->stream));
  }
#endif
  return DLIOM_OK;
}

// Launches the score-volume kernel; *pad_processed = padding points visited (each adds 1 to
// every sum).
static int run_score_volume(dliom_ctx* ctx, const dliom_cloud& cloud, const dliom_grid* grid,
                            const Candidates& c, int r_first, int r_last, DeviceCandidates* d,
                            unsigned long long** d_sums, int64_t* pad_processed, const FillJob* also_fill = nullptr) {
  // candidate tables, the zeroed score volume (and the caller's fill): pending jobs that the first kernel of the chain
  // carries (the box kernel's pre-pass) or one launch of their own (prep_flush) -- not a packet each
  PrepArgs prep;
  prep.n = 0;
  DLIOM_TRY(upload_candidates(ctx, c, d, &prep));
  const int64_t C = c.w.num_candidates;
  DLIOM_TRY(ctx->sums.reserve(static_cast<size_t>(C) * 8));
  *d_sums = ctx->sums.as<unsigned long long>();
  {
    FillJob fills[2] = {{*d_sums, static_cast<size_t>(C) * 8, 0u}, {nullptr, 0, 0u}};
    int nf = 1;
    if (also_fill != nullptr) fills[nf++] = *also_fill;
    for (int k = 0; k < nf; ++k)
      if (!prep_add(&prep, fills[k].p, nullptr, fills[k].bytes, fills[k].value)) DLIOM_TRY(fill_multi(ctx, &fills[k], 1));
  }
  const int R = static_cast<int>(c.w.num_rotations), T = static_cast<int>(c.w.num_translations);
  const int n = static_cast<int>(cloud.n);
  // 3: LDS-box kernel over the dense mirror (score_box.h; default when the search suits it),
  // 2: rotation per lane over the dense mirror, 1: rotation per lane over the leaf table,
  // 0: point per lane over the leaf table
  int mapping = ctx->tuning[DLIOM_TUNE_SCORE_KERNEL];
  ctx->last_box_refusal = mapping == 3 ? DLIOM_BOX_RAN : DLIOM_BOX_NOT_REQUESTED;
  if (ctx->force_dense_score && mapping > 2) {
    mapping = 2;
    ctx->last_box_refusal = DLIOM_BOX_REFUSED_FLAGGED;  // the box kernel flagged an inconsistency: this is the rerun
  }
  ctx->last_score_used_box = false;
  if (mapping >= 2) {
    // the mirror is a cache of the grid's contents: building it does not change the grid.  Grids beyond bits = 4 get a
    // WINDOW around this match's initial pose that holds every cell the search can read: a rotated point lies within
    // max ||p|| of the candidate's translation, and that within (L + 1) res sqrt(3) of the initial one
    const float res = grid->resolution;
    const int centre[3] = {static_cast<int>(std::lround(c.init.t.x / res)), static_cast<int>(std::lround(c.init.t.y / res)),
                           static_cast<int>(std::lround(c.init.t.z / res))};
    const double reach_cells = static_cast<double>(cloud.max_norm) / res + (c.w.linear_window_size + 1) * 1.7321 + 4.0;
    const int radius = reach_cells < 1.0e6 ? static_cast<int>(std::ceil(reach_cells)) : (1 << 20);
    if (const_cast<dliom_grid*>(grid)->ensure_dense_for(centre, radius) != DLIOM_OK) {
      if (mapping == 3) ctx->last_box_refusal = DLIOM_BOX_REFUSED_NO_MIRROR;
      mapping = 1;  // too large: leaf path
    }
  }
  // a windowed mirror has its own offset per axis: only the box kernel reads it, everything else takes the leaf table
  const bool windowed = grid->dense_windowed;
  if (windowed && mapping == 2) mapping = 1;
  const GridView g = grid->view();
  DLIOM_TRY(ensure_morton(ctx, &cloud));
  if (g.log2_leaves > 10 && mapping >= 1) {  // bits = 8: only the point-per-lane kernel has 64-bit table indices
    if (mapping == 3) ctx->last_box_refusal = DLIOM_BOX_REFUSED_NO_MIRROR;
    mapping = 0;
  }
  if (mapping == 3) {
    static const int box_min_pairs_log2 = env_int("DLIOM_BOX_MIN_LOG2", 24);  // small searches: launch-bound anyway
    const double pairs = static_cast<double>(C) * static_cast<double>(n);
    int s3 = DLIOM_ERR_CAPACITY;
    ctx->last_box_refusal = DLIOM_BOX_REFUSED_SMALL;  // unless the launch below is reached
    if (T >= 8 && pairs >= std::ldexp(1.0, box_min_pairs_log2)) {
      ctx->last_box_refusal = DLIOM_BOX_RAN;
#ifdef DLIOM_EXPERIMENTS
      constexpr size_t kBoxErrorBytes = 256 + 4096 * 32;  // + per-workgroup time stamps (Params::debug & 256)
#else
      constexpr size_t kBoxErrorBytes = 256;
#endif
      DLIOM_TRY(ctx->box_error.reserve(kBoxErrorBytes));
      if (!ctx->box_error_zeroed) {
        DLIOM_HIP_TRY(hipMemsetAsync(ctx->box_error.p, 0, kBoxErrorBytes, ctx->stream));
        ctx->box_error_zeroed = true;
      }
      s3 = launch_score_box(ctx, cloud, g, c, *d, r_first, r_last, *d_sums, ctx->box_error.as<unsigned>(), &prep);
    }
    if (s3 == DLIOM_OK) {
      *pad_processed = 0;
      ctx->last_score_used_box = true;
      ctx->last_score_mapping = 3;
      return DLIOM_OK;
    }
    if (s3 != DLIOM_ERR_CAPACITY) return s3;
    mapping = windowed ? 1 : 2;
  }
  DLIOM_TRY(prep_flush(ctx, &prep));  // the kernels below read the device copies
  static const int forced_ppt = env_int("DLIOM_SCORE_PPT", 0);   // tuning knobs
  static const int target_blocks = env_int("DLIOM_SCORE_BLOCKS", 8192);
  const int Rs = r_last - r_first;  // rotations of this shard
  int span = -1;
  int64_t processed = 0;
  ctx->last_score_mapping = mapping;
  if (mapping >= 1) {
    static const int forced_chunk = env_int("DLIOM_SCORE_CHUNK", 0);
    const int rot_groups = (Rs + kBlock - 1) / kBlock;
    // points per chunk: a multiple of 4 dividing the 4096-point padding; aim at >= target blocks
    int chunk = 4096;
    while (chunk > 64 && static_cast<int64_t>(rot_groups) * ((n + chunk - 1) / chunk) < target_blocks) chunk >>= 1;
    if (forced_chunk > 0) chunk = forced_chunk;
    const int point_chunks = (n + chunk - 1) / chunk;
    processed = static_cast<int64_t>(point_chunks) * chunk;
    const dim3 grid_dim(rot_groups, point_chunks), block(kBlock);
    span = ctx->begin_span(DLIOM_KERNEL_RTCSM_SCORE);
    static const int pts_per_iter = env_int("DLIOM_SCORE_P", 8);
    const size_t lds = static_cast<size_t>(T) * 16;
    if (lds > 100 * 1024) return DLIOM_ERR_INVALID_ARGUMENT;
    if (mapping == 2) {

      // block size: whole wavefronts, the fewest idle lanes in the last rotation group
      static const int forced_bs = env_int("DLIOM_SCORE_BLOCK", 0);
      static const int max_bs = env_int("DLIOM_SCORE_MAX_BLOCK", kDenseMaxBlock);
      int bs = max_bs;
      for (int cand = max_bs; cand >= 64; cand -= 64)
        if ((Rs + cand - 1) / cand * cand < (Rs + bs - 1) / bs * bs) bs = cand;
      if (forced_bs >= 64 && forced_bs <= kDenseMaxBlock && forced_bs % 64 == 0) bs = forced_bs;
      // small searches (a few hundred points after the adaptive voxel filter) would fill a handful
      // of workgroups: go down to one wavefront per group and 8-point chunks to spread them
      if (forced_bs == 0 && static_cast<int64_t>((Rs + bs - 1) / bs) * ((n + 63) / 64) < 256) bs = 64;
      const int rot_groups = (Rs + bs - 1) / bs;
      int chunk = 4096;
      while (chunk > 8 && static_cast<int64_t>(rot_groups) * ((n + chunk - 1) / chunk) < target_blocks) chunk >>= 1;
      if (forced_chunk > 0) chunk = forced_chunk;
      const int point_chunks = (n + chunk - 1) / chunk;
      processed = static_cast<int64_t>(point_chunks) * chunk;
      const dim3 block(bs);
      // Table domain: all real lookups lie within max ||p|| + max |translation component| of the origin.
      DenseDomain dom;
      dom.lo = 0;
      dom.size = g.dense_stride;
      dom.n_real = n;
      dom.pad[0] = dom.pad[1] = dom.pad[2] = kPadCoordinate;
      bool clamp = true;
      {
        float tmax = 0.f, t_init = norm3(c.init.t);
        for (const F3& t : c.trans) tmax = std::max(tmax, std::max(std::fabs(t.x), std::max(std::fabs(t.y), std::fabs(t.z))));
        const float res = g.resolution;
        const double reach = (static_cast<double>(cloud.max_norm) + tmax) / res + 3.0;  // cells from the origin
        // padding point: W = ((half + m) res, 0, 0) in the grid frame, mapped back through the initial
        // pose; under a candidate it moves by at most theta_max |W - t_init| + (L + 1) res sqrt(3)
        const double theta_max = 1.7321 * (c.w.angular_window_size + 1) * c.w.angular_step_size * 1.01 + 1e-4;
        const double rho = (g.half + 64) * static_cast<double>(res) + t_init;
        const double m_cells = 8.0 + std::ceil((theta_max * rho + (c.w.linear_window_size + 1) * res * 1.7321) / res);
        // the padding point itself may be carried up to (m - 8) cells outward as well
        const double hi_cell = std::max(reach, g.half + 2.0 * m_cells);
        const int lo_i = std::min(0, g.half + 1 - static_cast<int>(std::ceil(hi_cell)));
        const int hi_i = std::max(g.dense_stride, g.half + 1 + static_cast<int>(std::ceil(hi_cell)) + 1);
        static const int no_clamp = env_int("DLIOM_SCORE_NO_CLAMP", 1);
        if (no_clamp && m_cells <= 64.0 && hi_i - lo_i <= 1024 && std::isfinite(reach)) {
          clamp = false;
          dom.lo = lo_i;
          dom.size = hi_i - lo_i;
          const F3 w{static_cast<float>((g.half + m_cells) * res) - c.init.t.x, -c.init.t.y, -c.init.t.z};
          const QF qi{c.init.q.w, -c.init.q.x, -c.init.q.y, -c.init.q.z};
          const F3 p = qrot(qi, w);
          dom.pad[0] = p.x;
          dom.pad[1] = p.y;
          dom.pad[2] = p.z;
        }
      }
      // translations per pass over the points: as many as keep >= 4 workgroups per CU in LDS (160 KB);
      // large windows / big grids trade a few extra point rotations for occupancy
      static const int t_chunk_max = env_int("DLIOM_SCORE_TCHUNK", 27);
      const size_t fixed_lds = lds + static_cast<size_t>(dom.size) * 12;
      const size_t acc_budget = fixed_lds < 34 * 1024 ? 36 * 1024 - fixed_lds : 2 * 1024;
      int t_chunk = std::max(4, std::min(std::min(T, t_chunk_max), static_cast<int>(acc_budget / (static_cast<size_t>(bs) * 4))));
      const int chunks_per_xcd = (point_chunks + 7) / 8;
      // small searches (the reference's ~170 filtered points): translation slices over gridDim.y until ~4 workgroups per CU
      static const int small_slices = env_int("DLIOM_SCORE_SLICES", 1);
      int t_slices = 1;
      const int base_blocks = 8 * chunks_per_xcd * rot_groups;
      if (small_slices != 0 && base_blocks < 512 && T > 1) {
        const int want = std::min(T, (1024 + base_blocks - 1) / base_blocks);
        t_chunk = std::min(t_chunk, (T + want - 1) / want);
        t_slices = (T + t_chunk - 1) / t_chunk;
      }
      const size_t lds2 = fixed_lds + static_cast<size_t>(t_chunk) * bs * 4;
      const dim3 dense_grid(base_blocks, t_slices);
#define DLIOM_LAUNCH_DENSE(PP, CL)                                                                                    \
  hipLaunchKernelGGL((rtcsm_score_dense_kernel<PP, CL>), dense_grid, block, lds2, ctx->stream, g, dom, cloud.d_xs,     \
                     cloud.d_ys, cloud.d_zs, chunk, point_chunks, rot_groups, d->rot, R, r_first, r_last, d->trans4, T, \
                     t_chunk, *d_sums)
      static const int score_debug = env_int("DLIOM_SCORE_DEBUG", 0);  // timing-only pipe isolation
      if (score_debug >= 1 && score_debug <= 3 && !clamp) {
#define DLIOM_LAUNCH_DEBUG(DBG)                                                                                       \
  hipLaunchKernelGGL((rtcsm_score_dense_kernel<8, false, DBG>), dense_grid, block, lds2, ctx->stream, g, dom,         \
                     cloud.d_xs, cloud.d_ys, cloud.d_zs, chunk, point_chunks, rot_groups, d->rot, R, r_first, r_last, \
                     d->trans4, T, t_chunk, *d_sums)
        if (score_debug == 1) DLIOM_LAUNCH_DEBUG(1);
        else if (score_debug == 2) DLIOM_LAUNCH_DEBUG(2);
        else DLIOM_LAUNCH_DEBUG(3);
#undef DLIOM_LAUNCH_DEBUG
      } else if (pts_per_iter == 2) {
        if (clamp) DLIOM_LAUNCH_DENSE(2, true); else DLIOM_LAUNCH_DENSE(2, false);
      } else if (pts_per_iter == 4) {
        if (clamp) DLIOM_LAUNCH_DENSE(4, true); else DLIOM_LAUNCH_DENSE(4, false);
      } else {
        if (clamp) DLIOM_LAUNCH_DENSE(8, true); else DLIOM_LAUNCH_DENSE(8, false);
      }
#undef DLIOM_LAUNCH_DENSE
    } else if (T == 1) {
      hipLaunchKernelGGL((rtcsm_score_rot_kernel<1, 4>), grid_dim, block, lds, ctx->stream, g, cloud.d_xs,
                         cloud.d_ys, cloud.d_zs, chunk, d->rot, R, r_first, r_last, d->trans4, T, *d_sums);
    } else if (pts_per_iter == 2) {
      hipLaunchKernelGGL((rtcsm_score_rot_kernel<27, 2>), grid_dim, block, lds, ctx->stream, g, cloud.d_xs,
                         cloud.d_ys, cloud.d_zs, chunk, d->rot, R, r_first, r_last, d->trans4, T, *d_sums);
    } else if (pts_per_iter == 8) {
      hipLaunchKernelGGL((rtcsm_score_rot_kernel<27, 8>), grid_dim, block, lds, ctx->stream, g, cloud.d_xs,
                         cloud.d_ys, cloud.d_zs, chunk, d->rot, R, r_first, r_last, d->trans4, T, *d_sums);
    } else {
      hipLaunchKernelGGL((rtcsm_score_rot_kernel<27, 4>), grid_dim, block, lds, ctx->stream, g, cloud.d_xs,
                         cloud.d_ys, cloud.d_zs, chunk, d->rot, R, r_first, r_last, d->trans4, T, *d_sums);
    }
  } else {
    static const int debug_no_atomic = env_int("DLIOM_DEBUG_NO_ATOMIC", 0);
    int ppt = forced_ppt > 0 ? forced_ppt : (n >= 32 * 1024 ? 8 : (n >= 8 * 1024 ? 4 : (n >= 2048 ? 2 : 1)));
    while (ppt > 1 && (cloud.n_padded % (static_cast<int64_t>(kBlock) * ppt)) != 0) ppt >>= 1;
    const int tile = kBlock * ppt;
    const int point_tiles = (n + tile - 1) / tile;
    processed = static_cast<int64_t>(point_tiles) * tile;
    int rot_tiles = std::max(1, std::min(Rs, (target_blocks + point_tiles - 1) / point_tiles));
    const int rots_per_block = (Rs + rot_tiles - 1) / rot_tiles;
    rot_tiles = (Rs + rots_per_block - 1) / rots_per_block;
    const dim3 grid_dim(point_tiles, rot_tiles), block(kBlock);
    const size_t lds = static_cast<size_t>(T) * sizeof(unsigned);
    if (lds > 150 * 1024) return DLIOM_ERR_INVALID_ARGUMENT;  // (2L+1)^3 translations must fit LDS
    span = ctx->begin_span(DLIOM_KERNEL_RTCSM_SCORE);
#define DLIOM_LAUNCH_SCORE(PP)                                                                    \
  hipLaunchKernelGGL((rtcsm_score_kernel<PP>), grid_dim, block, lds, ctx->stream, g, cloud.d_xs,  \
                     cloud.d_ys, cloud.d_zs, d->rot, R, r_first, r_last, d->trans, T,             \
                     rots_per_block, *d_sums, debug_no_atomic)
    if (g.log2_leaves > 10) {  // bits = 8: the leaf table has 2^33 entries
      if (ppt >= 4)
        hipLaunchKernelGGL((rtcsm_score_kernel<4, true>), dim3((n + 4 * kBlock - 1) / (4 * kBlock), rot_tiles), block, lds, ctx->stream,
                           g, cloud.d_xs, cloud.d_ys, cloud.d_zs, d->rot, R, r_first, r_last, d->trans, T, rots_per_block, *d_sums,
                           debug_no_atomic);
      else
        hipLaunchKernelGGL((rtcsm_score_kernel<1, true>), dim3((n + kBlock - 1) / kBlock, rot_tiles), block, lds, ctx->stream, g,
                           cloud.d_xs, cloud.d_ys, cloud.d_zs, d->rot, R, r_first, r_last, d->trans, T, rots_per_block, *d_sums,
                           debug_no_atomic);
      processed = static_cast<int64_t>((n + (ppt >= 4 ? 4 : 1) * kBlock - 1) / ((ppt >= 4 ? 4 : 1) * kBlock)) * (ppt >= 4 ? 4 : 1) * kBlock;
    } else
    switch (ppt) {
      case 16: DLIOM_LAUNCH_SCORE(16); break;
      case 8: DLIOM_LAUNCH_SCORE(8); break;
      case 4: DLIOM_LAUNCH_SCORE(4); break;
      case 2: DLIOM_LAUNCH_SCORE(2); break;
      default: DLIOM_LAUNCH_SCORE(1); break;
    }
#undef DLIOM_LAUNCH_SCORE
  }
  ctx->end_span(span);
  DLIOM_HIP_TRY(hipGetLastError());
  *pad_processed = processed - n;
  return DLIOM_OK;
}

// LUT constants of probability_values.cc:27-36 in float, plus the affine fit used by the bounds.
struct LutModel {
  float k_scale, k_offset, k_unknown;
  double a, b, delta;
};
static const LutModel& lut_model() {
  static const LutModel m = [] {
    LutModel r;
    const float kMin = 0.1f, kMax = 1.f - 0.1f;
    r.k_scale = (kMax - kMin) / 32766.f;
    r.k_offset = kMin - r.k_scale;
    r.k_unknown = kMin;
    r.a = static_cast<double>(r.k_scale);
    r.b = static_cast<double>(r.k_offset);
    double d = std::fabs(static_cast<double>(kMin) - (r.a * 1.0 + r.b));  // value 0 counted as 1
    for (int v = 1; v < 32768; ++v) {
      const float f = v * r.k_scale + r.k_offset;
      d = std::max(d, std::fabs(static_cast<double>(f) - (r.a * v + r.b)));
    }
    r.delta = d * 1.0000001 + 1e-12;
    return r;
  }();
  return m;
}

// State of a (possibly sharded) match between its phases; lives in dliom_ctx::rtcsm_state.
// Launches the exact sequential-sum kernels for `count` candidates whose indices are in d_list
// (c -> translation c / R, rotation c % R): method 0 = one lane replays the loop, 1 = element scan,
// 2 = chunk scan (default).  Methods 1 and 2 need the 15-bit values of a candidate in LDS
// (n <= 65536) and fall back to method 0 beyond.  `scratch` receives values / chunk sums / chunk
// functions; d_count != nullptr: the kernels read the live candidate count on the device.
static int rescore_method_default() {
  static const int m = env_int("DLIOM_RESCORE", 2);
  return m;
}
// Small clouds (the reference's ~170 filtered points): the serial replay is one launch of ~5 us, the chunk scan three;
// all methods return identical bits (test_sequential_sum_kernels_bit_exact).
static int rescore_method_for(int64_t n) {
  static const int small_n = env_int("DLIOM_RESCORE_SERIAL_BELOW", 1024);
  return n <= small_n ? 0 : rescore_method_default();
}
static int launch_sequential_sums(dliom_ctx* ctx, int method, const GridView& gv, const dliom_cloud& cloud,
                                  const float4* d_rot, int R, const float* d_trans, const unsigned* d_list,
                                  const unsigned* d_count, unsigned count, DevBuf* scratch, float* d_ksums) {
  const LutModel& lm = lut_model();
  const int n = static_cast<int>(cloud.n);
  const int n_stride = (n + 7) & ~7;
  const size_t scan_lds = static_cast<size_t>(n_stride) * 2;
  if (method == 0 || scan_lds > 128 * 1024 || count > 65535) {
    hipLaunchKernelGGL(rtcsm_rescore_kernel, dim3(count), dim3(256), 0, ctx->stream, gv, cloud.d_x, cloud.d_y, cloud.d_z, n,
                       d_rot, R, d_trans, d_list, d_count, lm.k_scale, lm.k_offset, lm.k_unknown, d_ksums);
    DLIOM_HIP_TRY(hipGetLastError());
    return DLIOM_OK;
  }
  const int num_chunks = (n + kChunk - 1) / kChunk;
  const size_t values_bytes = (static_cast<size_t>(count) * n_stride * 2 + 255) & ~static_cast<size_t>(255);
  const size_t sums_bytes = (static_cast<size_t>(count) * num_chunks * 8 + 255) & ~static_cast<size_t>(255);
  const size_t fns_bytes = static_cast<size_t>(count) * num_chunks * sizeof(ChunkFns);
  DLIOM_TRY(scratch->reserve(values_bytes + sums_bytes + fns_bytes));
  char* base = static_cast<char*>(scratch->p);
  unsigned short* d_values = reinterpret_cast<unsigned short*>(base);
  double* d_chunk_sums = method == 2 ? reinterpret_cast<double*>(base + values_bytes) : nullptr;
  ChunkFns* d_fns = reinterpret_cast<ChunkFns*>(base + values_bytes + sums_bytes);
  hipLaunchKernelGGL(rtcsm_rescore_values_kernel, dim3((n_stride + 255) / 256, count), dim3(256), 0, ctx->stream, gv,
                     cloud.d_x, cloud.d_y, cloud.d_z, n, n_stride, d_rot, R, d_trans, d_list, d_count, d_values, lm.k_scale,
                     lm.k_offset, lm.k_unknown, d_chunk_sums, num_chunks);
  if (method == 2) {
    hipLaunchKernelGGL(rtcsm_rescore_chunk_fns_kernel, dim3((num_chunks + 3) / 4, count), dim3(256), 0, ctx->stream, d_values,
                       n, n_stride, lm.k_scale, lm.k_offset, lm.k_unknown, d_count, d_chunk_sums, num_chunks, d_fns);
    hipLaunchKernelGGL(rtcsm_rescore_chunk_scan_kernel, dim3(count), dim3(kScanThreads), scan_lds, ctx->stream, d_values, n,
                       n_stride, lm.k_scale, lm.k_offset, lm.k_unknown, d_count, d_fns, num_chunks, d_ksums);
  } else {
    hipLaunchKernelGGL(rtcsm_rescore_scan_kernel, dim3(count), dim3(kScanThreads), scan_lds, ctx->stream, d_values, n, n_stride,
                       lm.k_scale, lm.k_offset, lm.k_unknown, d_count, d_ksums);
  }
  DLIOM_HIP_TRY(hipGetLastError());
  return DLIOM_OK;
}

struct RtcsmState {
  dliom_rtcsm_options o;
  Candidates c;
  DeviceCandidates d;
  dliom_cloud cloud;
  const dliom_grid* grid = nullptr;
  int r_first = 0, r_last = 0;
  unsigned long long* d_sums = nullptr;
  float* d_hi = nullptr;
  unsigned* d_list = nullptr;
  unsigned* d_ctrs = nullptr;
  float best_score = -1.f;
  int64_t best_c = -1;
  bool active = false;
  double init7[7] = {0, 0, 0, 1, 0, 0, 0};
  int shard = 0, num_shards = 1;
  bool used_box = false;
};

// The LDS-box kernel checks its own fast index against the reference's arithmetic wherever the two could differ; an
// exact cell more than one cell away from the fast one contradicts the error budget of score_box.h ("cannot
// happen").  If it ever does, the kernel sets box_error[0] (sticky, dliom_rtcsm3d_box_error) and box_error[1]: the
// sums are then not trusted and the match is redone with the dense kernel, which has no fast path.
// A test walks this path in a build with -DDLIOM_TEST_HOOKS (libdliom_hooks.so), where knob 2 of dliom_ctx_set_tuning sets the
// word from the host; the library that ships has no such switch.
// `err1` is the word as read back; clears it on the device.
static int box_overflowed(dliom_ctx* ctx, unsigned err1, bool* overflow) {
  *overflow = err1 != 0u;
  if (*overflow)
    DLIOM_HIP_TRY(hipMemsetAsync(static_cast<char*>(ctx->box_error.p) + 4, 0, 4, ctx->stream));
  return DLIOM_OK;
}

static RtcsmState* state_of(dliom_ctx* ctx) {
  if (ctx->rtcsm_state == nullptr) {
    ctx->rtcsm_state = new RtcsmState;
    ctx->rtcsm_state_free = [](void* p) { delete static_cast<RtcsmState*>(p); };
  }
  return static_cast<RtcsmState*>(ctx->rtcsm_state);
}

// Phase 1: score volume + score bounds for the rotations [shard * R / num_shards, ...).
// local_best_lo_bits (optional) receives this shard's best lower bound (float bits) -- the
// quantity that is max-all-reduced when the window is sharded across GPUs.
static int match_begin(dliom_ctx* ctx, const dliom_rtcsm_options* o, const double init7[7],
                       const dliom_cloud& cloud, const dliom_grid* grid, int shard, int num_shards,
                       unsigned* local_best_lo_bits) {
  if (cloud.n <= 0) return DLIOM_ERR_EMPTY_CLOUD;
  if (cloud.n > (1 << 27) || num_shards < 1 || shard < 0 || shard >= num_shards)
    return DLIOM_ERR_INVALID_ARGUMENT;
  RtcsmState* st = state_of(ctx);
  st->active = false;
  st->o = *o;
  st->cloud = cloud;
  st->grid = grid;
  std::memcpy(st->init7, init7, sizeof st->init7);
  st->shard = shard;
  st->num_shards = num_shards;
  st->used_box = false;
  Candidates& c = st->c;
  generate_candidates(*o, grid->resolution, cloud.max_norm, init7, &c);
  const int64_t C = c.w.num_candidates;
  const int R = static_cast<int>(c.w.num_rotations), T = static_cast<int>(c.w.num_translations);
  if (C <= 0 || C > (int64_t{1} << 31)) return DLIOM_ERR_INVALID_ARGUMENT;
  st->r_first = static_cast<int>(static_cast<int64_t>(R) * shard / num_shards);
  st->r_last = static_cast<int>(static_cast<int64_t>(R) * (shard + 1) / num_shards);

  const size_t bytes_f = (static_cast<size_t>(C) * 4 + 255) & ~static_cast<size_t>(255);
  DLIOM_TRY(ctx->bounds.reserve(3 * bytes_f + 256));
  char* bb = static_cast<char*>(ctx->bounds.p);
  float* d_lo = reinterpret_cast<float*>(bb);
  st->d_hi = reinterpret_cast<float*>(bb + bytes_f);
  st->d_list = reinterpret_cast<unsigned*>(bb + 2 * bytes_f);
  st->d_ctrs = reinterpret_cast<unsigned*>(bb + 3 * bytes_f);  // [0] best_lo bits, [1] count
  const FillJob zero_ctrs{st->d_ctrs, 8, 0u};
  if (st->r_last <= st->r_first) DLIOM_TRY(fill_multi(ctx, &zero_ctrs, 1));
  if (st->r_last > st->r_first) {
    int64_t pad_processed = 0;
    DLIOM_TRY(run_score_volume(ctx, cloud, grid, c, st->r_first, st->r_last, &st->d, &st->d_sums,
                               &pad_processed, &zero_ctrs));
    st->used_box = ctx->last_score_used_box;
    const LutModel& lm = lut_model();
    BoundParams bp;
    bp.a = lm.a;
    bp.b = lm.b;
    bp.delta = lm.delta;
    bp.wt = o->translation_delta_cost_weight;
    bp.wr = o->rotation_delta_cost_weight;
    bp.n = static_cast<int>(cloud.n);
    bp.n_pad = static_cast<int>(pad_processed);
    bp.R = R;
    bp.T = T;
    bp.r_first = st->r_first;
    bp.r_last = st->r_last;
    const unsigned cblocks = static_cast<unsigned>((C + 255) / 256);
    const int span = ctx->begin_span(DLIOM_KERNEL_RTCSM_SELECT);
    hipLaunchKernelGGL(rtcsm_bounds_kernel, dim3(cblocks), dim3(256), 0, ctx->stream, st->d_sums,
                       static_cast<long long>(C), st->d.t_norm, st->d.r_angle, bp, d_lo, st->d_hi,
                       st->d_ctrs);
    ctx->end_span(span);
    DLIOM_HIP_TRY(hipGetLastError());
  }
  if (local_best_lo_bits != nullptr) {
    unsigned* h_err = reinterpret_cast<unsigned*>(static_cast<char*>(ctx->pinned) + ctx->pinned_bytes - 4096 + 2048);
    *h_err = 0u;
    DLIOM_HIP_TRY(hipMemcpyAsync(local_best_lo_bits, st->d_ctrs, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (st->used_box)
      DLIOM_HIP_TRY(hipMemcpyAsync(h_err, static_cast<char*>(ctx->box_error.p) + 4, 4, hipMemcpyDeviceToHost, ctx->stream));
    DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    bool overflow = false;
    DLIOM_TRY(box_overflowed(ctx, *h_err, &overflow));
    if (overflow && !ctx->force_dense_score) {
      ctx->force_dense_score = true;
      const dliom_rtcsm_options o_copy = st->o;
      const int s = match_begin(ctx, &o_copy, init7, cloud, grid, shard, num_shards, local_best_lo_bits);
      ctx->force_dense_score = false;
      return s;
    }
  }
  st->active = true;
  return DLIOM_OK;
}

// Phase 2: survivors against the (global) best lower bound, exact sequential rescoring, and this
// shard's winner packed as (score_bits << 32) | (0xFFFFFFFF - index): the max over shards of that
// word is the reference's first-maximum-in-generation-order winner.
static int match_finish(dliom_ctx* ctx, const unsigned* global_best_lo_bits, uint64_t* local_best_packed) {
  RtcsmState* st = state_of(ctx);
  if (!st->active) return DLIOM_ERR_INVALID_ARGUMENT;
  const dliom_rtcsm_options* o = &st->o;
  const Candidates& c = st->c;
  const int64_t C = c.w.num_candidates;
  const int R = static_cast<int>(c.w.num_rotations);
  const dliom_cloud& cloud = st->cloud;
  st->best_score = -1.f;
  st->best_c = -1;
  unsigned K = 0;
  if (st->r_last > st->r_first) {
    if (global_best_lo_bits != nullptr) {
      DLIOM_HIP_TRY(hipMemcpyAsync(st->d_ctrs, global_best_lo_bits, 4, hipMemcpyHostToDevice, ctx->stream));
    }
    const unsigned cblocks = static_cast<unsigned>((C + 255) / 256);
    int span = ctx->begin_span(DLIOM_KERNEL_RTCSM_SELECT);
    hipLaunchKernelGGL(rtcsm_select_kernel, dim3(cblocks), dim3(256), 0, ctx->stream, st->d_hi,
                       static_cast<long long>(C), st->d_ctrs, st->d_ctrs + 1, st->d_list);
    ctx->end_span(span);
    DLIOM_HIP_TRY(hipGetLastError());
  }
  // Exact sequential rescoring of the survivors, one workgroup each.  Usually a handful survive, so
  // the first kSpecK are rescored SPECULATIVELY in the same stream segment (the kernels read the
  // count on the device and idle blocks exit): one synchronisation for the whole match.  Only if
  // more survived does a second, exactly sized round run.
  constexpr unsigned kSpecK = 8;
  std::vector<unsigned> list;
  std::vector<float> ksums;
  if (st->r_last > st->r_first) {
    // readback block at the end of the pinned staging area: [count pair | list | sums]
    unsigned* h_ctrs = reinterpret_cast<unsigned*>(static_cast<char*>(ctx->pinned) + ctx->pinned_bytes - 4096);
    unsigned* h_list = h_ctrs + 2;
    float* h_sums = reinterpret_cast<float*>(h_ctrs + 2 + kSpecK);
    auto rescore = [&](unsigned count, const unsigned* d_count, size_t list_offset) -> int {
      DLIOM_TRY(ctx->rescore.reserve((static_cast<size_t>(count) * 4 + 255) & ~static_cast<size_t>(255)));
      const int span = ctx->begin_span(DLIOM_KERNEL_RTCSM_RESCORE);
      const int s = launch_sequential_sums(ctx, rescore_method_for(cloud.n), st->grid->view(), cloud, st->d.rot, R, st->d.trans,
                                           st->d_list + list_offset, d_count, count, &ctx->misc, ctx->rescore.as<float>());
      ctx->end_span(span);
      return s;
    };
    DLIOM_TRY(rescore(kSpecK, st->d_ctrs + 1, 0));
    // one read-back dispatch: [count pair | list | sums | box overflow word]
    unsigned* h_err = h_ctrs + 2 + 2 * kSpecK;
    *h_err = 0u;
    const bool want_err = st->used_box && global_best_lo_bits == nullptr;
    const GatherJob back[4] = {{st->d_ctrs, 2}, {st->d_list, kSpecK}, {ctx->rescore.p, kSpecK},
                               {static_cast<char*>(ctx->box_error.p) + 4, 1}};
    DLIOM_TRY(gather_and_wait(ctx, back, want_err ? 4 : 3, h_ctrs));
    {
      bool overflow = false;
      DLIOM_TRY(box_overflowed(ctx, *h_err, &overflow));
      if (overflow && !ctx->force_dense_score) {  // redo the whole match on the dense kernel
        ctx->force_dense_score = true;
        const dliom_rtcsm_options o_copy = st->o;
        const dliom_cloud cloud_copy = st->cloud;
        double init_copy[7];
        std::memcpy(init_copy, st->init7, sizeof init_copy);
        int s = match_begin(ctx, &o_copy, init_copy, cloud_copy, st->grid, st->shard, st->num_shards, nullptr);
        if (s == DLIOM_OK) s = match_finish(ctx, nullptr, local_best_packed);
        ctx->force_dense_score = false;
        return s;
      }
    }
    K = h_ctrs[1];
    const unsigned first = std::min(K, kSpecK);
    list.assign(h_list, h_list + first);
    ksums.assign(h_sums, h_sums + first);
    if (K > kSpecK) {  // the rest, exactly sized
      const unsigned rest = K - kSpecK;
      DLIOM_TRY(rescore(rest, nullptr, kSpecK));
      list.resize(K);
      ksums.resize(K);
      DLIOM_HIP_TRY(hipMemcpyAsync(list.data() + kSpecK, st->d_list + kSpecK, static_cast<size_t>(rest) * 4,
                                   hipMemcpyDeviceToHost, ctx->stream));
      DLIOM_HIP_TRY(hipMemcpyAsync(ksums.data() + kSpecK, ctx->rescore.p, static_cast<size_t>(rest) * 4,
                                   hipMemcpyDeviceToHost, ctx->stream));
      DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
  }
  if (K > 0) {
    const int n = static_cast<int>(cloud.n);
    // final scoring exactly as rtcsm_3d.cc:105-112, first maximum in generation order
    for (unsigned k = 0; k < K; ++k) {
      const int64_t cc = list[k];
      const int j = static_cast<int>(cc / R), r = static_cast<int>(cc % R);
      float score = ksums[k];
      score /= static_cast<float>(n);
      const double arg = c.t_norm[j] * o->translation_delta_cost_weight +
                         c.r_angle[r] * o->rotation_delta_cost_weight;
      score *= std::exp(-(arg * (arg * 1.0)));
      if (score > st->best_score || (score == st->best_score && cc < st->best_c)) {
        st->best_score = score;
        st->best_c = cc;
      }
    }
  }
  ctx->last_rtcsm.window = c.w;
  ctx->last_rtcsm.num_points = cloud.n;
  ctx->last_rtcsm.num_rescored = K;
  ctx->last_rtcsm.score_kernel = ctx->last_score_mapping;
  ctx->last_rtcsm.box_kernel_status = ctx->last_box_refusal;
  ctx->last_rtcsm.box_kernel_variant = ctx->last_box_refusal == DLIOM_BOX_RAN ? ctx->last_box_variant : -1;
  ctx->last_rtcsm.best_index = st->best_c;
  if (local_best_packed != nullptr) {
    uint64_t packed = 0;  // a shard without a positive-score survivor contributes nothing
    if (st->best_c >= 0 && st->best_score > 0.f) {
      uint32_t bits;
      std::memcpy(&bits, &st->best_score, 4);
      packed = (static_cast<uint64_t>(bits) << 32) | (0xFFFFFFFFull - static_cast<uint64_t>(st->best_c));
    }
    *local_best_packed = packed;
  }
  return DLIOM_OK;
}

// Phase 3: pose of the winning candidate (every shard can decode it: candidates are replicated).
static int match_decode(dliom_ctx* ctx, uint64_t best_packed, double out7[7], float* out_score) {
  RtcsmState* st = state_of(ctx);
  if (!st->active) return DLIOM_ERR_INVALID_ARGUMENT;
  if (best_packed == 0) return DLIOM_ERR_SCORE_NOT_POSITIVE;  // CHECK_GT(score, 0.f)
  const Candidates& c = st->c;
  const int R = static_cast<int>(c.w.num_rotations);
  const uint32_t bits = static_cast<uint32_t>(best_packed >> 32);
  const int64_t best_c = static_cast<int64_t>(0xFFFFFFFFull - (best_packed & 0xFFFFFFFFull));
  if (best_c < 0 || best_c >= c.w.num_candidates) return DLIOM_ERR_INVALID_ARGUMENT;
  float score;
  std::memcpy(&score, &bits, 4);
  const int j = static_cast<int>(best_c / R), r = static_cast<int>(best_c % R);
  out7[0] = c.trans[j].x;
  out7[1] = c.trans[j].y;
  out7[2] = c.trans[j].z;
  out7[3] = c.rot[r].w;
  out7[4] = c.rot[r].x;
  out7[5] = c.rot[r].y;
  out7[6] = c.rot[r].z;
  *out_score = score;
  ctx->last_rtcsm.best_index = best_c;
  return DLIOM_OK;
}

static int match_impl(dliom_ctx* ctx, const dliom_rtcsm_options* o, const double init7[7],
                      const dliom_cloud& cloud, const dliom_grid* grid, double out7[7],
                      float* out_score) {
#ifdef DLIOM_EXPERIMENTS
  static const int timing = env_int("DLIOM_TIMING", 0);
  const auto t0 = std::chrono::steady_clock::now();
#endif
  DLIOM_TRY(match_begin(ctx, o, init7, cloud, grid, 0, 1, nullptr));
#ifdef DLIOM_EXPERIMENTS
  const auto t1 = std::chrono::steady_clock::now();
#endif
  uint64_t packed = 0;
  DLIOM_TRY(match_finish(ctx, nullptr, &packed));
#ifdef DLIOM_EXPERIMENTS
  const auto t2 = std::chrono::steady_clock::now();
#endif
  const int s = match_decode(ctx, packed, out7, out_score);
#ifdef DLIOM_EXPERIMENTS
  if (timing) {
    const auto t3 = std::chrono::steady_clock::now();
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    std::fprintf(stderr, "TIMING match: begin (host prep + enqueue) %.1f us, finish (enqueue + wait) %.1f us, decode %.1f us\n", us(t0, t1),
                 us(t1, t2), us(t2, t3));
  }
#endif
  return s;
}

}  // namespace dliom

using namespace dliom;

// Sequential float sums, in point order, of the LUT probabilities of `cloud` under k explicit float
// poses: sums[i] = sum_p P(value(cell(q_i * p + t_i))) accumulated like `score += p`
// (low_resolution_matcher.cc:27-34; the same loop as rtcsm_3d.cc:101-104).  Uses ctx->bounds,
// ctx->rescore and ctx->misc as scratch; synchronises the stream.
namespace dliom {
int sequential_probability_sums(dliom_ctx* ctx, const dliom_cloud& cloud, const dliom_grid* grid, const float* poses7,
                                int k, float* sums) {
  if (k <= 0 || k > 65535 || cloud.n <= 0) return DLIOM_ERR_INVALID_ARGUMENT;
  const size_t K = static_cast<size_t>(k);
  const size_t rot_bytes = (K * 16 + 255) & ~static_cast<size_t>(255);
  const size_t trans_bytes = (K * 12 + 255) & ~static_cast<size_t>(255);
  const size_t list_bytes = (K * 4 + 255) & ~static_cast<size_t>(255);
  DLIOM_TRY(ctx->bounds.reserve(rot_bytes + trans_bytes + list_bytes));
  std::vector<char> host(rot_bytes + trans_bytes + list_bytes, 0);
  float* hr = reinterpret_cast<float*>(host.data());
  float* ht = reinterpret_cast<float*>(host.data() + rot_bytes);
  unsigned* hl = reinterpret_cast<unsigned*>(host.data() + rot_bytes + trans_bytes);
  for (size_t i = 0; i < K; ++i) {
    const float* p = poses7 + 7 * i;
    hr[4 * i] = p[3];
    hr[4 * i + 1] = p[4];
    hr[4 * i + 2] = p[5];
    hr[4 * i + 3] = p[6];
    ht[3 * i] = p[0];
    ht[3 * i + 1] = p[1];
    ht[3 * i + 2] = p[2];
    hl[i] = static_cast<unsigned>(i * K + i);  // the kernels decode c -> (translation c / R, rotation c % R), R = k
  }
  char* base = static_cast<char*>(ctx->bounds.p);
  DLIOM_HIP_TRY(hipMemcpyAsync(base, host.data(), host.size(), hipMemcpyHostToDevice, ctx->stream));
  const float4* d_rot = reinterpret_cast<const float4*>(base);
  const float* d_trans = reinterpret_cast<const float*>(base + rot_bytes);
  const unsigned* d_list = reinterpret_cast<const unsigned*>(base + rot_bytes + trans_bytes);
  DLIOM_TRY(ctx->rescore.reserve(list_bytes));
  float* d_ksums = ctx->rescore.as<float>();
  DLIOM_TRY(launch_sequential_sums(ctx, rescore_method_default(), grid->view(), cloud, d_rot, k, d_trans, d_list, nullptr,
                                   static_cast<unsigned>(k), &ctx->misc, d_ksums));
  if (K <= 1024) {  // a few sums (the loop-closure matcher asks for one at a time): packed by a kernel, polled
    const GatherJob job{d_ksums, static_cast<unsigned>(K)};
    float* h = reinterpret_cast<float*>(static_cast<char*>(ctx->pinned) + ctx->pinned_bytes - 8192);
    DLIOM_TRY(gather_and_wait(ctx, &job, 1, h));  // also keeps `host` alive long enough: the upload is in front of it
    std::memcpy(sums, h, K * 4);
    return DLIOM_OK;
  }
  DLIOM_HIP_TRY(hipMemcpyAsync(sums, d_ksums, K * 4, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));  // also keeps `host` alive long enough
  return DLIOM_OK;
}
}  // namespace dliom

extern "C" {

int dliom_rtcsm3d_window(const dliom_rtcsm_options* o, float resolution, const float* points_xyz,
                         int64_t n, dliom_rtcsm_window* w) {
  if (o == nullptr || w == nullptr || n < 0 || (n > 0 && points_xyz == nullptr))
    return DLIOM_ERR_INVALID_ARGUMENT;
  compute_window(*o, resolution, cloud_max_norm(points_xyz, n), w);
  return DLIOM_OK;
}

int dliom_rtcsm3d_match_cloud(dliom_ctx* ctx, const dliom_rtcsm_options* o, const double init7[7],
                              const dliom_cloud* cloud, const dliom_grid* grid, double out7[7],
                              float* score) {
  if (ctx == nullptr || o == nullptr || init7 == nullptr || cloud == nullptr || grid == nullptr ||
      out7 == nullptr || score == nullptr)
    return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  return match_impl(ctx, o, init7, *cloud, grid, out7, score);
}

int dliom_rtcsm3d_match(dliom_ctx* ctx, const dliom_rtcsm_options* o, const double init7[7],
                        const float* points_xyz, int64_t n, const dliom_grid* grid, double out7[7],
                        float* score) {
  if (ctx == nullptr || o == nullptr || init7 == nullptr || grid == nullptr || out7 == nullptr ||
      score == nullptr || n < 0 || (n > 0 && points_xyz == nullptr))
    return DLIOM_ERR_INVALID_ARGUMENT;
  if (n == 0) return DLIOM_ERR_EMPTY_CLOUD;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  DLIOM_TRY(ctx->points.reserve(staged_cloud_bytes(n)));
  dliom_cloud cloud;
  DLIOM_TRY(stage_cloud(ctx, points_xyz, n, &cloud));
  return match_impl(ctx, o, init7, cloud, grid, out7, score);
}

int dliom_rtcsm3d_shard_begin(dliom_ctx* ctx, const dliom_rtcsm_options* o, const double init7[7],
                              const dliom_cloud* cloud, const dliom_grid* grid, int shard, int num_shards,
                              uint32_t* local_best_lower_bound_bits) {
  if (ctx == nullptr || o == nullptr || init7 == nullptr || cloud == nullptr || grid == nullptr ||
      local_best_lower_bound_bits == nullptr)
    return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  return match_begin(ctx, o, init7, *cloud, grid, shard, num_shards, local_best_lower_bound_bits);
}

int dliom_rtcsm3d_shard_finish(dliom_ctx* ctx, uint32_t global_best_lower_bound_bits,
                               uint64_t* local_best_packed) {
  if (ctx == nullptr || local_best_packed == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  return match_finish(ctx, &global_best_lower_bound_bits, local_best_packed);
}

int dliom_rtcsm3d_shard_decode(dliom_ctx* ctx, uint64_t global_best_packed, double pose_estimate[7],
                               float* score) {
  if (ctx == nullptr || pose_estimate == nullptr || score == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  return match_decode(ctx, global_best_packed, pose_estimate, score);
}

// ---- config 4: the search window sharded over the ranks of a node, ONE 8-byte collective per match --------------
// Every rank scores its own contiguous range of candidate rotations and finds ITS winner exactly (own bounds, own
// exact rescoring).  The global winner -- the reference's first strictly greater score in generation order
// (rtcsm_3d.cc:46-51) -- is the maximum of the ranks' packed words (score_bits << 32 | ~index): positive floats order
// like their bit patterns and the complemented index lets the lower index win ties.  One MAX all-reduce of one
// uint64; its 8 bytes are latency, not bandwidth.  The three-phase calls above stay for callers that
// want the global lower bound exchanged first (less rescoring per rank, two collectives).
// reserved word of a rank that failed: above every real packed word (score bits of a finite positive float are at most
// 0x7F7FFFFF) under unsigned AND signed 64-bit MAX -- torch.distributed has no uint64, callers reduce int64
static constexpr uint64_t kShardFailed = 0x7FFFFFFFFFFFFFFFull;
int dliom_rtcsm3d_match_sharded(dliom_ctx* ctx, const dliom_rtcsm_options* o, const double init7[7], const dliom_cloud* cloud,
                                const dliom_grid* grid, int shard, int num_shards, dliom_allreduce_max_u64 exchange,
                                void* user, double out7[7], float* score) {
  if (num_shards > 1 && exchange == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  // Every rank ALWAYS takes part in the exchange: a rank that failed locally (bad arguments, a HIP error, an
  // allocation, a refusal, an empty cloud) contributes the reserved word kShardFailed -- the maximum, so every rank
  // sees it -- and returns its own status afterwards; the others return DLIOM_ERR_PEER_FAILED.  Returning before the
  // collective would leave the peers blocked in it (on RCCL: a device-side hang without an error).
  int local = ctx == nullptr || o == nullptr || init7 == nullptr || cloud == nullptr || grid == nullptr || out7 == nullptr ||
                      score == nullptr
                  ? DLIOM_ERR_INVALID_ARGUMENT
                  : DLIOM_OK;
  if (local == DLIOM_OK && hipSetDevice(ctx->device) != hipSuccess) local = DLIOM_ERR_HIP;
  uint64_t packed = 0;
  if (local == DLIOM_OK) local = match_begin(ctx, o, init7, *cloud, grid, shard, num_shards, nullptr);
  if (local == DLIOM_OK) local = match_finish(ctx, nullptr, &packed);
  if (local != DLIOM_OK) packed = kShardFailed;
  if (num_shards > 1 && exchange(&packed, user) != 0) return local != DLIOM_OK ? local : DLIOM_ERR_HIP;
  if (local != DLIOM_OK) return local;
  if (packed == kShardFailed) return DLIOM_ERR_PEER_FAILED;
  return match_decode(ctx, packed, out7, score);
}

}  // extern "C"

namespace {
struct RcclApi {
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  bool ok = false;
};
// librccl is looked up when the first sharded match runs: libdliom.so itself loads (and every other entry point
// works) on machines without RCCL.
const RcclApi& rccl_api() {
  static const RcclApi api = [] {
    RcclApi a;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (h == nullptr) return a;
    a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(h, "ncclCommCount"));
    a.CommUserRank = reinterpret_cast<decltype(a.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.ok = a.CommCount != nullptr && a.CommUserRank != nullptr && a.AllReduce != nullptr;
    return a;
  }();
  return api;
}
struct RcclExchange {
  dliom_ctx* ctx;
  ncclComm_t comm;
};
int rccl_exchange(uint64_t* value, void* user) {
  RcclExchange* x = static_cast<RcclExchange*>(user);
  dliom_ctx* ctx = x->ctx;
  if (ctx->misc.reserve(256) != DLIOM_OK) return 1;
  uint64_t* d = ctx->misc.as<uint64_t>();
  uint64_t* h = reinterpret_cast<uint64_t*>(static_cast<char*>(ctx->pinned) + ctx->pinned_bytes - 4096 + 3072);
  *h = *value;
  const int span = ctx->begin_span(DLIOM_KERNEL_ALLREDUCE);  // (profiling on: HIP events around the collective)
  bool ok = hipMemcpyAsync(d, h, 8, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
            rccl_api().AllReduce(d, d, 1, ncclUint64, ncclMax, x->comm, ctx->stream) == ncclSuccess &&
            hipMemcpyAsync(h, d, 8, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
  ctx->end_span(span);
  if (!ok) return 1;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return 1;
  *value = *h;
  return 0;
}
}  // namespace

extern "C" {

int dliom_rtcsm3d_match_sharded_rccl(dliom_ctx* ctx, const dliom_rtcsm_options* o, const double init7[7],
                                     const dliom_cloud* cloud, const dliom_grid* grid, void* nccl_comm, double out7[7],
                                     float* score) {
  if (ctx == nullptr || nccl_comm == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  const RcclApi& api = rccl_api();
  if (!api.ok) {
    set_last_error("dlopen(librccl.so.1): RCCL not available", hipErrorSharedObjectInitFailed, __FILE__, __LINE__);
    return DLIOM_ERR_HIP;
  }
  int rank = 0, size = 1;
  ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
  if (api.CommCount(comm, &size) != ncclSuccess || api.CommUserRank(comm, &rank) != ncclSuccess) return DLIOM_ERR_INVALID_ARGUMENT;
  RcclExchange x{ctx, comm};
  // a communicator of one rank still goes through ncclAllReduce (size 1 is a copy): same code path as on 8 GPUs
  if (size == 1) {
    if (ctx == nullptr || o == nullptr || init7 == nullptr || cloud == nullptr || grid == nullptr || out7 == nullptr ||
        score == nullptr)
      return DLIOM_ERR_INVALID_ARGUMENT;
    int local = hipSetDevice(ctx->device) == hipSuccess ? DLIOM_OK : DLIOM_ERR_HIP;
    uint64_t packed = 0;
    if (local == DLIOM_OK) local = match_begin(ctx, o, init7, *cloud, grid, 0, 1, nullptr);
    if (local == DLIOM_OK) local = match_finish(ctx, nullptr, &packed);
    if (local != DLIOM_OK) packed = kShardFailed;
    if (rccl_exchange(&packed, &x) != 0) return local != DLIOM_OK ? local : DLIOM_ERR_HIP;
    if (local != DLIOM_OK) return local;
    return match_decode(ctx, packed, out7, score);
  }
  return dliom_rtcsm3d_match_sharded(ctx, o, init7, cloud, grid, rank, size, rccl_exchange, &x, out7, score);
}

int dliom_rtcsm3d_box_error(dliom_ctx* ctx, uint32_t* flags) {
  if (ctx == nullptr || flags == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  *flags = 0;
  if (ctx->box_error.p == nullptr) return DLIOM_OK;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  DLIOM_HIP_TRY(hipMemcpyAsync(flags, ctx->box_error.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return DLIOM_OK;
}

#ifdef DLIOM_EXPERIMENTS
// experiments builds only (not in dliom.h): the box kernel's work counters (Params::debug & 128), read and cleared
extern "C" int dliom_exp_box_stats(dliom_ctx* ctx, uint32_t out[8]) {
  if (ctx == nullptr || out == nullptr || ctx->box_error.p == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipMemcpyAsync(out, static_cast<char*>(ctx->box_error.p) + 32, 32, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipMemsetAsync(static_cast<char*>(ctx->box_error.p) + 32, 0, 32, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return DLIOM_OK;
}
#endif

#ifdef DLIOM_EXPERIMENTS
// per-workgroup stamps of the last box-kernel launch: (start, end) in 10 ns ticks, tickets processed, XCC id
extern "C" int dliom_exp_box_stamps(dliom_ctx* ctx, uint64_t* out, int workgroups) {
  if (ctx == nullptr || out == nullptr || ctx->box_error.p == nullptr || workgroups > 4096) return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipMemcpyAsync(out, static_cast<char*>(ctx->box_error.p) + 256, static_cast<size_t>(workgroups) * 32,
                               hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return DLIOM_OK;
}
#endif

int dliom_rtcsm3d_last_stats(const dliom_ctx* ctx, dliom_rtcsm_stats* stats) {
  if (ctx == nullptr || stats == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  *stats = ctx->last_rtcsm;
  return DLIOM_OK;
}

int dliom_rtcsm3d_score_volume(dliom_ctx* ctx, const dliom_rtcsm_options* o, const double init7[7],
                               const float* points_xyz, int64_t n, const dliom_grid* grid,
                               uint64_t* sums, int64_t capacity, int64_t* num_candidates) {
  if (ctx == nullptr || o == nullptr || init7 == nullptr || grid == nullptr ||
      num_candidates == nullptr || n <= 0 || points_xyz == nullptr)
    return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  DLIOM_TRY(ctx->points.reserve(staged_cloud_bytes(n)));
  dliom_cloud cloud;
  DLIOM_TRY(stage_cloud(ctx, points_xyz, n, &cloud));
  Candidates c;
  generate_candidates(*o, grid->resolution, cloud.max_norm, init7, &c);
  *num_candidates = c.w.num_candidates;
  if (sums == nullptr) return DLIOM_OK;
  if (capacity < c.w.num_candidates) return DLIOM_ERR_CAPACITY;
  DeviceCandidates d;
  unsigned long long* d_sums = nullptr;
  int64_t pad_processed = 0;
  DLIOM_TRY(run_score_volume(ctx, cloud, grid, c, 0, static_cast<int>(c.w.num_rotations), &d, &d_sums,
                             &pad_processed));
  unsigned err1 = 0u;
  if (ctx->last_score_used_box)
    DLIOM_HIP_TRY(hipMemcpyAsync(&err1, static_cast<char*>(ctx->box_error.p) + 4, 4, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipMemcpyAsync(sums, d_sums, static_cast<size_t>(c.w.num_candidates) * 8,
                               hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  {
    bool overflow = false;
    DLIOM_TRY(box_overflowed(ctx, err1, &overflow));
    if (overflow) {
      ctx->force_dense_score = true;
      const int s = run_score_volume(ctx, cloud, grid, c, 0, static_cast<int>(c.w.num_rotations), &d, &d_sums, &pad_processed);
      ctx->force_dense_score = false;
      DLIOM_TRY(s);
      DLIOM_HIP_TRY(hipMemcpyAsync(sums, d_sums, static_cast<size_t>(c.w.num_candidates) * 8, hipMemcpyDeviceToHost, ctx->stream));
      DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
  }
  const uint64_t n_pad = static_cast<uint64_t>(pad_processed);
  for (int64_t i = 0; i < c.w.num_candidates; ++i) sums[i] -= n_pad;
  return DLIOM_OK;
}

int dliom_rtcsm3d_sequential_sums(dliom_ctx* ctx, const dliom_rtcsm_options* o, const double init7[7],
                                  const float* points_xyz, int64_t n, const dliom_grid* grid,
                                  const int64_t* candidate_indices, int64_t k, int method, float* sums) {
  if (ctx == nullptr || o == nullptr || init7 == nullptr || grid == nullptr || candidate_indices == nullptr ||
      sums == nullptr || n <= 0 || points_xyz == nullptr || k <= 0)
    return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  DLIOM_TRY(ctx->points.reserve(staged_cloud_bytes(n)));
  dliom_cloud cloud;
  DLIOM_TRY(stage_cloud(ctx, points_xyz, n, &cloud));
  Candidates c;
  generate_candidates(*o, grid->resolution, cloud.max_norm, init7, &c);
  DeviceCandidates d;
  DLIOM_TRY(upload_candidates(ctx, c, &d));
  std::vector<unsigned> list(static_cast<size_t>(k));
  for (int64_t i = 0; i < k; ++i) {
    if (candidate_indices[i] < 0 || candidate_indices[i] >= c.w.num_candidates) return DLIOM_ERR_INVALID_ARGUMENT;
    list[i] = static_cast<unsigned>(candidate_indices[i]);
  }
  const size_t lbytes = (static_cast<size_t>(k) * 4 + 255) & ~static_cast<size_t>(255);
  DLIOM_TRY(ctx->rescore.reserve(2 * lbytes));
  unsigned* d_list = ctx->rescore.as<unsigned>();
  float* d_ksums = reinterpret_cast<float*>(static_cast<char*>(ctx->rescore.p) + lbytes);
  DLIOM_HIP_TRY(hipMemcpyAsync(d_list, list.data(), static_cast<size_t>(k) * 4, hipMemcpyHostToDevice, ctx->stream));
  const int R = static_cast<int>(c.w.num_rotations);
  if (method < 0 || method > 2 || k > 65535) return DLIOM_ERR_INVALID_ARGUMENT;
  if (method != 0 && ((static_cast<size_t>(n) + 7) & ~static_cast<size_t>(7)) * 2 > 128 * 1024) return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_TRY(launch_sequential_sums(ctx, method, grid->view(), cloud, d.rot, R, d.trans, d_list, nullptr,
                                   static_cast<unsigned>(k), &ctx->misc, d_ksums));
  DLIOM_HIP_TRY(hipMemcpyAsync(sums, d_ksums, static_cast<size_t>(k) * 4, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return DLIOM_OK;
}

int dliom_probe_transform_cell_indices(dliom_ctx* ctx, const float pose[7], const float* points_xyz,
                                       int64_t n, float resolution, int32_t* cell_xyz) {
  if (ctx == nullptr || pose == nullptr || n < 0 || (n > 0 && (points_xyz == nullptr || cell_xyz == nullptr)))
    return DLIOM_ERR_INVALID_ARGUMENT;
  if (n == 0) return DLIOM_OK;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  DLIOM_TRY(ctx->points.reserve(staged_cloud_bytes(n)));
  dliom_cloud cloud;
  DLIOM_TRY(stage_cloud(ctx, points_xyz, n, &cloud));
  DLIOM_TRY(ctx->misc.reserve(static_cast<size_t>(n) * 12));
  int* d_out = ctx->misc.as<int>();
  const Quat4 q{pose[3], pose[4], pose[5], pose[6]};
  hipLaunchKernelGGL(probe_cells_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                     ctx->stream, q, pose[0], pose[1], pose[2], cloud.d_x, cloud.d_y, cloud.d_z,
                     static_cast<int>(n), resolution, d_out);
  DLIOM_HIP_TRY(hipGetLastError());
  DLIOM_HIP_TRY(hipMemcpyAsync(cell_xyz, d_out, static_cast<size_t>(n) * 12, hipMemcpyDeviceToHost,
                               ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return DLIOM_OK;
}

}  // extern "C"
