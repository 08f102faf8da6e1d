// Device functions shared by the gfx950 kernels.  Everything that decides a
// voxel index lives here so that every kernel (score volume, rescoring,
// insertion, probes) runs the same instructions.
//
// Built with -ffp-contract=off and IEEE float division (hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt), see d-liom_amd/Makefile: the
// reference's x86-64 build has no FMA and uses true division
// (mapping/3d/hybrid_grid.h:430-435, cmake/functions.cmake:91-92).
#ifndef DLIOM_CSRC_DEVICE_COMMON_H_
#define DLIOM_CSRC_DEVICE_COMMON_H_

#include <hip/hip_runtime.h>

#include "internal.h"

namespace dliom {

// std::lround semantics (common/port.h:41): round half away from zero.
// x - trunc(x) is exact in binary floating point.
__device__ __forceinline__ int lround_away(float x) {
  const float t = truncf(x);
  const float d = x - t;
  int k = static_cast<int>(t);
  k += (d >= 0.5f) ? 1 : 0;
  k -= (d <= -0.5f) ? 1 : 0;
  return k;
}

// HybridGridBase::GetCellIndex for one coordinate.
__device__ __forceinline__ int cell_of(float p, float resolution) {
  return lround_away(p / resolution);
}

struct Quat4 {
  float w, x, y, z;
};

// Eigen QuaternionBase::_transformVector, scalar order kept:
//   uv = u x v; uv += uv; return (v + w*uv) + u x uv
__device__ __forceinline__ void rotate_point(const Quat4 q, float vx, float vy, float vz,
                                             float& ox, float& oy, float& oz) {
  float uvx = q.y * vz - q.z * vy;
  float uvy = q.z * vx - q.x * vz;
  float uvz = q.x * vy - q.y * vx;
  uvx = uvx + uvx;
  uvy = uvy + uvy;
  uvz = uvz + uvz;
  const float cx = q.y * uvz - q.z * uvy;
  const float cy = q.z * uvx - q.x * uvz;
  const float cz = q.x * uvy - q.y * uvx;
  ox = (vx + q.w * uvx) + cx;
  oy = (vy + q.w * uvy) + cy;
  oz = (vz + q.w * uvz) + cz;
}

// DynamicGrid::value (hybrid_grid.h:263-281): shift, unsigned bounds check,
// leaf lookup, cell lookup.  Missing leaves map to slot 0 (all zeros), so the
// only branch-free special case is "outside the current extent".
__device__ __forceinline__ unsigned grid_value(const GridView& g, int ix, int iy, int iz) {
  const unsigned sx = static_cast<unsigned>(ix + g.half);
  const unsigned sy = static_cast<unsigned>(iy + g.half);
  const unsigned sz = static_cast<unsigned>(iz + g.half);
  const bool inside = (sx < g.grid_size) & (sy < g.grid_size) & (sz < g.grid_size);
  const size_t L = static_cast<size_t>(g.leaves_per_axis);
  const size_t tidx = inside ? ((sz >> 3) * L + (sy >> 3)) * L + (sx >> 3) : 0u;  // 2^33 entries at bits = 8
  unsigned slot = g.table[tidx];
  slot = inside ? slot : 0u;
  const unsigned cell = ((sz & 7u) << 6) | ((sy & 7u) << 3) | (sx & 7u);
  return g.pool[static_cast<size_t>(slot) * 512u + cell];
}

// Fast voxel index for the score-volume kernel: k = rint(p * (1/res)) whenever that provably
// equals lround(p / res), else *near is set and the caller recomputes with cell_of().
//   y = fl(p * fl(1/res)) differs from the real quotient q by <= |q| 2^-23; the reference's
//   decision lround(fl(q)) can flip only when q is within |q| 2^-24 of a half-integer.  So if y is
//   farther than 2.5e-7 |y| (> 1.5 * 2^-23 |y|) from every half-integer, y and q round to the same
//   integer -- and then round-half-even (v_rndne) and round-half-away agree as well.
__device__ __forceinline__ int cell_fast(float p, float inv_resolution, bool* near) {
  const float y = p * inv_resolution;
  const float fr = __builtin_amdgcn_fractf(y);
  *near |= fabsf(fr - 0.5f) <= fabsf(y) * 2.5e-7f;
  return __float2int_rn(y);
}

// grid_value() with shift/or index math and 32-bit byte offsets (pool <= 4 GiB, table <= 4 GiB: bits <= 7; callers
// send bits = 8 grids to grid_value()).
__device__ __forceinline__ unsigned grid_value_fast(const GridView& g, int ix, int iy, int iz) {
  const unsigned sx = static_cast<unsigned>(ix + g.half);
  const unsigned sy = static_cast<unsigned>(iy + g.half);
  const unsigned sz = static_cast<unsigned>(iz + g.half);
  const bool inside = (sx | sy | sz) < g.grid_size;  // grid_size is a power of two
  const unsigned lb = static_cast<unsigned>(g.log2_leaves);
  unsigned tidx = (((sz >> 3) << lb | (sy >> 3)) << lb) | (sx >> 3);
  tidx = inside ? tidx : 0u;
  unsigned slot = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(g.table) + (tidx << 2));
  slot = inside ? slot : 0u;
  const unsigned cell = ((sz & 7u) << 6) | ((sy & 7u) << 3) | (sx & 7u);
  const unsigned off = (slot << 10) | (cell << 1);
  return *reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(g.pool) + off);
}

// Sum over the 64 lanes of a wavefront; the total lands in lane 63.
// quad swaps, half-row mirror, row mirror, then the two row broadcasts --
// six DPP adds, no LDS traffic.
__device__ __forceinline__ unsigned wave_sum_lane63(unsigned v) {
  v += static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0xB1, 0xf, 0xf, false));
  v += static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x4E, 0xf, 0xf, false));
  v += static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x141, 0xf, 0xf, false));
  v += static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x140, 0xf, 0xf, false));
  v += static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x142, 0xa, 0xf, false));
  v += static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x143, 0xc, 0xf, false));
  return v;
}

// ---- small copies and fills that ride along in another kernel's launch (round 5) ------------------------------------
// A Match used to put six operations on the stream before its score kernel could start -- candidate tables H2D, fill of
// the score volume, box tables H2D, counter memset, extent pre-pass, score kernel -- each a packet of its own with ~5 us
// of launch / dependency latency while the device idles.  The tables are staged in device-visible pinned host memory
// anyway: a few surplus workgroups of the FIRST kernel of the chain read them from there (PCIe, ~40 KB) and do the fills.
// Everything is 4-byte granular; dst/src 16-byte aligned.
struct PrepJob {
  uint4* dst;
  const uint4* src;     // device-visible (pinned host or device) source; null: fill with `value`
  unsigned vec;         // whole 16-byte units
  unsigned tail_words;  // 0..3 words behind them
  unsigned value;
};
constexpr int kMaxPrepJobs = 6;
struct PrepArgs {
  PrepJob job[kMaxPrepJobs];
  int n;
};
// executed by `nblocks` workgroups of `nthreads` threads (block = this workgroup's index among them)
__device__ __forceinline__ void prep_block(const PrepArgs& a, unsigned block, unsigned nblocks, unsigned tid, unsigned nthreads) {
#pragma unroll
  for (int j = 0; j < kMaxPrepJobs; ++j) {  // static indices: the argument stays in SGPRs (no scratch copy)
    if (j >= a.n) break;
    const PrepJob& jb = a.job[j];
    const uint4 v4 = make_uint4(jb.value, jb.value, jb.value, jb.value);
    for (unsigned i = block * nthreads + tid; i < jb.vec; i += nblocks * nthreads) jb.dst[i] = jb.src != nullptr ? jb.src[i] : v4;
    if (block == 0 && tid < jb.tail_words) {
      unsigned* dt = reinterpret_cast<unsigned*>(jb.dst + jb.vec);
      dt[tid] = jb.src != nullptr ? reinterpret_cast<const unsigned*>(jb.src + jb.vec)[tid] : jb.value;
    }
  }
}

}  // namespace dliom

#endif  // DLIOM_CSRC_DEVICE_COMMON_H_
