// RTCSM3D score volume, LDS-box kernel (round 2; work distribution, extent pre-pass and staging arithmetic round 3).
// Included by rtcsm3d.hip only.
//
// Same exact integer sums as rtcsm_score_dense_kernel (sum_i max(v_i & 0x7fff, 1) per candidate,
// real_time_correlative_scan_matcher_3d.cc:97-104), restructured around what the gfx950 pipes
// can do (profiles/r2_ubench_instruction_rates.json, r2_score_pipe_experiment.json):
//   * a non-coalesced global gather costs the texture addresser 16-29 cycles per wave -- a u16 gather
//     out of LDS 2-6.  A WORKGROUP stages the sub-box of the dense mirror that its (up to 256) candidate
//     rotations can reach from a chunk of Morton-adjacent points ("LDS-staged tiles"); each wave walks the
//     chunk with its own 64 rotations and gathers with ds_read_u16;
//   * only v_add/sub/mul/fma_f32, v_add/sub_u32, v_and/or/xor, v_lshrrev run at 2 cycles per wave;
//     v_cvt/v_fract/v_floor/min/max/cmp, every VOP3 integer op, every SDWA op and any VALU op with an
//     SGPR operand take 4.  The per-lookup index math is: one v_add_f32 per axis (the translation is
//     added to a pre-scaled coordinate kept in [128, 256), where the float's own bit layout IS
//     [cell : 7 bits | fraction : 16 bits]) and three v_mad_u32_u16 that read the cell index straight
//     out of the high half: 7 VALU per lookup, no test, no branch, no scalar work;
//   * accumulators live in registers (the 27 translations of a pass are unrolled) and are flushed once per
//     wave (~7 M 64-bit atomics per launch instead of 74 M), and every kFlushPoints points so that clouds of any
//     size keep 32-bit accumulators;
//   * per iteration of kHotP points the next points are already in registers (fetched during the previous
//     iteration's 27 steps), the 3 kHotP band-bitmap words are fetched together and the list append is branch-free;
//     the values gathered in one step are accumulated after the next step's address arithmetic (kPipe, kLateAcc);
//   * (round 3) work units are (pass, rotation block) with single-chunk tickets, handed out most expensive first, and
//     workgroups move on to other units when theirs is empty; the per-point part of the boxes' bounding boxes comes from
//     a pre-pass (rtcsm_box_extent_kernel); the staging loop's index arithmetic avoids quarter-rate 32-bit multiplies.
// What bounds it and what was tried: DESIGN.md 3.1.  The timing experiments quoted there (DLIOM_BOX_EXP,
// Params::debug) produce wrong sums by design: they exist only in builds made with -DDLIOM_EXPERIMENTS
// (`make experiments` -> ab/libdliom_exp.so); the library that ships contains neither them nor any switch that
// reads the environment.
// Exactness without a per-lookup test.  A fast lookup can differ from the reference's cell only when its
// scaled coordinate w lies within a rounding band of a cell boundary (budget below).  Whether ANY of the
// 27 translations of a pass puts a rotated point into a band depends, per axis, only on the fraction of
// its scaled coordinate: the host marks those fractions in a bitmap (8192 buckets per axis); the kernel
// tests the three fractions once per (rotation, point) -- 1/27 of the lookups -- and lists the ~2 % that
// hit.  Listed (rotation, point) pairs are re-examined 64 at a time, one per lane (level 1: which
// translations are really inside a band), and those lookups are resolved with the exact IEEE division
// path, again one per lane (level 2); where the exact cell differs from the fast one the difference of
// the two grid values is added to the score volume with an atomic (rare).
//
// Error budget of the fast index (u = 2^-16 cells, the ulp of a float in [128, 256)):
//   reference  c = fl(r + t), Q = fl(c / res), i = lround(Q)            (hybrid_grid.h:430-435)
//   here       w = fl( fl(r * inv + Kb) + tau ),  l = floor(w) - 128    (inv = fl(1 / res))
//   with  Kb + tau == t / res + 128.5 + s u - i_lo  up to 2^-24 |tau|  (host, double)
//   |w - (W + s u)| <= u  (two roundings of results < 256)  +  |r / res| 2^-24  (inv)  +  |tau| 2^-24
//   the reference's i can differ from floor(q + 1/2), q = (r + t) / res real, only when q + 1/2 is
//   within 2 |q| 2^-24 of an integer (its two roundings; exact ties included)
//   => E = u + (2 qmax + rmax + taumax) 2^-24;  B = ceil(E / u + 1/4), shift s >= B, and a lookup is
//   resolved by the fast path only if frac16(w) > s + B -- then floor(w) is the reference's cell.
#ifndef DLIOM_CSRC_SCORE_BOX_H_
#define DLIOM_CSRC_SCORE_BOX_H_

#include "device_common.h"

namespace dliom {
namespace box {

constexpr int kTC = 27;        // translations per pass of the narrow kernels (register accumulators)
#if !defined(DLIOM_EXPERIMENTS) || !defined(DLIOM_BOX_TC_WIDE)
#undef DLIOM_BOX_TC_WIDE
#define DLIOM_BOX_TC_WIDE 49
#endif
// ... of the wide kernel (round 6): one staged box and one rotation per point serve 49 translations.  49 = 7 x 7: with a
// linear window of three cells (BASELINE config 5) a pass is one z-plane of the 7^3 translations -- no padding (54 fills
// the last pass of 343 up to 378: measured 180 ms against 159 on config 5) and a reach of one cell along z
constexpr int kTCWide = DLIOM_BOX_TC_WIDE;
#if !defined(DLIOM_EXPERIMENTS) || !defined(DLIOM_BOX_MAX_WAVES)
#undef DLIOM_BOX_MAX_WAVES
#define DLIOM_BOX_MAX_WAVES 4
#endif
constexpr int kWaves = DLIOM_BOX_MAX_WAVES;  // waves per workgroup at most: consecutive rotation groups, the same point chunks
constexpr int kMaxDim = 120;   // box cells per axis (scaled coordinates stay below 256)
constexpr int kMaxChunk = 64;  // points per chunk at most (one per lane in the bounding-box pass)
constexpr int kBuckets = 8192; // fraction buckets per axis of the band bitmap
constexpr int kBitmapWords = 3 * kBuckets / 32;
constexpr int kL1Cap = 320;    // level-1 list: (rotation lane, point, chunk record) per entry
constexpr int kL2Cap = 128;    // level-2 list: (level-1 slot, translation)
#ifndef DLIOM_BOX_HOT_P
#define DLIOM_BOX_HOT_P 4
#endif
constexpr int kHotP = DLIOM_BOX_HOT_P;  // points per hot-loop iteration (8 or 4)
#if !defined(DLIOM_EXPERIMENTS) || !defined(DLIOM_BOX_EXP)
#undef DLIOM_BOX_EXP
#define DLIOM_BOX_EXP 0  // 1, 2, 3, 4: timing experiments of the hot loop (wrong sums; -DDLIOM_EXPERIMENTS builds only)
#endif
#ifdef DLIOM_EXPERIMENTS
#define DLIOM_BOX_DBG(p, bit) (((p).debug & (bit)) != 0)
// work counters (Params::debug & 128): error[8 + k] += v by the lanes for which cond holds
#define DLIOM_BOX_STAT(p, cond, k, v)                                                                \
  do {                                                                                                \
    if (((p).debug & 128) && (cond)) atomicAdd((p).error + 8 + (k), static_cast<unsigned>(v));        \
  } while (0)
#else
#define DLIOM_BOX_DBG(p, bit) false
#define DLIOM_BOX_STAT(p, cond, k, v) do { } while (0)
#endif
#ifndef DLIOM_BOX_PIPE
#define DLIOM_BOX_PIPE 1
#endif
#ifndef DLIOM_BOX_LATE_ACC
#define DLIOM_BOX_LATE_ACC 1
#endif
constexpr int kPipe = DLIOM_BOX_PIPE;            // steps between a gather and the accumulation of its value
constexpr bool kLateAcc = DLIOM_BOX_LATE_ACC != 0;  // accumulate after the step's address arithmetic (else: anywhere)
constexpr int kRecords = 8;    // ring of chunk records (lo[3], first point) the level-1 entries refer to
constexpr int kListTrash = kL1Cap + kL2Cap + 4 * kRecords;  // a word nobody reads: where unlisted lanes "append"
constexpr int kListWords = kListTrash + 4;

struct Pass {      // one per translation pass
  int gi[3];       // floor(G), G = t_c / res + 128.5 + s u  (t_c: centre translation of the pass)
  float f[3];      // fraction of G rounded to 14 bits (Kb = float(gi - i_lo) + f is exact)
  float uc[3];     // float(t_c / res + 0.5): coordinate constant of the bounding-box pass
  float reach[3];  // max_j |tau_j| per axis + slack (cells)
  int tc;          // translations in this pass (<= kTC)
  int j0;          // first translation of the pass
  int pad[2];
};

struct Group {     // spread of a workgroup's rotations around its centre lane (host, double):
  int c_lane;      //   q_lane = q_centre * d_lane,  d_lane = exp(delta_lane)
  float dc[3];     // centre of the bounding box of the delta_lane (rotation vectors, point frame)
  float hd[3];     // its half extents
  float theta2;    // 0.51 max |delta|^2: bound of |R(delta) p - p - delta x p| / |p|
};

struct Params {
  const float4* tau;       // passes * kTC entries (x, y, z, 0): translation minus G, in cells
  const Pass* pass;
  const unsigned* bitmap;  // passes * kBitmapWords: fractions that reach a rounding band, per axis
  const float* trans;      // T x 3: the reference's float translations (exact path)
  const float4* rot;       // candidate rotations (w, x, y, z)
  const Group* group;      // one per workgroup's rotations (nw * 64 from r_first on)
  const float* ext;        // [rot_blocks][6][ext_stride]: per point, lower / upper end per axis of its lookups under the
                           // block's rotations, in cells, before the pass's translation (rtcsm_box_extent_kernel)
  int ext_stride;
  unsigned long long* sums;
  const unsigned* order;   // ticket -> chunk (most expensive chunks first, core.hip chunk_order_kernel) or null: identity
  unsigned* counters;      // one chunk dispenser per unit = (pass, rotation block), zeroed by the host before the launch
  unsigned* error;         // [0] sticky: an exact cell more than one cell from the fast one (cannot happen);
                           // [1] the same, cleared by the host when it reruns the match with the dense kernel
                           // (the level-1 list itself cannot overflow: the hot loop stops and drains it first)
  int R, r_first, r_last, T, passes;
  int n;                   // real points (Morton order)
  int chunk;               // points per chunk, multiple of 4
  int point_chunks, rot_groups, rot_blocks, nw, slots;  // nw waves per workgroup, rot_blocks = ceil(rot_groups / nw)
  int units;               // passes * rot_blocks
  unsigned thr;            // unresolved  <=>  (bits(w) & 0xffff) <= thr
  int cells;               // LDS box capacity per workgroup (cells)
  int split_points;        // 1: waves of a unit with fewer rotation groups than waves share the groups and split the points
  int debug;               // -DDLIOM_EXPERIMENTS builds only (wrong sums): 1 skip the lists, 2 skip staging, 4 skip the
                           // lookups, 8 no work at all, 16 unconditional flush atomics, 32 no flush, 64 no stealing,
                           // 128 work counters in error[8..15] (dliom_exp_box_stats), 256 per-workgroup stamps, 512 chunks in index
                           // order, 1024 next ticket drawn with the first box
};

typedef __attribute__((address_space(3))) const unsigned short lds_cu16;

__device__ __forceinline__ unsigned min_lo16(unsigned running, float w) {
  unsigned r;
  asm("v_min_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
      : "=v"(r)
      : "v"(running), "v"(w));
  return r;
}
// hi16(w) * m + a
__device__ __forceinline__ unsigned mad_hi16(float w, unsigned m, unsigned a) {
  unsigned r;
  asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(w), "v"(m), "v"(a));
  return r;
}
typedef float float2v __attribute__((ext_vector_type(2)));
// {a.x + t.x, a.y + t.x}, {a.x + t.y, a.y + t.y}: one VOP3P instruction for two lookups (the kernel is
// bound by instruction issue -- one per 4 cycles and SIMD whatever the type -- not by lane throughput)
__device__ __forceinline__ float2v pk_add_lo(float2v a, float2v t) {
  float2v r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(t));
  return r;
}
__device__ __forceinline__ float2v pk_add_hi(float2v a, float2v t) {
  float2v r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(t));
  return r;
}
__device__ __forceinline__ unsigned to_vgpr(unsigned s) {
  unsigned v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
  return v;
}
__device__ __forceinline__ float to_vgpr_f(float s) {
  float v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
  return v;
}
// Maximum over the 64 lanes (DPP butterfly: quad swaps, row half mirror, row mirror, the two row
// broadcasts; no LDS traffic), returned in every lane.
__device__ __forceinline__ float wave_max_f(float v) {
#define DLIOM_DPP_MAX(ctrl, row_mask)                                                                      \
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, row_mask, 0xf, false)))
  DLIOM_DPP_MAX(0xB1, 0xf);
  DLIOM_DPP_MAX(0x4E, 0xf);
  DLIOM_DPP_MAX(0x141, 0xf);
  DLIOM_DPP_MAX(0x140, 0xf);
  DLIOM_DPP_MAX(0x142, 0xa);
  DLIOM_DPP_MAX(0x143, 0xc);
#undef DLIOM_DPP_MAX
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_min_f(float v) { return -wave_max_f(-v); }

struct Geometry {   // wave-uniform description of the current box
  int lo[3];        // reference cell index of box cell (0, 0, 0)
  int dim[3];
  unsigned sx, sxy; // strides in cells (row, slab)
  float kb[3];
};

// value the matcher sums at reference cell (ix, iy, iz), straight from the mirror in HBM (outside: 1)
__device__ __forceinline__ unsigned mirror_value(const GridView& g, int ix, int iy, int iz) {
  const int S = g.dense_stride, B = g.dense_bricks;
  const int mx = ix + g.dense_off[0], my = iy + g.dense_off[1], mz = iz + g.dense_off[2];
  if (mx < 0 || mx >= S || my < 0 || my >= S || mz < 0 || mz >= S) return 1u;
  return g.dense[((static_cast<size_t>(mz >> 2) * B + (my >> 2)) * B + (mx >> 2)) * 64u +
                 static_cast<size_t>(((mz & 3) << 4) | ((my & 3) << 2) | (mx & 3))];
}

// ---- the lists ---------------------------------------------------------------------------------------
struct Lists {
  unsigned* l1;    // kL1Cap entries: lane | point-in-chunk << 6 | record << 12
  unsigned* l2;    // kL2Cap entries: level-1 slot | translation << 8
  int* rec;        // kRecords x (lo[3], first point of the chunk)
  int n1;          // wave-uniform counts
  int seq;         // chunk records written so far
};

// Level 2: one listed lookup per lane, resolved with the reference's own arithmetic.
__device__ __forceinline__ void resolve_l2(const GridView& g, const Params& p, const Pass& ps, const float4* lds_tau,
                                           const float* __restrict__ px, const float* __restrict__ py,
                                           const float* __restrict__ pz, const Lists& ls, int base1, int n2, int rot0,
                                           int lane) {
  DLIOM_BOX_STAT(p, lane == 0, 6, n2);  // level-2 entries
  for (int e0 = 0; e0 < n2; e0 += 64) {
    const int e = e0 + lane;
    if (e < n2) {
      const unsigned item = ls.l2[e];
      const unsigned ent = ls.l1[base1 + static_cast<int>(item & 0xFFu)];
      const int j = static_cast<int>(item >> 8);
      const int* rc = ls.rec + 4 * static_cast<int>((ent >> 12) & (kRecords - 1));
      const int r = rot0 + static_cast<int>(ent & 63u), pt = rc[3] + static_cast<int>((ent >> 6) & 63u);
      const float4 qq = p.rot[r];
      const Quat4 q{qq.x, qq.y, qq.z, qq.w};
      float rx, ry, rz;
      rotate_point(q, px[pt], py[pt], pz[pt], rx, ry, rz);
      // the cell the fast path read
      const float4 t = lds_tau[j];
      const float ax = __builtin_fmaf(rx, g.inv_resolution, static_cast<float>(ps.gi[0] - rc[0]) + ps.f[0]) + t.x;
      const float ay = __builtin_fmaf(ry, g.inv_resolution, static_cast<float>(ps.gi[1] - rc[1]) + ps.f[1]) + t.y;
      const float az = __builtin_fmaf(rz, g.inv_resolution, static_cast<float>(ps.gi[2] - rc[2]) + ps.f[2]) + t.z;
      const int fx = static_cast<int>((__float_as_uint(ax) >> 16) - 0x4300u) + rc[0];
      const int fy = static_cast<int>((__float_as_uint(ay) >> 16) - 0x4300u) + rc[1];
      const int fz = static_cast<int>((__float_as_uint(az) >> 16) - 0x4300u) + rc[2];
      // the reference's cell (hybrid_grid.h:430-435)
      const float* tr = p.trans + 3 * (ps.j0 + j);
      const int ex = cell_of(rx + tr[0], g.resolution), ey = cell_of(ry + tr[1], g.resolution),
                ez = cell_of(rz + tr[2], g.resolution);
      if (ex != fx || ey != fy || ez != fz) {
        if (abs(ex - fx) > 1 || abs(ey - fy) > 1 || abs(ez - fz) > 1) {  // cannot happen (error budget above)
          atomicOr(p.error, 1u);
          atomicOr(p.error + 1, 1u);  // the host redoes the match on the dense kernel
        }
        const long long delta = static_cast<long long>(mirror_value(g, ex, ey, ez)) -
                                static_cast<long long>(mirror_value(g, fx, fy, fz));
        if (delta != 0)
          atomicAdd(&p.sums[static_cast<size_t>(ps.j0 + j) * p.R + r], static_cast<unsigned long long>(delta));
      }
    }
  }
}

// Level 1: up to 64 listed (rotation, point) pairs, one per lane: which of the pass's translations put
// the point inside a rounding band?  Those lookups go to the level-2 list.
__device__ __forceinline__ void resolve_l1(const GridView& g, const Params& p, const Pass& ps, const float4* lds_tau,
                                           const float* __restrict__ px, const float* __restrict__ py,
                                           const float* __restrict__ pz, const Lists& ls, int base1, int count, int rot0,
                                           int lane) {
  const bool have = lane < count;
  DLIOM_BOX_STAT(p, lane == 0, 4, 1);      // level-1 rounds (one wave each)
  DLIOM_BOX_STAT(p, lane == 0, 5, count);  // level-1 entries
  float wx = 200.5f, wy = 200.5f, wz = 200.5f;  // idle lanes: a fraction of one half is never listed
  if (have) {
    const unsigned ent = ls.l1[base1 + lane];
    const int* rc = ls.rec + 4 * static_cast<int>((ent >> 12) & (kRecords - 1));
    const int r = rot0 + static_cast<int>(ent & 63u), pt = rc[3] + static_cast<int>((ent >> 6) & 63u);
    const float4 qq = p.rot[r];
    const Quat4 q{qq.x, qq.y, qq.z, qq.w};
    float rx, ry, rz;
    rotate_point(q, px[pt], py[pt], pz[pt], rx, ry, rz);
    wx = __builtin_fmaf(rx, g.inv_resolution, static_cast<float>(ps.gi[0] - rc[0]) + ps.f[0]);
    wy = __builtin_fmaf(ry, g.inv_resolution, static_cast<float>(ps.gi[1] - rc[1]) + ps.f[1]);
    wz = __builtin_fmaf(rz, g.inv_resolution, static_cast<float>(ps.gi[2] - rc[2]) + ps.f[2]);
  }
  int n2 = 0;
  for (int j = 0; j < ps.tc; ++j) {
    const float4 t = lds_tau[j];
    unsigned m = min_lo16(0xFFFFFFFFu, wx + t.x);
    m = min_lo16(m, wy + t.y);
    m = min_lo16(m, wz + t.z);
    const bool mine = have && m <= p.thr;
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(mine);
    if (mask != 0ull) {
      if (n2 + 64 > kL2Cap) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        resolve_l2(g, p, ps, lds_tau, px, py, pz, ls, base1, n2, rot0, lane);
        __builtin_amdgcn_wave_barrier();
        n2 = 0;
      }
      const int rank = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mask >> 32),
                                                 __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mask), 0u));
      if (mine) ls.l2[n2 + rank] = static_cast<unsigned>(lane) | (static_cast<unsigned>(j) << 8);
      n2 += __builtin_popcountll(mask);
    }
  }
  if (n2 > 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    resolve_l2(g, p, ps, lds_tau, px, py, pz, ls, base1, n2, rot0, lane);
    __builtin_amdgcn_wave_barrier();
  }
}

// Drains the level-1 list in groups of 64; keeps a remainder below 64 unless `all`.
__device__ __forceinline__ void drain_l1(const GridView& g, const Params& p, const Pass& ps, const float4* lds_tau,
                                         const float* __restrict__ px, const float* __restrict__ py,
                                         const float* __restrict__ pz, Lists& ls, bool all, int rot0, int lane) {
  int base = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  while (ls.n1 - base >= 64 || (all && ls.n1 > base)) {
    const int count = min(64, ls.n1 - base);
    resolve_l1(g, p, ps, lds_tau, px, py, pz, ls, base, count, rot0, lane);
    base += count;
  }
  if (base > 0) {  // move the remainder to the front
    const int rest = ls.n1 - base;
    unsigned v = 0;
    if (lane < rest) v = ls.l1[base + lane];
    __builtin_amdgcn_wave_barrier();
    if (lane < rest) ls.l1[lane] = v;
    ls.n1 = rest;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- the hot loop --------------------------------------------------------------------------------------
// Returns the first point NOT processed: the loop stops early when the level-1 list could not take the
// worst case of another iteration (every lane listed for every point) -- the caller drains it and re-enters.
template <int P, int TC>
__device__ __forceinline__ int main_loop(const GridView& g, const Params& p, const Geometry& geo, const Quat4 q,
                                          const float* __restrict__ px, const float* __restrict__ py,
                                          const float* __restrict__ pz, int i_begin, int i_end, int chunk_lo,
                                          const float4* lds_tau, const unsigned* lds_bitmap, unsigned box_base,
                                          Lists& ls, unsigned rec_id, bool lane_active, int lane,
                                          unsigned (&acc)[TC]) {
  const float inv = to_vgpr_f(g.inv_resolution);
  const float kbx = to_vgpr_f(geo.kb[0]), kby = to_vgpr_f(geo.kb[1]), kbz = to_vgpr_f(geo.kb[2]);
  const unsigned s1 = 2u * geo.sx, s2 = 2u * geo.sxy;
  const unsigned v2 = to_vgpr(2u), v1 = to_vgpr(s1), vs2 = to_vgpr(s2);
  const unsigned d0 = to_vgpr(box_base - 0x4300u * (2u + s1 + s2));
  int i = i_begin;
  // the points of the NEXT iteration are fetched while this one's 27 * P lookups run: the loads used to be waited
  // for right where they were issued, a full memory latency per iteration with nothing else in flight
  float nx[P], ny[P], nz[P];
#pragma unroll
  for (int k = 0; k < P; ++k) {
    nx[k] = px[i + k];
    ny[k] = py[i + k];
    nz[k] = pz[i + k];
  }
#pragma unroll 1
  for (; i < i_end; i += P) {
    if (ls.n1 + 64 * P > kL1Cap) break;
    float cx[P], cy[P], cz[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
      cx[k] = nx[k];
      cy[k] = ny[k];
      cz[k] = nz[k];
    }
    {
      const int in = min(i + P, i_end - P);  // the last iteration reloads its own points (in range, unused)
#pragma unroll
      for (int k = 0; k < P; ++k) {
        nx[k] = px[in + k];
        ny[k] = py[in + k];
        nz[k] = pz[in + k];
      }
    }
    float wx[P], wy[P], wz[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
      float rx, ry, rz;
#if DLIOM_BOX_EXP == 4
      // timing experiment (wrong sums): no rotation in the loop -- the ceiling of any scheme that takes the 33-instruction
      // quaternion rotation off the vector ALU (a matrix-core rotation, a pre-rotated table); q.w keeps the lanes apart
      rx = cx[k] + q.w, ry = cy[k] + q.x, rz = cz[k] + q.y;
#else
      rotate_point(q, cx[k], cy[k], cz[k], rx, ry, rz);
#endif
      wx[k] = __builtin_fmaf(rx, inv, kbx);
      wy[k] = __builtin_fmaf(ry, inv, kby);
      wz[k] = __builtin_fmaf(rz, inv, kbz);
    }
    // once per (rotation, point): can any translation of the pass put a coordinate into a rounding band?
    // All 3 P bitmap words are fetched together (one LDS latency per iteration, not P) and the list append is
    // branch-free: a lane that is not listed writes to the wave's scratch word.  The per-point version -- fetch,
    // wait, ballot, branch, append -- exposed an LDS round trip and two branches per point with nothing else to
    // issue, a third of the wave's time at 2-3 waves per SIMD.
    unsigned bkt[P][3], word[P][3];
#pragma unroll
    for (int k = 0; k < P; ++k) {
      // the scaled coordinate lies in [128, 256): its low 16 bits are the fraction, bucket = fraction >> 3
      bkt[k][0] = (__float_as_uint(wx[k]) >> 3) & (kBuckets - 1);
      bkt[k][1] = (__float_as_uint(wy[k]) >> 3) & (kBuckets - 1);
      bkt[k][2] = (__float_as_uint(wz[k]) >> 3) & (kBuckets - 1);
    }
#pragma unroll
    for (int k = 0; k < P; ++k) {
      word[k][0] = lds_bitmap[bkt[k][0] >> 5];
      word[k][1] = lds_bitmap[kBuckets / 32 + (bkt[k][1] >> 5)];
      word[k][2] = lds_bitmap[2 * (kBuckets / 32) + (bkt[k][2] >> 5)];
    }
#pragma unroll
    for (int k = 0; k < P; ++k) {
      const unsigned hit = ((word[k][0] >> (bkt[k][0] & 31u)) | (word[k][1] >> (bkt[k][1] & 31u)) | (word[k][2] >> (bkt[k][2] & 31u))) & 1u;
      const bool mine = lane_active && hit != 0u;
      const unsigned long long mask = __builtin_amdgcn_ballot_w64(mine);
      const int rank = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mask >> 32),
                                                 __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mask), 0u));
      const int slot = mine ? ls.n1 + rank : kListTrash;
      ls.l1[slot] = static_cast<unsigned>(lane) | (static_cast<unsigned>(i + k - chunk_lo) << 6) | (rec_id << 12);
      ls.n1 += __builtin_popcountll(mask);
    }
    // software pipeline, fixed by scheduling barriers: the translation of step j + 1 is fetched at the top of step j;
    // the values gathered in step j - kPipe are accumulated AFTER step j's address arithmetic and right before its own
    // gathers are issued, so that a gather has kPipe whole steps of another ~85 VALU cycles to return (accumulating
    // in the middle of the next step's arithmetic left ~45 cycles: the wave then sat in s_waitcnt 40 % of the time)
    float4 t = lds_tau[0];
    unsigned pv[kPipe][P];
#pragma unroll
    for (int d = 0; d < kPipe; ++d)
#pragma unroll
      for (int k = 0; k < P; ++k) pv[d][k] = 0u;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < TC; ++j) {
#if DLIOM_BOX_EXP == 3
      const float4 tn = t;  // timing experiment: no translation fetch
#else
      const float4 tn = lds_tau[j + 1 < TC ? j + 1 : j];  // same address in every lane: LDS broadcast
#endif
      unsigned a[P];
#if DLIOM_BOX_EXP == 2
      // timing experiment: no address arithmetic (wrong sums)
#pragma unroll
      for (int k = 0; k < P; ++k) a[k] = d0 + 0x4300u * (2u + s1 + s2) + static_cast<unsigned>(2 * (4 * j + k)) + (__float_as_uint(wx[k]) & 0x3cu);
      if (false) {
#else
      if (P % 2 == 0) {
#endif
        // level by level over the P lookups: a dependent v_mad_u32_u16 right behind its producer costs a wait state
        const float2v txy = {t.x, t.y}, tz0 = {t.z, t.w};
        float2v ax[P / 2 + 1], ay[P / 2 + 1], az[P / 2 + 1];  // + 1: no zero-length arrays in the P = 1 instantiation
#pragma unroll
        for (int k = 0; k < P / 2; ++k) {
          ax[k] = pk_add_lo(float2v{wx[2 * k], wx[2 * k + 1]}, txy);
          ay[k] = pk_add_hi(float2v{wy[2 * k], wy[2 * k + 1]}, txy);
          az[k] = pk_add_lo(float2v{wz[2 * k], wz[2 * k + 1]}, tz0);
        }
#pragma unroll
        for (int k = 0; k < P / 2; ++k) {
          a[2 * k] = mad_hi16(ax[k].x, v2, d0);
          a[2 * k + 1] = mad_hi16(ax[k].y, v2, d0);
        }
#pragma unroll
        for (int k = 0; k < P / 2; ++k) {
          a[2 * k] = mad_hi16(ay[k].x, v1, a[2 * k]);
          a[2 * k + 1] = mad_hi16(ay[k].y, v1, a[2 * k + 1]);
        }
#pragma unroll
        for (int k = 0; k < P / 2; ++k) {
          a[2 * k] = mad_hi16(az[k].x, vs2, a[2 * k]);
          a[2 * k + 1] = mad_hi16(az[k].y, vs2, a[2 * k + 1]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < P; ++k) {
          const float ax = wx[k] + t.x, ay = wy[k] + t.y, az = wz[k] + t.z;
          a[k] = mad_hi16(az, vs2, mad_hi16(ay, v1, mad_hi16(ax, v2, d0)));
        }
      }
      if (kLateAcc) __builtin_amdgcn_sched_barrier(0);
      if (j >= kPipe) {
        unsigned s = pv[kPipe - 1][0];
#pragma unroll
        for (int k = 1; k < P; ++k) s += pv[kPipe - 1][k];
        acc[j - kPipe] += s;
        asm volatile("" : "+v"(acc[j - kPipe]));  // keeps the add here (it would be sunk to the loop latch, values spilled)
      }
      if (kLateAcc) __builtin_amdgcn_sched_barrier(0);
      unsigned v[P];
#if DLIOM_BOX_EXP == 1
#pragma unroll
      for (int k = 0; k < P; ++k) v[k] = a[k] & 0x7fffu;  // timing experiment: no gathers (wrong sums)
#else
#pragma unroll
      for (int k = 0; k < P; ++k) v[k] = *reinterpret_cast<lds_cu16*>(a[k]);
#endif
#pragma unroll
      for (int d = kPipe - 1; d > 0; --d)
#pragma unroll
        for (int k = 0; k < P; ++k) pv[d][k] = pv[d - 1][k];
#pragma unroll
      for (int k = 0; k < P; ++k) pv[0][k] = v[k];
      t = tn;
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int d = kPipe - 1; d >= 0; --d) {
      unsigned s = pv[d][0];
#pragma unroll
      for (int k = 1; k < P; ++k) s += pv[d][k];
      acc[TC - 1 - d] += s;
      asm volatile("" : "+v"(acc[TC - 1 - d]));
    }
  }
  return i;
}

// The register accumulators are 32 bits wide and a value is at most 32767: after kFlushPoints points they are
// added to the 64-bit score volume and cleared (large clouds: 128 x 2048 returns and up).
constexpr int kFlushPoints = 131072;  // 131072 * 32767 < 2^32
template <int TC>
__device__ __forceinline__ void flush_acc(const Params& p, const Pass& ps, unsigned (&acc)[TC], int r, bool lane_active) {
#pragma unroll
  for (int j = 0; j < TC; ++j) {
    if (lane_active && j < ps.tc && acc[j] != 0u)
      atomicAdd(&p.sums[static_cast<size_t>(ps.j0 + j) * p.R + r], static_cast<unsigned long long>(acc[j]));
    acc[j] = 0u;
  }
}

// Bounding box of every lookup of the points held by lanes [0, n) (per-lane interval ends l3 / h3, in cells, pass centre
// included) under a workgroup's rotations and a pass's translations: wave-uniform geometry of the LDS box, false if it
// does not fit.  (Round 5 planned the boxes of every chunk ONCE in a pre-pass instead of in all four waves of every
// workgroup -- 10 % of the kernel's vector instructions gone -- and the kernel took exactly as long: 0.733 ms against
// 0.733 ms, A/B in one process, plus 32 us for the planning kernel.  The instructions were never the critical path;
// reverted)
__device__ __forceinline__ bool box_geometry(const GridView& g, const Params& p, const Pass& ps, const float (&l3)[3],
                                             const float (&h3)[3], int n, int lane, Geometry& geo) {
  const bool have = lane < n;
  bool fits = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    // clamp far outliers: a box that large is rejected below anyway
    const float l = fmaxf(wave_min_f(have ? l3[a] : 3.0e38f) - ps.reach[a], -1.0e6f);
    const float h = fminf(wave_max_f(have ? h3[a] : -3.0e38f) + ps.reach[a], 1.0e6f);
    geo.lo[a] = __builtin_amdgcn_readfirstlane(static_cast<int>(floorf(l)));
    geo.dim[a] = __builtin_amdgcn_readfirstlane(static_cast<int>(floorf(h))) - geo.lo[a] + 1;
  }
  {  // x: whole 4-cell groups of the bricked mirror (mirror coordinate = index + dense_off)
    const int m_lo = (geo.lo[0] + g.dense_off[0]) & ~3;
    const int m_hi = geo.lo[0] + g.dense_off[0] + geo.dim[0];  // exclusive
    geo.lo[0] = m_lo - g.dense_off[0];
    geo.dim[0] = ((m_hi - m_lo) + 3) & ~3;
  }
  geo.sx = static_cast<unsigned>(geo.dim[0]);
  if (((geo.sx >> 2) & 1u) == 0u) geo.sx += 4u;  // row stride = 2 * odd dwords: rows spread over banks
  const unsigned sy = static_cast<unsigned>(geo.dim[1]) | 1u;
  geo.sxy = geo.sx * sy;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    fits = fits && geo.dim[a] <= kMaxDim && abs(ps.gi[a] - geo.lo[a]) < 1000;
    geo.kb[a] = static_cast<float>(ps.gi[a] - geo.lo[a]) + ps.f[a];
  }
  fits = fits && static_cast<long long>(geo.sxy) * geo.dim[2] <= p.cells && geo.sxy <= 32767u;
  return fits;
}
// the per-lane interval ends of box_geometry for point `i` (pre-pass table + the pass's centre)
__device__ __forceinline__ void point_interval(const Params& p, const Pass& ps, const float* __restrict__ ext_rb, int i,
                                               float (&l3)[3], float (&h3)[3]) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    // 0.0502 instead of the 0.05 of the fused form fma(c -+ he, inv, uc): two more roundings of values < 1024
    l3[a] = (ext_rb[static_cast<size_t>(2 * a) * p.ext_stride + i] + ps.uc[a]) - 0.0502f;
    h3[a] = (ext_rb[static_cast<size_t>(2 * a + 1) * p.ext_stride + i] + ps.uc[a]) + 0.0502f;
  }
}

// Pre-pass of a launch: for every (rotation block, point) the interval per axis, in cells and before the pass's
// translation, that contains the point's image under every rotation of the block:
//   R_c (p + dc x p)  +-  |R_c| (hd (x) |p|)  +-  theta2 |p|      (metres)
// around the centre rotation's image -- first-order spread of the rotations plus the second-order bound (Group, host,
// double).  The score kernel used to compute this per box in every wave (~390 vector instructions an attempt, 10 % of
// everything it issued, and on the critical path of a workgroup between two boxes); now a box's extent is six loads
// and the reductions.  One thread per point, blockIdx.y = rotation block.
// Round 5: the workgroups with blockIdx.y == rot_blocks carry the match's pending copies and fills (PrepArgs,
// device_common.h) -- the candidate and box tables out of pinned host memory, the zeroed score volume and counters -- so
// this kernel is the first of the chain and therefore reads ITS OWN inputs (groups, rotations) from the staged copies
// (`group_src`, `rot_src`: the same bytes the prep workgroups are copying).
__global__ __launch_bounds__(256) void rtcsm_box_extent_kernel(Params p, float inv, const float* __restrict__ px,
                                                               const float* __restrict__ py, const float* __restrict__ pz,
                                                               float* __restrict__ ext, const Group* __restrict__ group_src,
                                                               const float4* __restrict__ rot_src, PrepArgs prep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, rb = blockIdx.y;
  if (rb >= p.rot_blocks) {
    prep_block(prep, blockIdx.x, gridDim.x, threadIdx.x, blockDim.x);
    return;
  }
  if (i >= p.n) return;
  const Group grp = group_src[rb];
  const float4 qcc = rot_src[p.r_first + rb * p.nw * 64 + grp.c_lane];
  const Quat4 qc{qcc.x, qcc.y, qcc.z, qcc.w};
  float rabs[9];  // |R(qc)|, row major
  {
    const float w = qc.w, x = qc.x, y = qc.y, z = qc.z;
    rabs[0] = fabsf(1.f - 2.f * (y * y + z * z));
    rabs[1] = fabsf(2.f * (x * y - w * z));
    rabs[2] = fabsf(2.f * (x * z + w * y));
    rabs[3] = fabsf(2.f * (x * y + w * z));
    rabs[4] = fabsf(1.f - 2.f * (x * x + z * z));
    rabs[5] = fabsf(2.f * (y * z - w * x));
    rabs[6] = fabsf(2.f * (x * z - w * y));
    rabs[7] = fabsf(2.f * (y * z + w * x));
    rabs[8] = fabsf(1.f - 2.f * (x * x + y * y));
  }
  const float x = px[i], y = py[i], z = pz[i];
  const float sx_ = grp.dc[1] * z - grp.dc[2] * y, sy_ = grp.dc[2] * x - grp.dc[0] * z, sz_ = grp.dc[0] * y - grp.dc[1] * x;
  float cx, cy, cz;
  rotate_point(qc, x + sx_, y + sy_, z + sz_, cx, cy, cz);
  const float fx = fabsf(x), fy = fabsf(y), fz = fabsf(z);
  const float hx = grp.hd[1] * fz + grp.hd[2] * fy, hy = grp.hd[2] * fx + grp.hd[0] * fz, hz = grp.hd[0] * fy + grp.hd[1] * fx;
  const float m2 = grp.theta2 * (fx + fy + fz) + 1.0e-5f * (fx + fy + fz);
  const float c3[3] = {cx, cy, cz};
  float* out = ext + static_cast<size_t>(rb) * 6 * p.ext_stride + i;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float he = rabs[3 * a] * hx + rabs[3 * a + 1] * hy + rabs[3 * a + 2] * hz + m2;
    out[static_cast<size_t>(2 * a) * p.ext_stride] = (c3[a] - he) * inv;
    out[static_cast<size_t>(2 * a + 1) * p.ext_stride] = (c3[a] + he) * inv;
  }
}

// The kernel's body for TC translations per pass.  Three instantiations (round 6, measured on BASELINE configs 2 and 5,
// DESIGN.md 3.1): <27> at four waves per SIMD (128 registers, 14 336-cell boxes: the fastest where one pass holds the
// whole translation window -- config 2: 0.69 ms against 0.71 at three waves, with or without larger boxes);
// <27> at three waves per SIMD (168 registers, 21 000-cell boxes, 64-point chunks) and <49> at three waves per SIMD for
// windows of several passes, where the translations' reach makes the boxes larger (config 5: 221 -> 188 -> 159 ms).
template <int TC>
__device__ __forceinline__ void score_box_body(const GridView& g, const Params& p, const float* __restrict__ px,
                                               const float* __restrict__ py, const float* __restrict__ pz) {
  extern __shared__ float4 lds_dyn4[];  // [kTC tau | band bitmap | ticket words | nw x lists | box]
  float4* lds_tau = lds_dyn4;
  unsigned* lds_bitmap = reinterpret_cast<unsigned*>(lds_dyn4 + TC);
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)), lane = threadIdx.x & 63, nthreads = blockDim.x;
  typedef __attribute__((address_space(3))) char lds_char;
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<size_t>((lds_char*)lds_dyn4));  // LDS byte address of the block
  const unsigned tick_off = static_cast<unsigned>(TC * sizeof(float4)) + kBitmapWords * 4u;
  const unsigned lists_off = tick_off + 16u;
  const unsigned box_off = lists_off + static_cast<unsigned>(p.nw) * kListWords * 4u;  // lists of the waves present
  const unsigned box_base = lds0 + box_off;
  unsigned short* box = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(lds_dyn4) + box_off);
  unsigned* tick = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(lds_dyn4) + tick_off);
  Lists ls;
  ls.l1 = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(lds_dyn4) + lists_off) + wave * kListWords;
  ls.l2 = ls.l1 + kL1Cap;
  ls.rec = reinterpret_cast<int*>(ls.l2 + kL2Cap);
  ls.n1 = 0;
  ls.seq = 0;
  unsigned acc[TC];
#pragma unroll
  for (int j = 0; j < TC; ++j) acc[j] = 0u;
  const int S = g.dense_stride, B = g.dense_bricks;
  // Work units are (pass, rotation block); each has its own chunk dispenser.  A workgroup starts in its HOME unit
  // (blockIdx -> unit as evenly as the launch allows; its first ticket is its own slot, no atomic: same-address
  // atomics serialise at ~0.15 us each) and, when that unit's tickets are gone, moves on to the other units and helps
  // there (its accumulators are flushed per unit).  A ticket is ONE chunk: with tickets of four chunks a unit's 512
  // tickets over 170 workgroups left all but two of them idle for the length of a fourth ticket -- a quarter of the
  // kernel (profiles/r2_pmc_score_kernel.json: 2.7 of 4 resident waves per SIMD on average; round 3, same box:
  // 0.871 ms with tickets of four, 0.797 with 4-2-1, 0.767 with single chunks).  Tickets map to chunks through
  // Params::order, most expensive chunks first (a chunk across a jump of the Morton curve takes several boxes and up to
  // five times the time of a compact one: handed out last it WAS the end of the launch), and the next ticket is drawn
  // late (below); same box, 20 launches each: 0.794 ms index order + early draw, 0.784 / 0.779 with one of the two,
  // 0.770 with both.  Control flow below is uniform over the WORKGROUP (barriers).
#ifdef DLIOM_EXPERIMENTS
  unsigned long long stamp_begin = 0ull;
  unsigned stamp_tickets = 0u;
  if (DLIOM_BOX_DBG(p, 256)) stamp_begin = wall_clock64();
#endif
  const int home = static_cast<int>(blockIdx.x % static_cast<unsigned>(p.units));
  const int slot = static_cast<int>(blockIdx.x / static_cast<unsigned>(p.units));
  int parity = 0;
  int loaded_pass = -1;
  const int hops = DLIOM_BOX_DBG(p, 64) ? 1 : p.units;
  for (int hop = 0; hop < hops; ++hop) {
    int unit = home + hop;
    if (unit >= p.units) unit -= p.units;
    unsigned* counter = p.counters + unit;
    int ticket, chunk_id;
    if (hop == 0) {
      ticket = DLIOM_BOX_DBG(p, 8) ? p.point_chunks : slot;
      chunk_id = (p.order != nullptr && ticket < p.point_chunks) ? static_cast<int>(p.order[ticket]) : ticket;
    } else {
      if (threadIdx.x == 0) {
        const int t = p.slots + static_cast<int>(atomicAdd(counter, 1u));
        tick[parity] = static_cast<unsigned>(t);
        tick[2 + parity] = (p.order != nullptr && t < p.point_chunks) ? p.order[t] : static_cast<unsigned>(t);
      }
      __syncthreads();
      ticket = static_cast<int>(tick[parity]);
      chunk_id = static_cast<int>(tick[2 + parity]);
      parity ^= 1;
    }
    if (ticket >= p.point_chunks) continue;  // nothing left here
    // this unit: (pass, rotation block)
    const int rb = unit % p.rot_blocks;
    const int tp = unit / p.rot_blocks;
    if (tp != loaded_pass) {
      __syncthreads();  // nobody still reads the previous pass's tables
      if (threadIdx.x < TC) lds_tau[threadIdx.x] = p.tau[tp * TC + threadIdx.x];
      for (int w = threadIdx.x; w < kBitmapWords; w += nthreads) lds_bitmap[w] = p.bitmap[tp * kBitmapWords + w];
      __syncthreads();
      loaded_pass = tp;
    }
    const int rot_b0 = p.r_first + rb * p.nw * 64;  // first rotation of the workgroup
    // Round 5: a unit with fewer rotation groups than the workgroup has waves -- the LAST rotation block: 1331 rotations
    // are five blocks of 256 and one of 51 -- used to run its one group on one wave while three waves idled, and since a
    // ticket takes as long with one wave as with four, that block's 2048 tickets cost a sixth of the launch's
    // workgroup-time for 4 % of its lookups.  Now the waves SHARE the groups: with g groups, wave w takes group w % g and
    // slice w / g of every box's points (nw / g slices), all slices adding into the same candidates' sums (the flush is
    // an atomic add anyway).
    const int groups_here = (p.r_last - rot_b0 + 63) >> 6;  // >= 1; more than nw: a full block
    // (four-wave workgroups only: one group -> four slices, two groups -> two slices each; no divisions here)
    const int split = (p.split_points != 0 && p.nw == 4) ? (groups_here == 1 ? 4 : (groups_here == 2 ? 2 : 1)) : 1;
    const int rot_group = split == 4 ? 0 : (split == 2 ? (wave & 1) : wave);
    const int slice = split == 4 ? wave : (split == 2 ? (wave >> 1) : 0);
    const int rot0 = rot_b0 + rot_group * 64;       // first rotation of this wave
    const bool wave_active = rot0 < p.r_last && slice < split;  // surplus waves only help staging
    const Pass ps = p.pass[tp];
    const bool lane_active = rot0 + lane < p.r_last;
    const float4 qq = p.rot[lane_active ? rot0 + lane : (wave_active ? rot0 : rot_b0)];  // idle lanes shadow a real rotation
    const Quat4 q{qq.x, qq.y, qq.z, qq.w};
    const float* ext_rb = p.ext + static_cast<size_t>(rb) * 6 * p.ext_stride;
    int n_guess = p.chunk;  // points per box that fitted last time
    int since_flush = 0;    // points added to the accumulators since they were last cleared (uniform)
    while (ticket < p.point_chunks) {
      // The next ticket is drawn when the LAST box of this one is about to be staged (the atomic returns during the
      // staging, the order table's entry during the lookups): a ticket drawn at the start of the current one is a
      // chunk nobody else can take for a whole ticket's time, and at the end of the launch that left workgroups
      // idle for up to two tickets while others still held one in reserve (round 3: workgroups ended between 627
      // and 788 us of a 788 us launch).
      unsigned next_raw = 0u, next_chunk = 0u;
      bool drawn = false;
#ifdef DLIOM_EXPERIMENTS
      ++stamp_tickets;
#endif
      {
        const int c = __builtin_amdgcn_readfirstlane(chunk_id);
        const int c_begin = c * p.chunk, c_end = min(c_begin + p.chunk, p.n);
        int lo = c_begin;
        while (lo < c_end) {
          int n;
          Geometry geo;
          bool fits_out = false;
          {
            // ---- bounding box of every lookup of (these points) x (the workgroup's rotations) x (this pass): the
            //      per-point extents come from the pre-pass (lanes = POINTS here), every wave reduces the same values;
            //      a box that does not fit is retried with fewer points -- only the reductions are redone
            n = min(c_end - lo, n_guess);
            float l3[3], h3[3];
            point_interval(p, ps, ext_rb, lo + (lane < n ? lane : 0), l3, h3);
            for (;;) {
              fits_out = box_geometry(g, p, ps, l3, h3, n, lane, geo);
              DLIOM_BOX_STAT(p, threadIdx.x == 0, 7, 1);  // bounding boxes computed
              if (fits_out) break;
              if (n == 1) break;
              n = n > 4 ? (((n >> 1) + 3) & ~3) : (n >> 1);
            }
          }
          n_guess = min(p.chunk, n >= 4 ? 2 * n : 4);
          if (since_flush + n > kFlushPoints) {
            flush_acc(p, ps, acc, rot0 + lane, lane_active);
            since_flush = 0;
          }
          since_flush += fits_out ? n : 1;
          if (!fits_out) {
            // a single point whose lookups do not fit the box (huge angular window / far outlier): the exact
            // path straight from the mirror in HBM, every lane its own rotation
            if (slice == 0) {  // once per rotation: the waves that share a rotation group split the BOXES' points only
              float rx, ry, rz;
              rotate_point(q, px[lo], py[lo], pz[lo], rx, ry, rz);
#pragma unroll
              for (int j = 0; j < TC; ++j) {
                const float* tr = p.trans + 3 * (ps.j0 + min(j, ps.tc - 1));
                acc[j] += mirror_value(g, cell_of(rx + tr[0], g.resolution), cell_of(ry + tr[1], g.resolution),
                                       cell_of(rz + tr[2], g.resolution));
              }
            }
            lo += 1;
            DLIOM_BOX_STAT(p, threadIdx.x == 0, 3, 1);  // points on the exact path
            continue;
          }
          if (!drawn && (lo + n >= c_end || DLIOM_BOX_DBG(p, 1024))) {  // uniform: the last box of the ticket
            if (threadIdx.x == 0) next_raw = atomicAdd(counter, 1u);
            drawn = true;
          }
          DLIOM_BOX_STAT(p, threadIdx.x == 0, 0, 1);                                   // boxes
          DLIOM_BOX_STAT(p, threadIdx.x == 0, 1, (geo.dim[0] >> 2) * geo.dim[1] * geo.dim[2]);  // staged quads
          DLIOM_BOX_STAT(p, threadIdx.x == 0, 2, n);                                   // points in boxes
          __syncthreads();  // every wave is done with the previous box
          // ---- stage the box, all threads: 4-cell groups (8 bytes) of the bricked mirror, outside reads 1
          if (!DLIOM_BOX_DBG(p, 2)) {
            // index arithmetic in 24-bit multiplies and one 32-bit byte offset: v_mul_lo_u32 and the 64-bit mads of
            // the plain form are quarter-rate instructions, seven of them per 8-byte group were a tenth of the
            // kernel's vector-ALU time (every factor here is below 2^24, the mirror below 4 GB)
            const unsigned quads = static_cast<unsigned>(geo.dim[0]) >> 2, dim1 = static_cast<unsigned>(geo.dim[1]);
            const unsigned total = __umul24(__umul24(quads, dim1), static_cast<unsigned>(geo.dim[2]));
            const float inv_q = 1.0f / static_cast<float>(quads), inv_dy = 1.0f / static_cast<float>(dim1);
            const int bx0 = geo.lo[0] + g.dense_off[0], by0 = geo.lo[1] + g.dense_off[1], bz0 = geo.lo[2] + g.dense_off[2];
            const unsigned uB = static_cast<unsigned>(B), uBB = __umul24(uB, uB), uS = static_cast<unsigned>(S);
            const char* dense_bytes = reinterpret_cast<const char*>(g.dense);
#pragma unroll 4
            for (unsigned e = threadIdx.x; e < total; e += static_cast<unsigned>(nthreads)) {
              const unsigned row = static_cast<unsigned>((static_cast<float>(e) + 0.5f) * inv_q);  // exact: e < 2^21
              const unsigned xq = e - __umul24(row, quads);
              const unsigned z = static_cast<unsigned>((static_cast<float>(row) + 0.5f) * inv_dy);
              const unsigned y = row - __umul24(z, dim1);
              const int mx = bx0 + static_cast<int>(4u * xq), my = by0 + static_cast<int>(y), mz = bz0 + static_cast<int>(z);
              unsigned long long val = 0x0001000100010001ull;
              if (static_cast<unsigned>(mx) < 4u * uB && static_cast<unsigned>(my) < uS && static_cast<unsigned>(mz) < uS) {
                const unsigned umx = static_cast<unsigned>(mx), umy = static_cast<unsigned>(my), umz = static_cast<unsigned>(mz);
                const unsigned off = ((__umul24(umz >> 2, uBB) + __umul24(umy >> 2, uB) + (umx >> 2)) << 7) |
                                     (((umz & 3u) << 5) | ((umy & 3u) << 3));
                val = *reinterpret_cast<const unsigned long long*>(dense_bytes + off);
              }
              *reinterpret_cast<unsigned long long*>(box + (__umul24(z, geo.sxy) + __umul24(y, geo.sx) + 4u * xq)) = val;
            }
          }
          // chunk record for this wave's level-1 entries of these points
          const unsigned rec_id = static_cast<unsigned>(ls.seq & (kRecords - 1));
          if (lane < 4) ls.rec[4 * rec_id + lane] = lane < 3 ? geo.lo[lane] : lo;
          ++ls.seq;
          __syncthreads();  // the box is complete
          if (drawn && threadIdx.x == 0) {  // in flight during the lookups
            const int t = p.slots + static_cast<int>(next_raw);
            next_chunk = (p.order != nullptr && t < p.point_chunks) ? p.order[t] : static_cast<unsigned>(t);
          }
          // ---- all lookups of these points under this wave's rotations
          if (wave_active && !DLIOM_BOX_DBG(p, 4)) {
            // this wave's slice of the box's points (the whole box unless rotation groups are shared): multiples of 4
            const int part = split > 1 ? ((((n + split - 1) >> (split == 4 ? 2 : 1)) + 3) & ~3) : n;
            const int s_lo = lo + min(n, slice * part), s_n = min(n, (slice + 1) * part) - min(n, slice * part);
            int i = s_lo;
            const int e8 = kHotP == 8 ? s_lo + (s_n & ~7) : s_lo, e4 = s_lo + (s_n & ~3), e1 = s_lo + s_n;
            for (;;) {
              if (kHotP == 8 && i < e8)
                i = main_loop<kHotP, TC>(g, p, geo, q, px, py, pz, i, e8, lo, lds_tau, lds_bitmap, box_base, ls, rec_id, lane_active, lane, acc);
              if (i >= e8 && i < e4)
                i = main_loop<4, TC>(g, p, geo, q, px, py, pz, i, e4, lo, lds_tau, lds_bitmap, box_base, ls, rec_id, lane_active, lane, acc);
              if (i >= e4 && i < e1)
                i = main_loop<1, TC>(g, p, geo, q, px, py, pz, i, e1, lo, lds_tau, lds_bitmap, box_base, ls, rec_id, lane_active, lane, acc);
              if (i >= e1) break;
              drain_l1(g, p, ps, lds_tau, px, py, pz, ls, false, rot0, lane);  // the list was too full to go on
            }
            // ---- listed pairs: 64 at a time; everything before the oldest record is overwritten
            if (!DLIOM_BOX_DBG(p, 1) && (ls.n1 >= 64 || (ls.seq & (kRecords - 1)) == 0))
              drain_l1(g, p, ps, lds_tau, px, py, pz, ls, (ls.seq & (kRecords - 1)) == 0, rot0, lane);
          }
          lo += n;
        }
      }
      // the next ticket, through LDS (two sets of words used alternately: one barrier per ticket)
      if (threadIdx.x == 0) {
        if (!drawn) {  // the ticket ended on the exact path (no box): draw now
          const int t = p.slots + static_cast<int>(atomicAdd(counter, 1u));
          next_raw = static_cast<unsigned>(t - p.slots);
          next_chunk = (p.order != nullptr && t < p.point_chunks) ? p.order[t] : static_cast<unsigned>(t);
        }
        tick[parity] = static_cast<unsigned>(p.slots) + next_raw;
        tick[2 + parity] = next_chunk;
      }
      __syncthreads();
      ticket = static_cast<int>(tick[parity]);
      chunk_id = static_cast<int>(tick[2 + parity]);
      parity ^= 1;
    }
    // leaving the unit: the lists refer to its rotations and pass, the accumulators to its candidates
    if (wave_active && !DLIOM_BOX_DBG(p, 1)) drain_l1(g, p, ps, lds_tau, px, py, pz, ls, true, rot0, lane);
    ls.n1 = 0;
    if (DLIOM_BOX_DBG(p, 16)) {  // timing experiment: unconditional atomics of the accumulators' flush, even for zeros
#pragma unroll
      for (int j = 0; j < TC; ++j) {
        if (lane_active && j < ps.tc)
          atomicAdd(&p.sums[static_cast<size_t>(ps.j0 + j) * p.R + rot0 + lane], static_cast<unsigned long long>(acc[j]));
        acc[j] = 0u;
      }
    } else if (!DLIOM_BOX_DBG(p, 32)) {
      flush_acc(p, ps, acc, rot0 + lane, lane_active);
    }
  }
#ifdef DLIOM_EXPERIMENTS
  if (DLIOM_BOX_DBG(p, 256) && threadIdx.x == 0 && blockIdx.x < 4096u) {
    unsigned long long* st = reinterpret_cast<unsigned long long*>(p.error + 64) + 4 * blockIdx.x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    st[0] = stamp_begin;
    st[1] = wall_clock64();
    st[2] = stamp_tickets;
    st[3] = xcc;
  }
#endif
}


__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(4, 8))) void rtcsm_score_box_kernel(
    GridView g, Params p, const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz) {
  score_box_body<kTC>(g, p, px, py, pz);
}
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(3, 8))) void rtcsm_score_box_kernel_w3(
    GridView g, Params p, const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz) {
  score_box_body<kTC>(g, p, px, py, pz);
}
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(3, 8))) void rtcsm_score_box_kernel_wide(
    GridView g, Params p, const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz) {
  score_box_body<kTCWide>(g, p, px, py, pz);
}

}  // namespace box
}  // namespace dliom

#endif  // DLIOM_CSRC_SCORE_BOX_H_
