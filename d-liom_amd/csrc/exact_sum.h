// Exact PARALLEL replay of a SEQUENTIAL float sum  acc = ((acc0 + v[0]) + v[1]) + ...  for one workgroup of 1024
// threads; addends of either sign.  Used where the reference's result depends on the order of float additions:
// ComputeCentroid (rotational_scan_matcher.cc:52-59) and histogram(bucket) += value (:49).
// tests/cpp/exact_sum_model.h is this algorithm in plain C++ (pinned against the plain loop on 4000 random arrays:
// signed, monotone, hovering around zero and around powers of two, wild magnitudes, ties everywhere); the comments there
// carry the error bound.  In short: while the accumulator stays in one binade it is an integer counter and every addend
// a function {parity} -> {increment, parity} (ParityFn) that composes associatively; the real (double) prefix sums prove,
// per chunk of 16 addends, in which binade the accumulator is while it crosses the chunk ("safe" chunks, 94-98 % on
// LiDAR slices); a wave then walks the chunk functions 64 at a time with a scan and adds the values of the other chunks
// one after the other.  Anything not proven falls back to those sequential additions, which are right by definition.
#ifndef DLIOM_CSRC_EXACT_SUM_H_
#define DLIOM_CSRC_EXACT_SUM_H_

#include <hip/hip_runtime.h>

namespace dliom {
namespace exact_sum {

constexpr int kThreads = 1024;          // the workgroup size the block scans below are written for
constexpr int kChunk = 16;              // addends per chunk (32 needed more registers than a 1024-thread workgroup has: spills)
constexpr int kChunksPerBlock = 1024;   // chunks per super-block (one per thread); longer arrays are streamed
constexpr int kNoCode = 0x7fffffff;

struct Fn {
  int s0, s1;       // increment of the counter for parity-in 0 / 1
  unsigned p0, p1;  // parity out
};
__device__ __forceinline__ Fn identity_fn() { return Fn{0, 0, 0u, 1u}; }
__device__ __forceinline__ Fn compose(const Fn& a, const Fn& b) {  // a first, then b
  Fn r;
  r.s0 = a.s0 + (a.p0 ? b.s1 : b.s0);
  r.p0 = a.p0 ? b.p1 : b.p0;
  r.s1 = a.s1 + (a.p1 ? b.s1 : b.s0);
  r.p1 = a.p1 ? b.p1 : b.p0;
  return r;
}
__device__ __forceinline__ Fn shfl_up_fn(const Fn& f, int off) {
  Fn r;
  r.s0 = __shfl_up(f.s0, off, 64);
  r.s1 = __shfl_up(f.s1, off, 64);
  r.p0 = __shfl_up(f.p0, off, 64);
  r.p1 = __shfl_up(f.p1, off, 64);
  return r;
}
__device__ __forceinline__ Fn wave_inclusive_scan_fn(Fn f, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const Fn o = shfl_up_fn(f, off);
    if (lane >= off) f = compose(o, f);
  }
  return f;
}

// binade of a float: sign << 16 | biased exponent; kNoCode for zero, denormals, inf, nan
__device__ __forceinline__ int code_of(float a) {
  const unsigned u = __float_as_uint(a);
  const int be = static_cast<int>((u >> 23) & 0xffu);
  if (be == 0 || be == 255) return kNoCode;
  return static_cast<int>((u >> 31) << 16) | be;
}

// The Fn of adding x to an accumulator of binade `code`; *ok = false when x does not fit the model (never in a safe chunk)
__device__ __forceinline__ Fn element_fn(float x, int code, bool* ok) {
  const unsigned u = __float_as_uint(x);
  const unsigned mant = u & 0x7fffffu;
  const int bex = static_cast<int>((u >> 23) & 0xffu);
  if (bex == 255) {
    *ok = false;
    return identity_fn();
  }
  if (bex == 0 && mant == 0u) return identity_fn();
  const unsigned mx = bex == 0 ? mant : (mant | 0x800000u);
  const int ex = bex == 0 ? 1 : bex;
  const int sh = (code & 0xff) - ex;  // x / ulp = +-mx 2^-sh
  if (sh <= 0) {
    *ok = false;
    return identity_fn();
  }
  if (sh >= 25) return identity_fn();
  const bool negative = (u >> 31) != static_cast<unsigned>(code >> 16);  // sign of x relative to the accumulator's
  const unsigned q = mx >> sh, rem = mx & ((1u << sh) - 1u), half = 1u << (sh - 1);
  if (rem == half) {  // tie: of base and its neighbour away from zero the one that makes the counter even
    const int base = negative ? -static_cast<int>(q) : static_cast<int>(q);
    const int other = negative ? base - 1 : base + 1;
    Fn f;
    f.s0 = (base & 1) == 0 ? base : other;
    f.s1 = (base & 1) != 0 ? base : other;
    f.p0 = f.p1 = 0u;
    return f;
  }
  const int c = static_cast<int>(q + (rem > half ? 1u : 0u));
  const int inc = negative ? -c : c;
  return Fn{inc, inc, static_cast<unsigned>(c & 1), static_cast<unsigned>((c & 1) ^ 1)};
}

// compose(f, element_fn(x, code)) in straight-line code (16 of these per chunk are unrolled; branches cost registers):
// f = (s0, s1, p0, p1) is updated in place.  *ok = false when x does not fit the model (never in a safe chunk).
__device__ __forceinline__ void append_element(float x, int biased_exponent, unsigned sign, int& s0, int& s1, unsigned& p0,
                                               unsigned& p1, bool& ok) {
  const unsigned u = __float_as_uint(x);
  const unsigned mant = u & 0x7fffffu;
  const int bex = static_cast<int>((u >> 23) & 0xffu);
  const unsigned mx = bex == 0 ? mant : (mant | 0x800000u);
  const int sh = biased_exponent - max(bex, 1);  // x / ulp = +-mx 2^-sh
  ok = ok && bex != 255 && (sh > 0 || mx == 0u);
  const int shc = min(max(sh, 1), 25);  // from 25 on the counter does not move: q = 0 and rem = mx < half
  const unsigned q = mx >> shc, rem = mx & ((1u << shc) - 1u), half = 1u << (shc - 1);
  const bool tie = rem == half;
  const bool negative = (u >> 31) != sign;
  const unsigned c = q + (rem > half ? 1u : 0u);
  const int inc = negative ? -static_cast<int>(c) : static_cast<int>(c);
  const int base = negative ? -static_cast<int>(q) : static_cast<int>(q);
  const int other = negative ? base - 1 : base + 1;
  const int even = (base & 1) ? other : base, odd = (base & 1) ? base : other;  // increment for an even / odd counter at a tie
  s0 += tie ? (p0 ? odd : even) : inc;
  s1 += tie ? (p1 ? odd : even) : inc;
  const unsigned b = c & 1u;
  p0 = tie ? 0u : (p0 ^ b);
  p1 = tie ? 0u : (p1 ^ b);
}

// LDS scratch of one call: the chunk descriptors of a super-block for K arrays, the values of the chunks that will be
// added one by one, the scans' wave partials
constexpr int kStageSlots = 64;  // unsafe chunks per super-block whose values wait in LDS for the walk (others: from v)
template <int K>
struct Scratch {
  int4 desc[K][kChunksPerBlock];  // x: binade code, y / z: increment of the run from this chunk to its end for parity 0 / 1,
                                  // w: parity out (bits 0, 1) | last chunk of the run << 2 | (stage slot + 1) << 16
  float stage[K][kStageSlots][kChunk];
  double wave_part[K][kThreads / 64];
  int4 wave_agg[K][kThreads / 64];
  unsigned stage_count[K];
  float result[K];      // the float accumulators
  double carry_p[K];    // real prefix sums so far
  double carry_mx[K];   // largest |real prefix| so far
};

// Wave scans by DPP moves (row_shr inside the rows of 16 lanes, row_bcast:15 / :31 across them): a lane without a source
// gets `old`, the operation's identity.  (Until round 5 these were six __shfl_up steps of two ds_bpermute each, and the
// two block scans were 14 000 of the 34 000 cycles an array's classification took: the sixteen waves of the workgroup
// share four SIMDs, so the phase is bound by instruction issue, not by the shuffles' latency.)
template <int kCtrl, int kRowMask>
__device__ __forceinline__ int dpp_or(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, kCtrl, kRowMask, 0xf, false);
}
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_double_or_zero(double v) {
  const unsigned long long u = static_cast<unsigned long long>(__double_as_longlong(v));
  const unsigned lo = static_cast<unsigned>(dpp_or<kCtrl, kRowMask>(0, static_cast<int>(static_cast<unsigned>(u))));
  const unsigned hi = static_cast<unsigned>(dpp_or<kCtrl, kRowMask>(0, static_cast<int>(static_cast<unsigned>(u >> 32))));
  return __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo));
}
// inclusive scan over the wave's lanes: sums, or maxima of NON-NEGATIVE values (identity +0 either way).
// EVERY lane of the wave must be active at the call (update_dpp keeps `old` = +0 for a disabled source lane: a partial
// wave would silently add zeros where neighbours should have been); both callers run with all 1024 threads.  The sum
// variant adds +0.0 in lanes without a source, so a -0.0 prefix comes out as +0.0: the classification downstream
// compares magnitudes and parities of non-negative sums, for which the sign of a zero is irrelevant.
template <bool kMax>
__device__ __forceinline__ double wave_inclusive_scan_double(double v) {
#define DLIOM_ES_STEP(ctrl, mask)                                      \
  {                                                                    \
    const double o = dpp_double_or_zero<ctrl, mask>(v);                \
    v = kMax ? fmax(v, o) : v + o;                                     \
  }
  DLIOM_ES_STEP(0x111, 0xf)  // row_shr:1
  DLIOM_ES_STEP(0x112, 0xf)  // row_shr:2
  DLIOM_ES_STEP(0x114, 0xf)  // row_shr:4
  DLIOM_ES_STEP(0x118, 0xf)  // row_shr:8
  DLIOM_ES_STEP(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
  DLIOM_ES_STEP(0x143, 0xc)  // row_bcast:31 into rows 2 and 3
#undef DLIOM_ES_STEP
  return v;
}
// Inclusive block scans of one double per thread (1024 threads): sum / maximum of non-negative values.  `part`: 16 doubles of LDS.
template <bool kMax>
__device__ __forceinline__ double block_inclusive_scan(double v, double* part, double* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_inclusive_scan_double<kMax>(v);
  __syncthreads();  // `part` may still be read from an earlier scan
  if (lane == 63) part[wave] = v;
  __syncthreads();
  double before = 0.0, all = 0.0;
  for (int w = 0; w < kThreads / 64; ++w) {
    const double s = part[w];
    if (w < wave) before = kMax ? fmax(before, s) : before + s;
    all = kMax ? fmax(all, s) : all + s;
  }
  *total = all;
  return kMax ? fmax(before, v) : before + v;
}

// A run of safe chunks as seen from its first chunk: the composed function, where the run ends, and whether it ends
// inside the range this value covers (the segmented scan's flag)
struct Run {
  Fn f;
  int end;
  int closed;
};
__device__ __forceinline__ Run join(const Run& x, const Run& y) {  // x in front of y
  if (x.closed) return x;
  return Run{compose(x.f, y.f), y.end, y.closed};
}
// ... in three words, for the scan over the lanes: meta = parity out (bits 0, 1) | closed << 2 | end << 3; kNullRun: no run
struct RunW {
  int s0, s1;
  unsigned meta;
};
constexpr unsigned kNullRun = 0xFFFFFFFFu;
__device__ __forceinline__ RunW pack_run(const Run& r) {
  return RunW{r.f.s0, r.f.s1, r.f.p0 | (r.f.p1 << 1) | (static_cast<unsigned>(r.closed) << 2) | (static_cast<unsigned>(r.end) << 3)};
}
__device__ __forceinline__ Run unpack_run(const RunW& w) {
  return Run{Fn{w.s0, w.s1, w.meta & 1u, (w.meta >> 1) & 1u}, static_cast<int>(w.meta >> 3), static_cast<int>((w.meta >> 2) & 1u)};
}
__device__ __forceinline__ RunW join_w(const RunW& x, const RunW& y) {  // x in front of y; y may be kNullRun
  if (y.meta == kNullRun || (x.meta & 4u) != 0u) return x;
  const unsigned xp0 = x.meta & 1u, xp1 = (x.meta >> 1) & 1u, yp0 = y.meta & 1u, yp1 = (y.meta >> 1) & 1u;
  RunW r;
  r.s0 = x.s0 + (xp0 ? y.s1 : y.s0);
  r.s1 = x.s1 + (xp1 ? y.s1 : y.s0);
  r.meta = (y.meta & ~3u) | (xp0 ? yp1 : yp0) | ((xp1 ? yp1 : yp0) << 1);
  return r;
}
template <int kCtrl>
__device__ __forceinline__ RunW dpp_run(const RunW& r) {  // lanes without a source: kNullRun
  return RunW{dpp_or<kCtrl, 0xf>(0, r.s0), dpp_or<kCtrl, 0xf>(0, r.s1),
              static_cast<unsigned>(dpp_or<kCtrl, 0xf>(static_cast<int>(kNullRun), static_cast<int>(r.meta)))};
}
__device__ __forceinline__ RunW readlane_run(const RunW& r, int lane) {
  return RunW{__builtin_amdgcn_readlane(r.s0, lane), __builtin_amdgcn_readlane(r.s1, lane),
              static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(r.meta), lane))};
}
// every lane's run joined with the runs of the lanes behind it (a segmented SUFFIX scan): row_shl inside the rows of 16
// lanes, then the rows behind a lane's own from their first lanes (which hold their whole rows by then)
__device__ __forceinline__ RunW wave_suffix_scan_runs(RunW r, int lane) {
  r = join_w(r, dpp_run<0x101>(r));  // row_shl:1
  r = join_w(r, dpp_run<0x102>(r));  // row_shl:2
  r = join_w(r, dpp_run<0x104>(r));  // row_shl:4
  r = join_w(r, dpp_run<0x108>(r));  // row_shl:8
  const RunW a1 = readlane_run(r, 16), a2 = readlane_run(r, 32), a3 = readlane_run(r, 48);
  const RunW t1 = join_w(a2, a3), t0 = join_w(a1, t1);
  const int row = lane >> 4;
  RunW tail = a3;  // behind row 2
  if (row == 1) tail = t1;
  if (row == 0) tail = t0;
  if (row == 3) tail.meta = kNullRun;
  return join_w(r, tail);
}

// All 1024 threads call this; every thread returns with out[k] = the sequential float sum of v[k][0 .. n) started at acc0[k].
// v[k] may point to LDS or global memory (generic pointers); it must stay unchanged during the call.
#ifdef DLIOM_EXPERIMENTS
__device__ unsigned long long dbg_es[16];
#define DLIOM_ES_STAMP(i) if (threadIdx.x == 0 && blockIdx.x == 0) dbg_es[(i)] = __builtin_readcyclecounter()
#else
#define DLIOM_ES_STAMP(i)
#endif
template <int K>
__device__ void block_sequential_sums(const float* const (&v)[K], int n, const float (&acc0)[K], float (&out)[K], Scratch<K>& S) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Per-array state (real prefix, largest |prefix| so far, the float accumulator) lives in LDS and the arrays are
  // classified one after the other by the SAME code (a loop that is not unrolled): unrolled over K the two copies'
  // live ranges overlapped and a 1024-thread workgroup has 128 registers per thread -- 400 to 650 bytes of scratch per
  // lane, and the centroid of a 10 000-point floor slice took 46 us (round 4's first version; 10 us of work).
  __syncthreads();  // (S may lie over arrays the caller was still reading)
  DLIOM_ES_STAMP(0);
  if (tid < K) {
    float a0 = acc0[0];
#pragma unroll
    for (int k = 1; k < K; ++k)
      if (tid == k) a0 = acc0[k];
    S.result[tid] = a0;
    S.carry_p[tid] = static_cast<double>(a0);
    S.carry_mx[tid] = fabs(static_cast<double>(a0));
  }
  const int num_chunks = (n + kChunk - 1) / kChunk;
  for (int cb = 0; cb < num_chunks; cb += kChunksPerBlock) {  // uniform
    const int chunks_here = min(kChunksPerBlock, num_chunks - cb);
    const bool mine = tid < chunks_here;
    const int c = cb + tid;
    const int i0 = c * kChunk, i1 = min(n, i0 + kChunk);
    if (tid < K) S.stage_count[tid] = 0u;
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
      const float* vk = v[0];
#pragma unroll
      for (int kk = 1; kk < K; ++kk)
        if (k == kk) vk = v[kk];
      const double P = S.carry_p[k], Mx = S.carry_mx[k];
      // the chunk's addends, all loads in flight at once (a loop of load -> add pays the memory latency once per addend)
      float x[kChunk];
#pragma unroll
      for (int j = 0; j < kChunk; ++j) x[j] = (mine && i0 + j < i1) ? vk[i0 + j] : 0.f;
      double p = 0.0, lo = 0.0, hi = 0.0;
#pragma unroll
      for (int j = 0; j < kChunk; ++j) {
        p += static_cast<double>(x[j]);
        lo = fmin(lo, p);
        hi = fmax(hi, p);
      }
      DLIOM_ES_STAMP(10);
      double total, mtotal;
      const double incl = block_inclusive_scan<false>(p, S.wave_part[0], &total);
      const double Pc = P + (incl - p);  // real prefix in front of this thread's chunk
      const double reach = mine ? fmax(fabs(Pc + lo), fabs(Pc + hi)) : 0.0;
      const double mx = fmax(Mx, block_inclusive_scan<true>(reach, S.wave_part[0], &mtotal));
      if (tid == 0) {  // (everybody has read the old values: two barriers ago at the latest)
        S.carry_p[k] = P + total;
        S.carry_mx[k] = fmax(Mx, mtotal);
      }
      DLIOM_ES_STAMP(11);
      int code = kNoCode;
      Fn f = identity_fn();
      if (mine) {
        const double count = static_cast<double>(i1);
        const double E = 1.1 * count * 5.9604644775390625e-8 * mx + 1e-300;
        const double a = Pc + lo - E, b = Pc + hi + E;
        if (count <= 1048576.0 && ((a > 0.0 && b > 0.0) || (a < 0.0 && b < 0.0))) {
          const double m0 = fmin(fabs(a), fabs(b)), m1 = fmax(fabs(a), fabs(b));
          // binade exponents from the doubles' own exponent fields (both are normal, far from the double range's ends)
          const long long u0 = __double_as_longlong(m0), u1 = __double_as_longlong(m1);
          const int e0 = static_cast<int>((u0 >> 52) & 0x7ff) - 1023, e1 = static_cast<int>((u1 >> 52) & 0x7ff) - 1023;
          const bool power_of_two = (u0 & 0xfffffffffffffll) == 0ll;  // m0 == 2^e0: not strictly inside
          const int be = e0 + 127;
          if (e0 == e1 && !power_of_two && be >= 30 && be <= 250) {
            const int cd = ((a < 0.0 ? 1 : 0) << 16) | be;
            bool ok = true;
#pragma unroll
            for (int j = 0; j < kChunk; ++j)  // (padding zeros are the identity)
              append_element(x[j], be, a < 0.0 ? 1u : 0u, f.s0, f.s1, f.p0, f.p1, ok);
            if (ok) code = cd;
          }
        }
      }
      DLIOM_ES_STAMP(12);
      // chunks that will be added one value after the other: their values wait in LDS
      int slot = -1;
      if (mine && code == kNoCode) {
        const unsigned s = atomicAdd(&S.stage_count[k], 1u);
        if (s < static_cast<unsigned>(kStageSlots)) {
          slot = static_cast<int>(s);
#pragma unroll
          for (int j = 0; j < kChunk; j += 4)
            *reinterpret_cast<float4*>(&S.stage[k][slot][j]) = make_float4(x[j], x[j + 1], x[j + 2], x[j + 3]);
        }
      }
      // runs of consecutive chunks of one binade: a segmented suffix scan gives every chunk the composition from itself
      // to the end of its run, so that the walk below crosses a run in one step wherever it enters it
      S.desc[k][tid].x = mine ? code : kNoCode;
      __syncthreads();
      const int next_code = (tid + 1 < chunks_here) ? S.desc[k][tid + 1].x : kNoCode;
      Run r = unpack_run(wave_suffix_scan_runs(pack_run(Run{f, tid, (code == kNoCode || next_code != code) ? 1 : 0}), lane));
      if (lane == 0) {
        S.wave_agg[0][wave] = make_int4(r.f.s0, r.f.s1, static_cast<int>(r.f.p0 | (r.f.p1 << 1) | (static_cast<unsigned>(r.closed) << 2)), r.end);
      }
      __syncthreads();
      for (int w = wave + 1; w < kThreads / 64 && !r.closed; ++w) {
        const int4 g = S.wave_agg[0][w];
        r = join(r, Run{Fn{g.x, g.y, static_cast<unsigned>(g.z) & 1u, (static_cast<unsigned>(g.z) >> 1) & 1u}, g.w, (g.z >> 2) & 1});
      }
      if (mine)
        S.desc[k][tid] = make_int4(code, r.f.s0, r.f.s1,
                                   static_cast<int>(r.f.p0 | (r.f.p1 << 1) | (static_cast<unsigned>(r.end) << 2) |
                                                    (static_cast<unsigned>(slot + 1) << 16)));
      __syncthreads();  // (wave_agg and wave_part are shared by the arrays)
      DLIOM_ES_STAMP(1 + k);
    }
    __syncthreads();
    // the walk: wave k takes array k, every lane with the same numbers.  The chunk descriptors are held 64 at a time
    // across the lanes and the staged values of the chunks that are added one by one lie across the lanes as well (lane j:
    // the 16 values of stage slot j), so that a step reads registers (v_readlane), not LDS: a step was ~660 cycles of
    // dependent LDS reads, ~70 steps an array, 45 000 of the 113 000 cycles of a 10 000-point slice's centroid.
    if (wave < K) {
      float a = S.result[wave];
      const float* vp = v[0];
#pragma unroll
      for (int k = 1; k < K; ++k)
        if (wave == k) vp = v[k];
      float sv[kChunk];
      {
        const int staged = static_cast<int>(min(S.stage_count[wave], static_cast<unsigned>(kStageSlots)));
        static_assert(kStageSlots == 64, "one stage slot per lane");
#pragma unroll
        for (int j = 0; j < kChunk; j += 4) {
          float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
          if (lane < staged) q = *reinterpret_cast<const float4*>(&S.stage[wave][lane][j]);
          sv[j] = q.x;
          sv[j + 1] = q.y;
          sv[j + 2] = q.z;
          sv[j + 3] = q.w;
        }
      }
      int cc = 0, win = -64;
      int4 dwin = make_int4(kNoCode, 0, 0, 0);
#ifdef DLIOM_EXPERIMENTS
      int dbg_steps = 0, dbg_seq = 0, dbg_far = 0;
#endif
      while (cc < chunks_here) {
#ifdef DLIOM_EXPERIMENTS
        ++dbg_steps;
#endif
        if (cc >= win + 64) {  // (the walk only moves forward)
          win = cc;
          dwin = cc + lane < chunks_here ? S.desc[wave][cc + lane] : make_int4(kNoCode, 0, 0, 0);
        }
        const int j = cc - win;
        const int dx = __builtin_amdgcn_readlane(dwin.x, j), dy = __builtin_amdgcn_readlane(dwin.y, j);
        const int dz = __builtin_amdgcn_readlane(dwin.z, j), dw = __builtin_amdgcn_readlane(dwin.w, j);
        const unsigned bits = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(__float_as_uint(a))));
        if (dx != kNoCode && dx == code_of(__uint_as_float(bits))) {
          const int kcount = static_cast<int>((bits & 0x7fffffu) | 0x800000u);
          const int k2 = kcount + ((kcount & 1) ? dz : dy);
          if (k2 >= (1 << 23) && k2 < (1 << 24)) {
            a = __uint_as_float((bits & 0xff800000u) | (static_cast<unsigned>(k2) & 0x7fffffu));
            cc = static_cast<int>((static_cast<unsigned>(dw) >> 2) & 0x3fffu) + 1;
            continue;
          }
        }
        // this chunk's values one after the other
        const int j0 = (cb + cc) * kChunk, j1 = min(n, j0 + kChunk);
        const int slot = static_cast<int>(static_cast<unsigned>(dw) >> 16) - 1;
#ifdef DLIOM_EXPERIMENTS
        ++dbg_seq;
        if (slot < 0) ++dbg_far;
#endif
        float x[kChunk];
        if (slot >= 0) {
#pragma unroll
          for (int t = 0; t < kChunk; ++t)
            x[t] = __uint_as_float(static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(sv[t])), slot)));
        } else {
#pragma unroll
          for (int t = 0; t < kChunk; ++t) x[t] = j0 + t < j1 ? vp[j0 + t] : 0.f;
        }
        if (j1 - j0 == kChunk) {
#pragma unroll
          for (int t = 0; t < kChunk; ++t) a += x[t];
        } else {
#pragma unroll
          for (int t = 0; t < kChunk; ++t)
            if (j0 + t < j1) a += x[t];  // (a padding +0 would turn an accumulator of -0 into +0)
        }
        cc += 1;
      }
      if (lane == 0) S.result[wave] = a;
#ifdef DLIOM_EXPERIMENTS
      if (lane == 0 && blockIdx.x == 0 && wave == 0) {
        dbg_es[13] = S.stage_count[0];
        dbg_es[14] = static_cast<unsigned long long>(dbg_steps);
        dbg_es[15] = static_cast<unsigned long long>(dbg_seq) | (static_cast<unsigned long long>(dbg_far) << 32);
      }
#endif
    }
    __syncthreads();
    DLIOM_ES_STAMP(8);
  }
#pragma unroll
  for (int k = 0; k < K; ++k) out[k] = S.result[k];
  __syncthreads();  // (the caller may reuse S)
}

}  // namespace exact_sum
}  // namespace dliom

#endif  // DLIOM_CSRC_EXACT_SUM_H_
