// Model (plain C++) of the exact PARALLEL replay of a SEQUENTIAL float sum that rotational_histogram.hip uses for
// ComputeCentroid (rotational_scan_matcher.cc:52-59: sum += point, one after the other, signed addends) and for
// histogram(bucket) += value (:49).  Float addition is not associative, but while the accumulator stays inside one binade
// (sign s, exponent e: |acc| in [2^e, 2^(e+1)), ulp U = 2^(e-23)) it is an integer counter k = |acc| / U, and adding x
// changes it by round(s x / U) with ties to even -- a function {parity of k} -> {increment, parity out} (ParityFn),
// which composes associatively.  The real (double) prefix sums locate the accumulator within a rigorous error bound:
//   chunk c (16 addends) is SAFE in binade (s, e) when [P_c + lo_c - E_c, P_c + hi_c + E_c] lies strictly inside it, where
//   P_c is the real prefix, lo/hi the extremes of the partial sums inside the chunk and
//   E_c >= |float accumulator - real prefix| anywhere up to the end of chunk c: every addition rounds by at most half an
//   ulp of its result, i.e. 2^-24 (|S_i| + err), so err <= i 2^-24 Mx / (1 - i 2^-24) with Mx the largest |real prefix|
//   so far (i <= 2^20 additions: factor 1.07; 1.1 is used, plus the double prefix's own rounding, 2^-52 i Mx).
// Safe chunks get their ParityFn in parallel; a walk then applies runs of safe chunks of the accumulator's current binade
// at once (a wave scan on the device) and adds the others one value after the other in float.  Every decision that is
// not provably right falls back to the sequential float additions, which are right by definition.
#ifndef TESTS_CPP_EXACT_SUM_MODEL_H_
#define TESTS_CPP_EXACT_SUM_MODEL_H_
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace exact_sum_model {

struct ParityFn {
  int s0, s1;        // increment of the counter for parity-in 0 / 1
  unsigned p0, p1;   // parity out
};
inline ParityFn identity_fn() { return {0, 0, 0u, 1u}; }
inline ParityFn compose(const ParityFn& a, const ParityFn& b) {  // a first, then b
  ParityFn r;
  r.s0 = a.s0 + (a.p0 ? b.s1 : b.s0);
  r.p0 = a.p0 ? b.p1 : b.p0;
  r.s1 = a.s1 + (a.p1 ? b.s1 : b.s0);
  r.p1 = a.p1 ? b.p1 : b.p0;
  return r;
}
inline uint32_t bits_of(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float float_of(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

constexpr int kChunk = 16;
constexpr int kNoCode = 0x7fffffff;
// binade code of a float: sign << 16 | (biased exponent); kNoCode for 0, denormals, inf, nan
inline int code_of(float a) {
  const uint32_t u = bits_of(a);
  const int be = static_cast<int>((u >> 23) & 0xffu);
  if (be == 0 || be == 255) return kNoCode;
  return static_cast<int>((u >> 31) << 16) | be;
}
// The ParityFn of adding x to an accumulator of binade `code`; *ok = false when x is too large for that (cannot happen
// in a safe chunk: accumulator and result in the binade imply |x| < 2^e)
inline ParityFn element_fn(float x, int code, bool* ok) {
  const uint32_t u = bits_of(x);
  const uint32_t mant = u & 0x7fffffu;
  const int bex = static_cast<int>((u >> 23) & 0xffu);
  if (bex == 255) { *ok = false; return identity_fn(); }
  if (bex == 0 && mant == 0u) return identity_fn();              // +-0
  const uint32_t mx = bex == 0 ? mant : (mant | 0x800000u);       // x = +-mx 2^(ex - 23)
  const int ex = bex == 0 ? 1 : bex;                              // biased exponent, denormals share exponent 1
  const int e = code & 0xff;
  const int sh = e - ex;                                          // x / U = +-mx 2^-sh
  if (sh <= 0) { *ok = false; return identity_fn(); }
  if (sh >= 25) return identity_fn();                             // |x / U| < 1/2: the counter does not move
  const bool negative = ((u >> 31) != static_cast<uint32_t>(code >> 16));  // sign of s x
  const uint32_t q = mx >> sh, rem = mx & ((1u << sh) - 1u), half = 1u << (sh - 1);
  if (rem == half) {  // tie: of base and base +- 1 the one that makes the counter even
    const int base = negative ? -static_cast<int>(q) : static_cast<int>(q);
    const int other = negative ? base - 1 : base + 1;
    ParityFn f;
    f.s0 = (base & 1) == 0 ? base : other;
    f.s1 = (base & 1) != 0 ? base : other;
    f.p0 = f.p1 = 0u;
    return f;
  }
  const int c = static_cast<int>(q + (rem > half ? 1u : 0u));
  const int inc = negative ? -c : c;
  return {inc, inc, static_cast<unsigned>(c & 1), static_cast<unsigned>((c & 1) ^ 1)};
}

struct ChunkInfo {
  int code;      // binade the chunk is safe in, or kNoCode
  ParityFn fn;
};

// All chunk descriptors ("in parallel": each depends only on the double prefix scan).
inline void classify(const float* v, int n, float acc0, std::vector<ChunkInfo>* out, long* unsafe = nullptr) {
  const int nc = (n + kChunk - 1) / kChunk;
  out->assign(nc, ChunkInfo{kNoCode, identity_fn()});
  std::vector<double> sum(nc), lo(nc), hi(nc);
  for (int c = 0; c < nc; ++c) {  // pass 1: per chunk
    double p = 0, l = 0, h = 0;
    for (int i = c * kChunk; i < std::min(n, (c + 1) * kChunk); ++i) {
      p += static_cast<double>(v[i]);
      l = std::min(l, p);
      h = std::max(h, p);
    }
    sum[c] = p; lo[c] = l; hi[c] = h;
  }
  double P = static_cast<double>(acc0), Mx = std::fabs(P);  // scan
  for (int c = 0; c < nc; ++c) {
    Mx = std::max(Mx, std::max(std::fabs(P + lo[c]), std::fabs(P + hi[c])));
    const double count = static_cast<double>(std::min(n, (c + 1) * kChunk));
    const double E = 1.1 * count * 5.9604644775390625e-8 * Mx + 1e-300;
    const double a = P + lo[c] - E, b = P + hi[c] + E;
    P += sum[c];
    if (!(count <= 1048576.0)) continue;  // the bound's factor is written for <= 2^20 additions
    if (!((a > 0 && b > 0) || (a < 0 && b < 0))) continue;
    const double m0 = std::min(std::fabs(a), std::fabs(b)), m1 = std::max(std::fabs(a), std::fabs(b));
    int e0, e1;
    std::frexp(m0, &e0);  // m0 = f 2^e0, f in [0.5, 1): binade exponent e0 - 1
    std::frexp(m1, &e1);
    if (e0 != e1) continue;
    const int be = e0 - 1 + 127;
    if (be < 30 || be > 250) continue;
    // strictly inside: m0 > 2^(e0-1) and m1 < 2^e0 (frexp puts an exact power of two at f = 0.5: exclude it)
    if (m0 == std::ldexp(1.0, e0 - 1)) continue;
    const int code = ((a < 0 ? 1 : 0) << 16) | be;
    ParityFn f = identity_fn();
    bool ok = true;
    for (int i = c * kChunk; i < std::min(n, (c + 1) * kChunk); ++i) f = compose(f, element_fn(v[i], code, &ok));
    if (ok) (*out)[c] = ChunkInfo{code, f};
  }
  if (unsafe != nullptr)
    for (int c = 0; c < nc; ++c) *unsafe += (*out)[c].code == kNoCode;
}

inline float exact_sequential_sum(const float* v, int n, float acc0 = 0.f, long* unsafe = nullptr, long* sequential_adds = nullptr) {
  std::vector<ChunkInfo> info;
  classify(v, n, acc0, &info, unsafe);
  const int nc = static_cast<int>(info.size());
  float acc = acc0;
  int c = 0;
  while (c < nc) {
    const int code = code_of(acc);
    bool applied = false;
    if (code != kNoCode && info[c].code == code) {
      // a run of chunks of the accumulator's binade (the device takes up to 64 at a time: a wave scan)
      ParityFn f = identity_fn();
      int r = c;
      while (r < nc && r < c + 64 && info[r].code == code) f = compose(f, info[r++].fn);
      const uint32_t u = bits_of(acc);
      const int64_t k = static_cast<int64_t>((u & 0x7fffffu) | 0x800000u);
      const int64_t k2 = k + ((k & 1) ? f.s1 : f.s0);
      if (k2 >= (1 << 23) && k2 < (1 << 24)) {
        acc = float_of((u & 0xff800000u) | (static_cast<uint32_t>(k2) & 0x7fffffu));
        c = r;
        applied = true;
      }
    }
    if (!applied) {
      for (int i = c * kChunk; i < std::min(n, (c + 1) * kChunk); ++i) {
        volatile float t = acc + v[i];
        acc = t;
        if (sequential_adds != nullptr) ++*sequential_adds;
      }
      ++c;
    }
  }
  return acc;
}

}  // namespace exact_sum_model
#endif  // TESTS_CPP_EXACT_SUM_MODEL_H_
