// exact_sum_model.h (the parallel replay of a sequential float sum, as the device does it) against the plain loop on random
// arrays: signed coordinates, histogram values, sweeps, sums hovering around zero / powers of two, wild magnitudes, ties.
// Usage: exact_sum_model_test -> "mismatches: 0 of 4000; ..."
#include "exact_sum_model.h"
#include <cstdio>
#include <random>
using namespace exact_sum_model;
static float seq(const std::vector<float>& v, float a0) { volatile float a = a0; for (float x : v) a = a + x; return a; }
int main() {
  std::mt19937 rng(7);
  long bad = 0, cases = 0, unsafe = 0, chunks = 0, seqadds = 0, total = 0;
  for (int it = 0; it < 4000; ++it) {
    int n = (it % 7 == 0) ? rng() % 200000 : rng() % 20000;
    int kind = it % 10;
    std::vector<float> v(n);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::normal_distribution<float> N(0.f, 1.f);
    for (int i = 0; i < n; ++i) {
      float x;
      switch (kind) {
        case 0: x = 30.f * U(rng); break;                                   // signed coordinates
        case 1: x = std::fabs(U(rng)); break;                               // histogram values
        case 2: x = 20.f * std::cos(6.2831853f * i / (n + 1)) + 0.01f * N(rng); break;  // a sweep
        case 3: x = (i % 2 ? 1.f : -1.f) * (1.f + 1e-3f * U(rng)); break;  // hovering around zero
        case 4: x = std::ldexp(U(rng), (int)(rng() % 40) - 20); break;     // wild magnitudes
        case 5: x = -std::fabs(U(rng)) - 1.8f; break;                       // z of a floor
        case 6: x = (rng() % 4 == 0) ? 0.f : 0.5f; break;                  // ties: halves
        case 7: x = std::ldexp(1.f, -(int)(rng() % 30)); break;            // powers of two (ties everywhere)
        case 8: x = (rng() % 3 == 0 ? -1.f : 1.f) * std::ldexp(1.f, -(int)(rng() % 26)); break;
        default: x = 1024.f + U(rng); break;                                // hovering near 2^k boundaries
      }
      v[i] = x;
    }
    float a0 = (it % 3 == 0) ? 0.f : 100.f * U(rng);
    long u = 0, s = 0;
    float got = exact_sequential_sum(v.data(), n, a0, &u, &s), want = seq(v, a0);
    unsafe += u; chunks += (n + kChunk - 1) / kChunk; seqadds += s; total += n;
    ++cases;
    if (bits_of(got) != bits_of(want)) { ++bad; if (bad < 10) printf("MISMATCH kind %d n %d got %a want %a\n", kind, n, got, want); }
  }
  printf("mismatches: %ld of %ld; unsafe chunks %ld of %ld; sequential adds %ld of %ld\n", bad, cases, unsafe, chunks, seqadds, total);
  return bad != 0;
}
