// Model (plain C++) of the blocked walk rothist_big.h uses to mark the nodes of the path 0 -> next(0) -> next(next(0)) ...
// of AddPointCloudSliceToHistogram's `last_point` (rotational_scan_matcher.cc:61-92) when the chain's arrays are in LDS:
// next(i) > i, so the path only moves forward, and instead of squaring all m pointers ceil(log2 m) times (pointer
// doubling: 15 levels over 10 000 positions, 50 us) the positions are cut into blocks of 64:
//   1. one thread per block, positions from the last to the first: x1(i) = the first path node at or behind the end of
//      the block (next(i) if that already is, else x1(next(i)) -- a later position of the same block);
//   2. one thread hops from block to block with x1 and notes where the path enters each block;
//   3. one thread per block marks from its block's entry, hopping with next.
// (blocked_walk_three_levels is the first version -- thread blocks of ceil(m / 1024), a wave level x2 in between --,
// equally right and slower on the device; kept as a second witness.)
// Checked against the plain walk on random forward-pointing arrays: steps of one, short jumps, long jumps, jumps past the
// end, every size around the block boundaries.  Usage: chain_walk_model [cases] -> "mismatches: 0 of N".
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

namespace {
constexpr int kThreads = 1024, kWave = 64;

std::vector<char> plain_walk(const std::vector<int>& next, int m) {
  std::vector<char> mark(m + 1, 0);
  for (int i = 0; i < m; i = next[i]) mark[i] = 1;
  return mark;
}

std::vector<char> blocked_walk(const std::vector<int>& next, int m) {
  constexpr int kBlock = 64;
  const int blocks = (m + kBlock - 1) / kBlock;
  std::vector<int> x1(m + 1, m), entry(blocks + 1, m);
  for (int b = 0; b < blocks; ++b) {
    const int lo = b * kBlock, hi = std::min(m, lo + kBlock);
    for (int i = hi - 1; i >= lo; --i) x1[i] = next[i] >= hi ? next[i] : x1[next[i]];
  }
  for (int e = 0; e < m; e = x1[e]) entry[e / kBlock] = e;
  std::vector<char> mark(m + 1, 0);
  for (int b = 0; b < blocks; ++b) {
    const int hi = std::min(m, b * kBlock + kBlock);
    for (int i = entry[b]; i < hi; i = next[i]) mark[i] = 1;
  }
  return mark;
}

std::vector<char> blocked_walk_three_levels(const std::vector<int>& next, int m) {
  const int per = (m + kThreads - 1) / kThreads;
  auto lo_of = [&](int t) { return std::min(m, t * per); };
  auto hi_of = [&](int t) { return std::min(m, lo_of(t) + per); };
  std::vector<int> x1(m + 1, m), x2(m + 1, m);
  // 1. per thread, descending
  for (int t = 0; t < kThreads; ++t)
    for (int i = hi_of(t) - 1; i >= lo_of(t); --i) x1[i] = next[i] >= hi_of(t) ? next[i] : x1[next[i]];
  // 2. per wave, thread blocks descending
  const int waves = kThreads / kWave;
  auto wave_hi = [&](int w) { return hi_of(w * kWave + kWave - 1); };
  for (int w = 0; w < waves; ++w)
    for (int b = kWave - 1; b >= 0; --b) {
      const int t = w * kWave + b;
      for (int i = lo_of(t); i < hi_of(t); ++i) x2[i] = x1[i] >= wave_hi(w) ? x1[i] : x2[x1[i]];
    }
  // 3. the waves' entries (m: the path does not visit the wave's range)
  std::vector<int> entry_w(waves, m);
  {
    int e = 0;  // the current path node
    for (int w = 0; w < waves && e < m; ++w) {
      if (e < wave_hi(w)) {  // (e >= the range's start: the previous hop ended at or behind the previous range's end)
        entry_w[w] = e;
        e = x2[e];
      }
    }
  }
  // 4. per wave: the entries of its thread blocks
  std::vector<int> entry_t(kThreads, m);
  for (int w = 0; w < waves; ++w) {
    int e = entry_w[w];
    while (e < wave_hi(w)) {  // (e = m ends it: wave_hi <= m)
      entry_t[e / per] = e;
      e = x1[e];
    }
  }
  // 5. per thread: marks
  std::vector<char> mark(m + 1, 0);
  for (int t = 0; t < kThreads; ++t)
    for (int i = entry_t[t]; i < hi_of(t); i = next[i]) mark[i] = 1;
  return mark;
}
}  // namespace

int main(int argc, char** argv) {
  const int cases = argc > 1 ? std::atoi(argv[1]) : 400;
  std::mt19937 rng(7);
  long bad = 0;
  const int sizes[] = {1, 2, 63, 64, 65, 1023, 1024, 1025, 2047, 2048, 2049, 4097, 8191, 10462, 14460, 15356, 16384, 16385, 20090, 30000};
  for (int c = 0; c < cases; ++c) {
    const int m = c < 60 ? sizes[c % 20] : 1 + static_cast<int>(rng() % 30000);
    const int kind = c % 5;
    std::vector<int> next(m + 1, m);
    for (int i = 0; i < m; ++i) {
      int step;
      if (kind == 0) step = 1;
      else if (kind == 1) step = 1 + static_cast<int>(rng() % 3);
      else if (kind == 2) step = (rng() % 10 == 0) ? 1 + static_cast<int>(rng() % 2000) : 1;
      else if (kind == 3) step = 1 + static_cast<int>(rng() % (m + 5));
      else step = (rng() % 50 == 0) ? m : 1 + static_cast<int>(rng() % 40);
      next[i] = std::min(m, i + step);
    }
    const std::vector<char> want = plain_walk(next, m), got = blocked_walk(next, m), got3 = blocked_walk_three_levels(next, m);
    bool same = true;
    for (int i = 0; i < m; ++i) same = same && want[i] == got[i] && want[i] == got3[i];
    if (!same) {
      ++bad;
      std::printf("case %d m %d kind %d differs\n", c, m, kind);
    }
  }
  std::printf("mismatches: %ld of %d\n", bad, cases);
  return bad == 0 ? 0 : 1;
}
