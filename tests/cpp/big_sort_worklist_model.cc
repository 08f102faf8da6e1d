// Model of the CONTROL FLOW of big_sort_order (d-liom_amd/csrc/rothist_big.h): which segments of introsort's replay are
// partitioned by the whole workgroup, which go to LDS in batches, what the work list, the ring of wave_sort_arrangement
// and the depth limit allow -- i.e. everything that can make the device REFUSE a slice (the histogram then takes the
// host's path).  The partitions themselves are libstdc++'s (median of three + unguarded partition on (key, id) items),
// whose equality with the device's ballots is what wave_sort_model.cc and std_sort_model.cc check; here only the order
// std::sort leaves is compared once more, and the refusals are counted by cause:
//   depth   std::sort's depth limit on a segment too large for one lane's heap sort in LDS (> 4096 elements): documented
//   list    the work list of 256 segments overflowed                                        must not happen
//   ring    the ring of the wave-per-segment stage overflowed                               must not happen
//   stuck   a work list without an entry that fits LDS                                      must not happen
// Round 6's soaks found two refusals this model reproduces with the rules they met (argument "old"): the ring that held
// every segment ever queued, and the workgroup-wide branch chosen for a 1 203-element segment at the depth limit.
// Usage: big_sort_worklist_model [cases] [new|old] [arrays.txt|-] [seed] -> "... refusals: depth D, list 0, ring 0, stuck 0".
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

namespace {
constexpr int kMaxSlice = 4096;      // rotational.h
constexpr int kWorkListCap = 256;    // rothist_big.h
constexpr int kQueueCap = 1024;      // rotational_histogram.hip
constexpr size_t kBigLdsBytes = 150 * 1024;

struct Item { float key; int id; };
struct Seg { int first, last, depth; };
struct Counts { long depth = 0, list = 0, ring = 0, stuck = 0, mismatches = 0, lds_path = 0, hbm_path = 0, heap_sorts = 0; };
bool g_old_rules = false;

// libstdc++'s __move_median_to_first + __unguarded_partition on [first, last); returns the cut
int partition(std::vector<Item>& a, int first, int last) {
  const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
  const float ka = a[ia].key, kb = a[ib].key, kc = a[ic].key;
  int md;
  if (ka < kb) md = kb < kc ? ib : (ka < kc ? ic : ia);
  else md = ka < kc ? ia : (kb < kc ? ic : ib);
  std::swap(a[first], a[md]);
  const float pivot = a[first].key;
  int i = first + 1, j = last;
  for (;;) {
    while (a[i].key < pivot) ++i;
    --j;
    while (pivot < a[j].key) --j;
    if (!(i < j)) return i;
    std::swap(a[i], a[j]);
    ++i;
  }
}
void heap_sort(std::vector<Item>& a, int first, int last) {  // std::__partial_sort(first, last, last)
  auto cmp = [](const Item& x, const Item& y) { return x.key < y.key; };
  std::make_heap(a.begin() + first, a.begin() + last, cmp);
  std::sort_heap(a.begin() + first, a.begin() + last, cmp);
}
int tied_in(const std::vector<Item>& a, const std::vector<char>& tied, int first, int last) {
  int n = 0;
  for (int p = first; p < last; ++p) n += tied[a[p].id];
  return n;
}

// wave_sort_arrangement on the queued segments: level by level, a ring of `capacity` entries
bool wave_stage(std::vector<Item>& a, const std::vector<char>& tied, const std::vector<Seg>& initial, size_t capacity, Counts* c) {
  std::vector<Seg> ring(capacity);
  size_t reserved = 0, level_begin = 0;
  bool overflow = false;
  auto push = [&](const Seg& s) {
    const size_t slot = reserved++;
    const bool fits = g_old_rules ? slot < capacity : slot - level_begin < capacity;
    if (fits) ring[slot % capacity] = s; else overflow = true;
  };
  for (const Seg& s : initial) push(s);
  for (;;) {
    const size_t level_end = g_old_rules ? std::min(reserved, capacity) : reserved;
    if (level_begin >= level_end || overflow) break;
    for (size_t e = level_begin; e < level_end; ++e) {
      const Seg s = ring[e % capacity];
      if (s.depth == 0) { ++c->heap_sorts; heap_sort(a, s.first, s.last); continue; }
      const int cut = partition(a, s.first, s.last);
      if (cut - s.first > 16 && tied_in(a, tied, s.first, cut) >= 2) push(Seg{s.first, cut, s.depth - 1});
      if (s.last - cut > 16 && tied_in(a, tied, cut, s.last) >= 2) push(Seg{cut, s.last, s.depth - 1});
    }
    level_begin = level_end;
  }
  return !overflow;
}

// returns false when the device would refuse; `a` then holds garbage
bool big_sort_arrangement(std::vector<Item>& a, const std::vector<char>& tied, Counts* c) {
  const int m = static_cast<int>(a.size());
  int depth0 = 0;
  for (int v = m; v > 1; v >>= 1) ++depth0;
  depth0 *= 2;
  std::vector<Seg> wl;
  if (m > 16) wl.push_back(Seg{0, m, depth0});
  auto replace_by_children = [&](int pick, int cut) {  // thread 0 of the device: the entry leaves, its children join
    const Seg s = wl[pick];
    wl[pick] = wl.back();
    wl.pop_back();
    bool ok = true;
    const int cf[2] = {s.first, cut}, cl[2] = {cut, s.last};
    for (int k = 0; k < 2; ++k)
      if (cl[k] - cf[k] > 16 && tied_in(a, tied, cf[k], cl[k]) >= 2) {
        if (static_cast<int>(wl.size()) < kWorkListCap) wl.push_back(Seg{cf[k], cl[k], s.depth - 1}); else ok = false;
      }
    return ok;
  };
  const size_t mm = static_cast<size_t>(m);
  const size_t need = (mm + 2) / 2 * 8 + 2 * ((mm + 8 + 3) / 4 * 8) + (mm + 8 + 15) / 16 * 16 + 16 + (2 * mm / 17 + 64) * 8 + 16;
  if (need <= kBigLdsBytes && m < 65536) {
    // ---- the whole replay in LDS: workgroup-wide partitions for segments above kMaxSlice, the rest by waves
    ++c->lds_path;
    for (;;) {
      int pick = -1, best = kMaxSlice;
      for (int e = 0; e < static_cast<int>(wl.size()); ++e)
        if (wl[e].last - wl[e].first > best) { best = wl[e].last - wl[e].first; pick = e; }
      if (pick < 0) break;
      if (wl[pick].depth == 0) { ++c->depth; return false; }
      const int cut = partition(a, wl[pick].first, wl[pick].last);
      if (!replace_by_children(pick, cut)) { ++c->list; return false; }
    }
    if (!wave_stage(a, tied, wl, 2 * mm / 17 + 64, c)) { ++c->ring; return false; }
    return true;
  }
  // ---- arrays in HBM: the largest segment by the workgroup while much is left, batches through LDS otherwise
  ++c->hbm_path;
  for (long guard = 0; guard < (1 << 20) && !wl.empty(); ++guard) {
    int pick = -1, best = 0, total = 0;
    for (int e = 0; e < static_cast<int>(wl.size()); ++e) {
      const int len = wl[e].last - wl[e].first;
      total += len;
      if (len > best) { best = len; pick = e; }
    }
    bool by_workgroup = best > kMaxSlice || (total > kMaxSlice && best > kMaxSlice / 4 && (g_old_rules || wl[pick].depth > 0));
    const int mode = (by_workgroup && static_cast<int>(wl.size()) < kWorkListCap - 2) ? 1 : 2;
    if (mode == 2) {
      // entries in list order while they fit; the chosen ones move to the END of the list (as the device does it)
      int at = 0, taken = 0;
      const int n = static_cast<int>(wl.size());
      for (int e = 0; e < n - taken;) {
        const int len = wl[e].last - wl[e].first;
        if (len <= kMaxSlice && at + len <= kMaxSlice && taken < kWorkListCap) {
          std::swap(wl[e], wl[n - taken - 1]);
          at += len;
          ++taken;
        } else {
          ++e;
        }
      }
      if (taken == 0) { ++c->stuck; return false; }
      const std::vector<Seg> batch(wl.end() - taken, wl.end());
      wl.resize(wl.size() - static_cast<size_t>(taken));
      if (!wave_stage(a, tied, batch, kQueueCap, c)) { ++c->ring; return false; }
      continue;
    }
    if (wl[pick].depth == 0) { ++c->depth; return false; }
    const int cut = partition(a, wl[pick].first, wl[pick].last);
    if (!replace_by_children(pick, cut)) { ++c->list; return false; }
  }
  return true;
}

void check(const std::vector<float>& keys, Counts* c) {
  const int n = static_cast<int>(keys.size());
  std::vector<Item> a(n), sorted(n);
  for (int i = 0; i < n; ++i) a[i] = sorted[i] = Item{keys[i], i};
  std::stable_sort(sorted.begin(), sorted.end(), [](const Item& x, const Item& y) { return x.key < y.key; });
  std::vector<char> tied(n, 0);
  for (int j = 0; j + 1 < n; ++j)
    if (sorted[j].key == sorted[j + 1].key) tied[sorted[j].id] = tied[sorted[j + 1].id] = 1;
  if (!big_sort_arrangement(a, tied, c)) return;
  // a group of equal keys ends up in arrangement order (the final insertion sort is stable); elsewhere the keys decide
  std::vector<int> pos(n);
  for (int q = 0; q < n; ++q) pos[a[q].id] = q;
  std::vector<Item> got = sorted;
  std::stable_sort(got.begin(), got.end(), [&](const Item& x, const Item& y) { return x.key < y.key || (x.key == y.key && pos[x.id] < pos[y.id]); });
  std::vector<Item> want(n);
  for (int i = 0; i < n; ++i) want[i] = Item{keys[i], i};
  std::sort(want.begin(), want.end(), [](const Item& x, const Item& y) { return x.key < y.key; });
  for (int i = 0; i < n; ++i)
    if (want[i].id != got[i].id) { ++c->mismatches; return; }
}
}  // namespace

int main(int argc, char** argv) {
  const long cases = argc > 1 ? atol(argv[1]) : 400;
  g_old_rules = argc > 2 && std::strcmp(argv[2], "old") == 0;
  Counts c;
  // the soak's two findings, as their generators made them
  {
    std::vector<float> k(9716);
    for (int i = 0; i < 9716; ++i) k[i] = -static_cast<float>((i + 1) / 2);  // numpy: -arange(n) // 2 (floor division)
    check(k, &c);
  }
  const long after_first = c.ring;
  std::mt19937_64 rng(argc > 4 ? static_cast<unsigned long long>(atoll(argv[4])) : 20261001ull);
  auto uniform = [&](double lo, double hi) { return lo + (hi - lo) * (static_cast<double>(rng() >> 11) / 9007199254740992.0); };
  long lopsided_depth = 0;
  for (long t = 0; t < cases; ++t) {
    const int n = 4097 + static_cast<int>(rng() % 36000);
    const int kind = static_cast<int>(t % 6);
    std::vector<float> k(n);
    if (kind == 0) {  // sorted keys, a fifth overwritten by copies (the second finding's family)
      std::vector<double> u(n);
      for (double& v : u) v = uniform(-3, 3);
      std::sort(u.begin(), u.end());
      for (int r = 0; r < n / 5; ++r) u[rng() % n] = u[rng() % n];
      for (int i = 0; i < n; ++i) k[i] = static_cast<float>(u[i]);
    } else if (kind == 1) {  // descending runs of ties (the first finding's family)
      const int run = 1 + static_cast<int>(rng() % 8), shift = static_cast<int>(rng() % 2);
      for (int i = 0; i < n; ++i) k[i] = -static_cast<float>((i + shift) / run);
    } else if (kind == 2) {
      const int mod = 1 + static_cast<int>(rng() % 70);
      for (int i = 0; i < n; ++i) k[i] = static_cast<float>(i % mod);
    } else if (kind == 3) {
      const int div = 1 + static_cast<int>(rng() % 40);
      for (int i = 0; i < n; ++i) k[i] = static_cast<float>(rng() % std::max(1, n / div));
    } else if (kind == 4) {  // angles rounded to a grid: ties everywhere, random order
      for (int i = 0; i < n; ++i) k[i] = static_cast<float>(std::floor(uniform(-3.2, 3.2) * 2000.0) / 2000.0);
    } else {  // a handful of tied pairs among distinct keys
      for (int i = 0; i < n; ++i) k[i] = static_cast<float>(uniform(-3.2, 3.2));
      for (int r = 0; r < 3; ++r) k[rng() % n] = k[rng() % n];
    }
    const long depth_before = c.depth;
    check(k, &c);
    if (kind == 1 && c.depth != depth_before) ++lopsided_depth;
  }
  long file_arrays = 0, file_depth = 0;
  if (argc > 3) {  // arrays from a file (n, then n floats as hex words): the test writes the soak's second finding there
    FILE* f = std::fopen(argv[3], "r");
    int n;
    while (f != nullptr && std::fscanf(f, "%d", &n) == 1) {
      std::vector<float> k(n);
      for (int i = 0; i < n; ++i) {
        unsigned bits;
        if (std::fscanf(f, "%x", &bits) != 1) return 2;
        std::memcpy(&k[i], &bits, 4);
      }
      const long depth_before = c.depth;
      check(k, &c);
      ++file_arrays;
      file_depth += c.depth - depth_before;
    }
    if (f != nullptr) std::fclose(f);
  }
  std::printf("from the file: %ld arrays, %ld refused at the depth limit\n", file_arrays, file_depth);
  std::printf("arrays %ld (+ 1 fixed): %ld in LDS, %ld in HBM, heap sorts in LDS %ld; mismatches %ld; refusals: depth %ld (%ld of them descending tied runs), "
              "list %ld, ring %ld (%ld on the soak's 9 716 keys), stuck %ld\n",
              cases, c.lds_path, c.hbm_path, c.heap_sorts, c.mismatches, c.depth, lopsided_depth, c.list, c.ring, after_first, c.stuck);
  if (g_old_rules) return 0;
  return c.mismatches == 0 && c.list == 0 && c.ring == 0 && c.stuck == 0 ? 0 : 1;
}
