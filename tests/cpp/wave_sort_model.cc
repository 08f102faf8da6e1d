// Model of wave_sort_arrangement (d-liom_amd/csrc/rotational_histogram.hip): introsort's partitions of the segments that
// still hold two tied elements, one "wave" (64 lanes, emulated by loops: ballots become bit masks, lane ranks become
// prefix counts) per segment, the segments served level by level from a queue; then a stable sort by (key, arrangement
// position).  Checked against this machine's std::sort on arrays full of ties and on the files of slice angles the other
// models use.  The queue is the device's RING of 2 n / 17 + 64 entries (1024 up to 4096 keys): what is alive at a time --
// the rest of the level being served and the children appended so far, disjoint segments of more than 16 elements each --
// fits; what was EVER queued need not (paths of lopsided partitions: descending keys in tied runs, the soak's finding of
// round 6).  Usage: wave_sort_model [cases] [slices.txt] -> "mismatches: 0 of N ... ring overflows 0".
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

struct Item { unsigned key; int id; };
struct RefItem { float key; int id; bool operator<(const RefItem& o) const { return key < o.key; } };
static unsigned ordered_bits(float f) {
  if (f == 0.f) f = 0.f;
  unsigned u; std::memcpy(&u, &f, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static void heap_adjust(Item* first, int hole, int len, Item value) {
  const int top = hole; int child = hole;
  while (child < (len - 1) / 2) { child = 2 * (child + 1); if (first[child].key < first[child - 1].key) --child; first[hole] = first[child]; hole = child; }
  if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); first[hole] = first[child - 1]; hole = child - 1; }
  int parent = (hole - 1) / 2;
  while (hole > top && first[parent].key < value.key) { first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2; }
  first[hole] = value;
}
static void heap_sort(Item* first, int len) {
  if (len >= 2) for (int parent = (len - 2) / 2;; --parent) { heap_adjust(first, parent, len, first[parent]); if (parent == 0) break; }
  for (int last = len; last > 1;) { --last; const Item v = first[last]; first[last] = first[0]; heap_adjust(first, 0, last, v); }
}
struct Seg { int first, last, depth; };
static long g_levels = 0, g_heaps = 0, g_overflows = 0, g_most_alive_permille = 0, g_ever_above_capacity = 0;
struct Ring {  // Queue of rotational_histogram.hip: queue_push / the reads of wave_sort_arrangement
  std::vector<Seg> seg;
  size_t reserved = 0, level_begin = 0;
  explicit Ring(size_t capacity) : seg(capacity) {}
  void push_back(const Seg& s) {
    const size_t slot = reserved++;
    if (slot - level_begin < seg.size()) seg[slot % seg.size()] = s; else ++g_overflows;
    const long permille = static_cast<long>(1000 * (slot - level_begin + 1) / seg.size());
    if (permille > g_most_alive_permille) g_most_alive_permille = permille;
  }
  const Seg& at(size_t e) const { return seg[e % seg.size()]; }
};

static void wave_partition(std::vector<Item>& a, const std::vector<char>& tied, const Seg& s, std::vector<int>& tmp_l, std::vector<int>& tmp_r,
                           Ring* queue) {
  const int first = s.first, last = s.last, depth = s.depth;
  if (depth == 0) { ++g_heaps; heap_sort(a.data() + first, last - first); return; }
  {  // lane 0
    const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
    const unsigned ka = a[ia].key, kb = a[ib].key, kc = a[ic].key;
    int md;
    if (ka < kb) { if (kb < kc) md = ib; else if (ka < kc) md = ic; else md = ia; }
    else if (ka < kc) md = ia; else if (kb < kc) md = ic; else md = ib;
    std::swap(a[first], a[md]);
  }
  const unsigned pivot = a[first].key;
  int* stops_l = tmp_l.data() + first + 1;
  int* stops_r = tmp_r.data() + first + 1;
  int cnt_l = 0, cnt_r = 0;
  for (int base = first + 1; base < last; base += 64) {
    uint64_t ml = 0, mr = 0;
    for (int lane = 0; lane < 64; ++lane) {
      const int p = base + lane;
      const bool in = p < last;
      const unsigned x = in ? a[p].key : 0u;
      if (in && !(x < pivot)) ml |= 1ull << lane;
      if (in && !(pivot < x)) mr |= 1ull << lane;
    }
    for (int lane = 0; lane < 64; ++lane) {
      const int p = base + lane;
      if (ml >> lane & 1) stops_l[cnt_l + __builtin_popcountll(ml & ((1ull << lane) - 1))] = p;
      if (mr >> lane & 1) stops_r[cnt_r + __builtin_popcountll(mr & ((1ull << lane) - 1))] = p;
    }
    cnt_l += __builtin_popcountll(ml);
    cnt_r += __builtin_popcountll(mr);
  }
  const int lim = std::min(cnt_l, cnt_r);
  int K = lim;
  for (int k0 = 0; k0 < lim; k0 += 64) {
    uint64_t mv = 0;
    for (int lane = 0; lane < 64; ++lane) {
      const int k = k0 + lane;
      if (k < lim && stops_l[k] < stops_r[cnt_r - 1 - k]) mv |= 1ull << lane;
    }
    if (mv != ~0ull) { K = k0 + __builtin_ctzll(~mv); break; }
  }
  for (int k = 0; k < K; ++k) std::swap(a[stops_l[k]], a[stops_r[cnt_r - 1 - k]]);
  int cut = 1 << 30;
  if (K < cnt_l) cut = stops_l[K];
  if (K > 0) cut = std::min(cut, stops_r[cnt_r - K]);
  int tied_l = 0, tied_r = 0;
  for (int p = first; p < last; ++p) { if (tied[a[p].id]) { if (p < cut) ++tied_l; else ++tied_r; } }
  if (cut - first > 16 && tied_l >= 2) queue->push_back(Seg{first, cut, depth - 1});
  if (last - cut > 16 && tied_r >= 2) queue->push_back(Seg{cut, last, depth - 1});
}

static std::vector<int> model_order(const std::vector<float>& keys) {
  const int n = static_cast<int>(keys.size());
  std::vector<Item> sorted(n), a(n);
  for (int i = 0; i < n; ++i) sorted[i] = a[i] = Item{ordered_bits(keys[i]), i};
  std::stable_sort(sorted.begin(), sorted.end(), [](const Item& x, const Item& y) { return x.key < y.key; });
  std::vector<char> tied(n, 0);
  bool any = false;
  for (int j = 0; j + 1 < n; ++j) if (sorted[j].key == sorted[j + 1].key) tied[sorted[j].id] = tied[sorted[j + 1].id] = 1, any = true;
  std::vector<int> order(n);
  if (!any) { for (int j = 0; j < n; ++j) order[j] = sorted[j].id; return order; }
  Ring queue(n <= 4096 ? 1024 : 2 * static_cast<size_t>(n) / 17 + 64);  // kQueueCap / rothist_big.h's queue_entries
  std::vector<int> tmp_l(n + 8), tmp_r(n + 8);
  if (n > 16) { int depth = 0; for (int v = n; v > 1; v >>= 1) ++depth; queue.push_back(Seg{0, n, 2 * depth}); }
  size_t level_begin = 0;
  for (;;) {
    const size_t level_end = queue.reserved;
    queue.level_begin = level_begin;
    if (level_begin >= level_end) break;
    ++g_levels;
    for (size_t e = level_begin; e < level_end; ++e) { const Seg s = queue.at(e); wave_partition(a, tied, s, tmp_l, tmp_r, &queue); }
    level_begin = level_end;
  }
  if (queue.reserved > queue.seg.size()) ++g_ever_above_capacity;
  std::vector<std::pair<std::pair<unsigned, int>, int>> fin(n);
  for (int q = 0; q < n; ++q) fin[q] = {{a[q].key, q}, a[q].id};
  std::sort(fin.begin(), fin.end());
  for (int j = 0; j < n; ++j) order[j] = fin[j].second;
  return order;
}
static bool check(const std::vector<float>& keys) {
  const int n = static_cast<int>(keys.size());
  std::vector<RefItem> want(n);
  for (int i = 0; i < n; ++i) want[i] = RefItem{keys[i], i};
  std::sort(want.begin(), want.end());
  const std::vector<int> got = model_order(keys);
  for (int i = 0; i < n; ++i) if (want[i].id != got[i]) return false;
  return true;
}
int main(int argc, char** argv) {
  const long cases = argc > 1 ? atol(argv[1]) : 3000;
  std::mt19937 rng(4321);
  long bad = 0;
  for (long c = 0; c < cases; ++c) {
    const int sizes[] = {0, 1, 2, 15, 16, 17, 18, 31, 33, 64, 65, 100, 257, 700, 1000, 2048, 4096};
    int n = sizes[c % 17];
    if (c % 5 == 0) n = static_cast<int>(rng() % 4097);
    const int kind = static_cast<int>(rng() % 7);
    std::vector<float> k(n);
    for (int i = 0; i < n; ++i) switch (kind) {
      case 0: k[i] = static_cast<float>(rng() % 3); break;
      case 1: k[i] = static_cast<float>(rng() % 50); break;
      case 2: k[i] = static_cast<float>(rng() % (n / 2 + 1)); break;
      case 3: k[i] = static_cast<float>(rng()) * 1e-6f; break;
      case 4: k[i] = static_cast<float>(i / 7); break;
      case 5: k[i] = static_cast<float>((n - i) / 3); break;
      default: k[i] = 5.f; break;
    }
    if (kind == 3 && n > 8) for (int r = 0; r < 3; ++r) k[rng() % n] = k[rng() % n];
    if (!check(k)) { ++bad; if (bad < 5) std::printf("MISMATCH n=%d kind=%d\n", n, kind); }
  }
  // paths of lopsided partitions: descending keys in runs of ties, beyond 4096 keys as well (there the device serves the
  // largest segments by the whole workgroup first; the ring's bound does not depend on who partitions)
  long lopsided = 0;
  for (int n : {9716, 4096, 3000, 12000, 15800})
    for (int run : {2, 3, 5, 8})
      for (int shift : {0, 1}) {
        std::vector<float> k(n);
        for (int i = 0; i < n; ++i) k[i] = -static_cast<float>((i + shift) / run);
        ++lopsided;
        if (!check(k)) { ++bad; std::printf("MISMATCH lopsided n=%d run=%d shift=%d\n", n, run, shift); }
      }
  long file_cases = 0;
  if (argc > 2) {
    FILE* f = std::fopen(argv[2], "r");
    int n;
    while (f != nullptr && std::fscanf(f, "%d", &n) == 1) {
      std::vector<float> k(n);
      for (int i = 0; i < n; ++i) { unsigned bits; if (std::fscanf(f, "%x", &bits) != 1) return 2; std::memcpy(&k[i], &bits, 4); }
      if (n <= 4096) { ++file_cases; if (!check(k)) ++bad; }
    }
    if (f != nullptr) std::fclose(f);
  }
  std::printf("mismatches: %ld of %ld (+ %ld from the file, + %ld lopsided); levels %ld, heap sorts %ld; ring overflows %ld, most alive %ld permille of the ring, "
              "arrays that queued more segments than the ring holds: %ld\n",
              bad, cases, file_cases, lopsided, g_levels, g_heaps, g_overflows, g_most_alive_permille, g_ever_above_capacity);
  return bad == 0 && g_overflows == 0 ? 0 : 1;
}
