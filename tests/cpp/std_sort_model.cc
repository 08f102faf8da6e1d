// Model of what rotational_histogram.hip does instead of calling std::sort (there is none on the device): the order in
// which libstdc++'s std::sort leaves EQUAL keys is part of the reference's result (SortSlice, rotational_scan_matcher.cc:
// 97-121 sorts (angle, point) pairs by angle only, and two returns of one slice share an angle in three scans out of
// four), so the device reproduces the algorithm, not just a sorted order:
//   introsort_loop  = rounds of median-of-three + unguarded partition on every segment above 16 elements, all segments
//                     of a round at once, each partition written as data-parallel steps (flags, two prefix counts, the
//                     k-th stop of the left pointer swaps with the k-th stop of the right pointer while they have not
//                     crossed);
//   final insertion = a STABLE sort of what the rounds left (insertion sort never moves an element past an equal one).
// This file is that formulation in plain C++, checked against the real std::sort of this machine's libstdc++ on arrays
// full of ties.  Usage: std_sort_model [cases] -> "mismatches: 0 of N".
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

struct Item {
  float key;
  int id;
  bool operator<(const Item& o) const { return key < o.key; }
};

// one round over all segments; returns false when no segment was above the threshold
static bool partition_round(std::vector<Item>& a, std::vector<int>& seg_first, std::vector<int>& seg_last) {
  const int n = static_cast<int>(a.size());
  bool any = false;
  // (a) median of three to the front, one "thread" per segment head
  for (int p = 0; p < n; ++p) {
    if (seg_first[p] != p) continue;
    const int first = p, last = seg_last[p];
    if (last - first <= 16) continue;
    any = true;
    const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
    int m;
    if (a[ia] < a[ib]) {
      if (a[ib] < a[ic]) m = ib;
      else if (a[ia] < a[ic]) m = ic;
      else m = ia;
    } else if (a[ia] < a[ic]) m = ia;
    else if (a[ib] < a[ic]) m = ic;
    else m = ib;
    std::swap(a[first], a[m]);
  }
  if (!any) return false;
  // (b) flags and ranks, every element of a large segment except its head
  std::vector<int> rank_l(n, -1), rank_r(n, -1), tmp_l(n, -1), tmp_r(n, -1), cnt_l(n, 0), cnt_r(n, 0);
  for (int p = 0; p < n; ++p) {
    const int first = seg_first[p], last = seg_last[p];
    if (last - first <= 16 || p == first) continue;
    const Item& pivot = a[first];
    if (!(a[p] < pivot)) {  // the left pointer stops here
      int k = 0;
      for (int q = first + 1; q < p; ++q) k += !(a[q] < pivot);  // a prefix count on the device
      tmp_l[first + 1 + k] = p;
    }
    if (!(pivot < a[p])) {  // the right pointer stops here
      int k = 0;
      for (int q = p + 1; q < last; ++q) k += !(pivot < a[q]);
      tmp_r[first + 1 + k] = p;
    }
  }
  // (c) pairs that have not crossed swap; the cut
  std::vector<int> cut(n, -1);
  std::vector<Item> b = a;
  for (int p = 0; p < n; ++p) {
    if (seg_first[p] != p) continue;
    const int first = p, last = seg_last[p];
    if (last - first <= 16) continue;
    int K = 0;
    while (first + 1 + K < last && tmp_l[first + 1 + K] >= 0 && tmp_r[first + 1 + K] >= 0 && tmp_l[first + 1 + K] < tmp_r[first + 1 + K]) {
      std::swap(b[tmp_l[first + 1 + K]], b[tmp_r[first + 1 + K]]);
      ++K;
    }
    // where the left pointer stops next: its (K+1)-th original stop if that lies before the last swapped right position
    int i = 1 << 30;
    if (first + 1 + K < last && tmp_l[first + 1 + K] >= 0) i = tmp_l[first + 1 + K];
    if (K > 0) i = std::min(i, tmp_r[first + 1 + K - 1]);
    cut[first] = i;
  }
  a.swap(b);
  // (d) new segments
  std::vector<int> nf = seg_first, nl = seg_last;
  for (int p = 0; p < n; ++p) {
    const int first = seg_first[p], last = seg_last[p];
    if (last - first <= 16) continue;
    const int c = cut[first];
    if (p < c) { nf[p] = first; nl[p] = c; } else { nf[p] = c; nl[p] = last; }
  }
  seg_first.swap(nf);
  seg_last.swap(nl);
  return true;
}

// std::__partial_sort(first, last, last) = __make_heap + __sort_heap (bits/stl_heap.h), restated: what introsort does with
// a segment that is still above the threshold when its depth limit is used up.  Sequential, one "thread" per segment.
static void adjust_heap(Item* first, int hole, int len, Item value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (first[child] < first[child - 1]) --child;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;  // __push_heap
  while (hole > top && first[parent] < value) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}
static void heap_sort(Item* first, int len) {
  if (len >= 2)
    for (int parent = (len - 2) / 2;; --parent) {
      adjust_heap(first, parent, len, first[parent]);
      if (parent == 0) break;
    }
  for (int last = len; last > 1;) {
    --last;
    const Item value = first[last];
    first[last] = first[0];
    adjust_heap(first, 0, last, value);
  }
}

static long g_depth_limit_cases = 0;
static bool model_sort(std::vector<Item>& a) {
  const int n = static_cast<int>(a.size());
  if (n == 0) return true;
  int depth = 0;
  for (int m = n; m > 1; m >>= 1) ++depth;
  depth *= 2;
  std::vector<int> seg_first(n, 0), seg_last(n, n);
  for (;;) {
    bool large = false;
    for (int p = 0; p < n; ++p) large = large || (seg_first[p] == p && seg_last[p] - p > 16);
    if (!large) break;
    if (depth == 0) {  // every segment still above the threshold is heap-sorted where it lies
      ++g_depth_limit_cases;
      for (int p = 0; p < n; ++p)
        if (seg_first[p] == p && seg_last[p] - p > 16) heap_sort(a.data() + p, seg_last[p] - p);
      break;
    }
    --depth;
    partition_round(a, seg_first, seg_last);
  }
  std::stable_sort(a.begin(), a.end());  // the final insertion sort
  return true;
}

int main(int argc, char** argv) {
  const long cases = argc > 1 ? atol(argv[1]) : 3000;
  std::mt19937 rng(12345);
  long bad = 0;
  for (long c = 0; c < cases; ++c) {
    const int sizes[] = {0, 1, 2, 15, 16, 17, 18, 31, 33, 64, 100, 257, 700, 1000, 2048, 4096};
    int n = sizes[c % 16];
    if (c % 5 == 0) n = static_cast<int>(rng() % 4097);
    const int kinds = 1 + static_cast<int>(rng() % 6);
    std::vector<Item> a(n);
    for (int i = 0; i < n; ++i) {
      float k;
      switch (kinds) {
        case 1: k = static_cast<float>(rng() % 3); break;                       // almost everything ties
        case 2: k = static_cast<float>(rng() % 50); break;
        case 3: k = static_cast<float>(rng() % (n / 2 + 1)); break;             // pairs
        case 4: k = static_cast<float>(rng()) * 1e-6f; break;                   // few ties
        case 5: k = static_cast<float>(i / 7); break;                           // sorted runs of ties
        default: k = static_cast<float>((n - i) / 3); break;                    // reversed runs of ties
      }
      a[i] = Item{k, i};
    }
    std::vector<Item> want = a, got = a;
    std::sort(want.begin(), want.end());
    model_sort(got);
    for (int i = 0; i < n; ++i)
      if (want[i].id != got[i].id) { ++bad; break; }
  }
  // arrays from a file: [n, n keys] ... as text (the slices of real scans, where the depth limit IS reached)
  long file_cases = 0;
  if (argc > 2) {
    FILE* f = std::fopen(argv[2], "r");
    int n;
    while (f != nullptr && std::fscanf(f, "%d", &n) == 1) {
      std::vector<Item> a(n);
      for (int i = 0; i < n; ++i) {
        unsigned bits;
        if (std::fscanf(f, "%x", &bits) != 1) return 2;
        float k;
        std::memcpy(&k, &bits, 4);
        a[i] = Item{k, i};
      }
      std::vector<Item> want = a, got = a;
      std::sort(want.begin(), want.end());
      model_sort(got);
      ++file_cases;
      for (int i = 0; i < n; ++i)
        if (want[i].id != got[i].id) { ++bad; break; }
    }
    if (f != nullptr) std::fclose(f);
  }
  std::printf("mismatches: %ld of %ld (+ %ld from the file; depth limit reached in %ld)\n", bad, cases, file_cases, g_depth_limit_cases);
  return bad == 0 ? 0 : 1;
}
