// Model (plain C++) of the formulation rotational_histogram.hip uses for height slices that do not fit its LDS path
// (more than 4096 returns in one 0.2 m slice: every scan with a floor), against a direct restatement of
// RotationalScanMatcher::ComputeHistogram (rotational_scan_matcher.cc:29-123,159-170) with this machine's std::sort and
// atan2f.  What is modelled, each an order dependence of the reference kept to the bit:
//   * ComputeCentroid / histogram(bucket) += value: sequential float sums as parity functions (exact_sum_model.h);
//   * SortSlice: a STABLE sort by (angle, input position) -- a radix sort on the device -- and, where equal angles end up
//     next to each other, the order std::sort would have left them in: introsort's partition rounds replayed on the
//     input-order array (std_sort_model.cc's formulation), of whose arrangement only the positions of the tied elements
//     are needed: the final insertion sort is stable, so every tie group is ordered by arrangement position;
//   * AddPointCloudSliceToHistogram's `last_point`: next(i) = first live j > i farther than kMaxDistance from i; the
//     anchors are the nodes on the path 0 -> next(0) -> ...; found by pointer doubling (marks spread with next^(2^d))
//     instead of walking 13 000 jumps of a floor slice one after the other; a point's last_point is the last anchor
//     before it, a point that is an anchor itself contributes nothing.
// Usage: hist_big_model cloud.bin [histogram_size]   (cloud.bin: int32 n, then n x 3 float32); prints "mismatches: 0".
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "exact_sum_model.h"

namespace {
constexpr float kMinDistance = 0.2f, kMaxDistance = 0.9f, kSliceHeight = 0.2f;
struct P3 { float x, y, z; };

int round_to_int(float x) { return static_cast<int>(std::lround(x)); }
float norm2(float x, float y) { return std::sqrt(x * x + y * y); }
int bucket_of(float angle, int size) {
  const float pi = static_cast<float>(M_PI);
  while (angle > pi) angle -= pi;
  while (angle < 0.f) angle += pi;
  const float zero_to_one = angle / pi;
  return std::min(std::max(round_to_int(static_cast<float>(size) * zero_to_one - 0.5f), 0), size - 1);
}

// ---- the reference, restated directly
P3 centroid_ref(const std::vector<P3>& s) {
  volatile float x = 0.f, y = 0.f, z = 0.f;
  for (const P3& p : s) { x = x + p.x; y = y + p.y; z = z + p.z; }
  const float n = static_cast<float>(s.size());
  return {x / n, y / n, z / n};
}
void add_slice_ref(const std::vector<P3>& slice, std::vector<float>* h) {
  if (slice.empty()) return;
  const P3 c = centroid_ref(slice);
  P3 last = slice.front();
  for (const P3& p : slice) {
    const float dx = p.x - last.x, dy = p.y - last.y, ex = p.x - c.x, ey = p.y - c.y;
    const float distance = norm2(dx, dy);
    if (distance < kMinDistance || norm2(ex, ey) < kMinDistance) continue;
    if (distance > kMaxDistance) { last = p; continue; }
    const float angle = std::atan2(dy, dx);
    const float dn = norm2(ex, ey);
    const float value = std::max(0.f, 1.f - std::abs((dx / distance) * (ex / dn) + (dy / distance) * (ey / dn)));
    volatile float t = (*h)[bucket_of(angle, static_cast<int>(h->size()))] + value;
    (*h)[bucket_of(angle, static_cast<int>(h->size()))] = t;
  }
}
std::vector<P3> sort_slice_ref(const std::vector<P3>& slice) {
  struct AP { float angle; P3 p; bool operator<(const AP& o) const { return angle < o.angle; } };
  const P3 c = centroid_ref(slice);
  std::vector<AP> by;
  for (const P3& p : slice) {
    const float dx = p.x - c.x, dy = p.y - c.y;
    if (norm2(dx, dy) < kMinDistance) continue;
    by.push_back(AP{std::atan2(dy, dx), p});
  }
  std::sort(by.begin(), by.end());
  std::vector<P3> r;
  for (const AP& a : by) r.push_back(a.p);
  return r;
}
std::vector<float> histogram_ref(const std::vector<P3>& cloud, int size) {
  std::vector<float> h(size, 0.f);
  std::map<int, std::vector<P3>> slices;
  for (const P3& p : cloud) slices[round_to_int(p.z / kSliceHeight)].push_back(p);
  for (const auto& s : slices) add_slice_ref(sort_slice_ref(s.second), &h);
  return h;
}

// ---- the device's formulation
struct Item { unsigned key; int id; };
unsigned ordered_bits(float f) {
  if (f == 0.f) f = 0.f;
  const unsigned u = exact_sum_model::bits_of(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// introsort's loop as rounds over all segments (std_sort_model.cc), heap sort at the depth limit; returns the arrangement
void heap_adjust(Item* first, int hole, int len, Item value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (first[child].key < first[child - 1].key) --child;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && first[parent].key < value.key) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}
void heap_sort(Item* first, int len) {
  if (len >= 2)
    for (int parent = (len - 2) / 2;; --parent) {
      heap_adjust(first, parent, len, first[parent]);
      if (parent == 0) break;
    }
  for (int last = len; last > 1;) {
    --last;
    const Item value = first[last];
    first[last] = first[0];
    heap_adjust(first, 0, last, value);
  }
}
long g_rounds = 0, g_active_elements = 0, g_round_elements = 0;
// `tied[id]`: the element shares its key with another one.  Only segments holding at least two tied elements matter
// (equal keys can only be told apart by where the partitions put them); the others are frozen -- the device stops
// looking at them, which is what makes the replay affordable on 15 000 ... 140 000 elements.
void introsort_arrangement(std::vector<Item>& a, const std::vector<char>& tied) {
  const int n = static_cast<int>(a.size());
  int depth = 0;
  for (int m = n; m > 1; m >>= 1) ++depth;
  depth *= 2;
  std::vector<int> seg_first(n, 0), seg_last(n, n);
  std::vector<char> active(n, 1);
  for (;;) {
    // a segment is active when it is above the threshold and holds >= 2 tied elements
    bool any = false;
    {
      std::vector<int> tprefix(n + 1, 0);
      for (int p = 0; p < n; ++p) tprefix[p + 1] = tprefix[p] + (tied[a[p].id] ? 1 : 0);
      for (int p = 0; p < n; ++p) {
        const int f = seg_first[p], l = seg_last[p];
        active[p] = (l - f > 16 && tprefix[l] - tprefix[f] >= 2) ? 1 : 0;
        any = any || active[p];
        g_active_elements += active[p];
      }
      g_round_elements += n;
    }
    if (!any) break;
    ++g_rounds;
    if (depth == 0) {
      for (int p = 0; p < n; ++p)
        if (seg_first[p] == p && active[p]) heap_sort(a.data() + p, seg_last[p] - p);
      break;
    }
    --depth;
    for (int p = 0; p < n; ++p) {  // (a) median of three
      if (seg_first[p] != p || !active[p]) continue;
      const int first = p, last = seg_last[p];
      const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
      int m;
      if (a[ia].key < a[ib].key) {
        if (a[ib].key < a[ic].key) m = ib;
        else if (a[ia].key < a[ic].key) m = ic;
        else m = ia;
      } else if (a[ia].key < a[ic].key) m = ia;
      else if (a[ib].key < a[ic].key) m = ic;
      else m = ib;
      std::swap(a[first], a[m]);
    }
    // (b) stops of the two pointers: prefix counts over the whole array, ranks relative to the segment
    std::vector<int> g(n + 1, 0), l(n + 1, 0), tmp_l(n, -1), tmp_r(n, -1);
    for (int p = 0; p < n; ++p) {
      int ge = 0, le = 0;
      if (active[p] && p != seg_first[p]) {
        const unsigned pivot = a[seg_first[p]].key;
        ge = a[p].key < pivot ? 0 : 1;
        le = pivot < a[p].key ? 0 : 1;
      }
      g[p + 1] = g[p] + ge;
      l[p + 1] = l[p] + le;
    }
    for (int p = 0; p < n; ++p) {
      if (!active[p] || p == seg_first[p]) continue;
      const int first = seg_first[p], last = seg_last[p];
      if (g[p + 1] != g[p]) tmp_l[first + 1 + (g[p] - g[first + 1])] = p;
      if (l[p + 1] != l[p]) tmp_r[first + 1 + (l[last] - l[p + 1])] = p;
    }
    // (c) swaps, cut
    std::vector<Item> b = a;
    std::vector<int> cut(n, -1);
    for (int q = 0; q < n; ++q) {
      if (!active[q] || q == seg_first[q]) continue;
      const int first = seg_first[q], last = seg_last[q];
      const int kk = q - (first + 1);
      const int cnt_l = g[last] - g[first + 1], cnt_r = l[last] - l[first + 1];
      auto valid = [&](int j) { return j < cnt_l && j < cnt_r && tmp_l[first + 1 + j] < tmp_r[first + 1 + j]; };
      const bool v = valid(kk);
      if (v) {
        b[tmp_l[q]] = a[tmp_r[q]];
        b[tmp_r[q]] = a[tmp_l[q]];
      }
      int K = -1;
      if (kk == 0 && !v) K = 0;
      else if (v && !valid(kk + 1)) K = kk + 1;
      if (K >= 0) {
        int i = 1 << 30;
        if (K < cnt_l) i = tmp_l[first + 1 + K];
        if (K > 0) i = std::min(i, tmp_r[first + 1 + K - 1]);
        cut[first] = i;
      }
    }
    a.swap(b);
    std::vector<int> nf = seg_first, nl = seg_last;
    for (int p = 0; p < n; ++p) {
      if (!active[p]) continue;
      const int c = cut[seg_first[p]];
      if (p < c) nl[p] = c;
      else nf[p] = c;
    }
    seg_first.swap(nf);
    seg_last.swap(nl);
  }
}

long g_unsafe = 0, g_seq_adds = 0, g_sum_elements = 0, g_levels = 0;
float exact_sum(const std::vector<float>& v) {
  g_sum_elements += static_cast<long>(v.size());
  return exact_sum_model::exact_sequential_sum(v.data(), static_cast<int>(v.size()), 0.f, &g_unsafe, &g_seq_adds);
}

struct Contribution { int bucket; float value; };
void slice_model(const std::vector<P3>& slice, int size, std::vector<Contribution>* out) {
  const int count = static_cast<int>(slice.size());
  if (count == 0) return;
  std::vector<float> cx(count), cy(count), cz(count);
  for (int i = 0; i < count; ++i) { cx[i] = slice[i].x; cy[i] = slice[i].y; cz[i] = slice[i].z; }
  const float n = static_cast<float>(count);
  const float c0x = exact_sum(cx) / n, c0y = exact_sum(cy) / n;
  (void)cz;  // the reference sums z too; nothing reads it
  // valid items in input order, stable sort by (angle, position)
  std::vector<Item> items;
  for (int i = 0; i < count; ++i) {
    const float dx = slice[i].x - c0x, dy = slice[i].y - c0y;
    if (norm2(dx, dy) < kMinDistance) continue;
    items.push_back(Item{ordered_bits(std::atan2(dy, dx)), i});
  }
  const int m = static_cast<int>(items.size());
  if (m == 0) return;
  std::vector<Item> sorted = items;
  std::stable_sort(sorted.begin(), sorted.end(), [](const Item& a, const Item& b) { return a.key < b.key; });
  // tie groups: ordered by where introsort's partitions left their members
  std::vector<char> tied(count, 0);
  bool any_tie = false;
  for (int j = 0; j + 1 < m; ++j)
    if (sorted[j].key == sorted[j + 1].key) tied[sorted[j].id] = tied[sorted[j + 1].id] = 1, any_tie = true;
  if (any_tie) {
    std::vector<Item> arr = items;
    introsort_arrangement(arr, tied);
    std::vector<int> pos_of(count, -1);
    for (int q = 0; q < m; ++q)
      if (tied[arr[q].id]) pos_of[arr[q].id] = q;
    for (int j = 0; j < m;) {
      int e = j + 1;
      while (e < m && sorted[e].key == sorted[j].key) ++e;
      if (e - j > 1) {  // rank of a member = members with a smaller arrangement position
        std::vector<Item> grp(sorted.begin() + j, sorted.begin() + e);
        for (const Item& it : grp) {
          int r = 0;
          for (const Item& o : grp) r += pos_of[o.id] < pos_of[it.id];
          sorted[j + r] = it;
        }
      }
      j = e;
    }
  }
  std::vector<float> px(m), py(m);
  for (int j = 0; j < m; ++j) { px[j] = slice[sorted[j].id].x; py[j] = slice[sorted[j].id].y; }
  const float c1x = exact_sum(px) / static_cast<float>(m), c1y = exact_sum(py) / static_cast<float>(m);
  // squared threshold of `distance > kMaxDistance`
  float s2 = kMaxDistance * kMaxDistance;
  while (std::sqrt(s2) > kMaxDistance) s2 = std::nextafter(s2, 0.f);
  while (!(std::sqrt(s2) > kMaxDistance)) s2 = std::nextafter(s2, 2.f);
  std::vector<char> dead(m);
  for (int j = 0; j < m; ++j) dead[j] = norm2(px[j] - c1x, py[j] - c1y) < kMinDistance;
  std::vector<int> jump(m + 1, m), jump2(m + 1, m);
  for (int i = 0; i < m; ++i) {
    int j = i + 1;
    for (; j < m; ++j) {
      if (dead[j]) continue;
      const float dx = px[j] - px[i], dy = py[j] - py[i];
      if (dx * dx + dy * dy >= s2) break;
    }
    jump[i] = j;
  }
  std::vector<char> mark(m + 1, 0);
  mark[0] = 1;
  for (int d = 0; (1 << d) < 2 * m; ++d) {  // marks spread with next^(2^d), then the pointers double
    ++g_levels;
    for (int i = 0; i < m; ++i)
      if (mark[i]) mark[jump[i]] = 1;
    for (int i = 0; i <= m; ++i) jump2[i] = jump[jump[i]];
    jump.swap(jump2);
  }
  int anchor = 0;  // exclusive prefix maximum of the marked positions
  for (int j = 0; j < m; ++j) {
    const int a = anchor;
    if (mark[j]) anchor = j;
    if (dead[j] || (mark[j] && j != 0)) continue;
    const float dx = px[j] - px[a], dy = py[j] - py[a];
    const float distance = norm2(dx, dy);
    if (distance < kMinDistance) continue;
    const float ex = px[j] - c1x, ey = py[j] - c1y, dn = norm2(ex, ey);
    const float value = std::max(0.f, 1.f - std::abs((dx / distance) * (ex / dn) + (dy / distance) * (ey / dn)));
    out->push_back(Contribution{bucket_of(std::atan2(dy, dx), size), value});
  }
}
std::vector<float> histogram_model(const std::vector<P3>& cloud, int size) {
  std::map<int, std::vector<P3>> slices;
  for (const P3& p : cloud) slices[round_to_int(p.z / kSliceHeight)].push_back(p);
  std::vector<Contribution> all;
  for (const auto& s : slices) slice_model(s.second, size, &all);
  std::vector<float> h(size, 0.f);
  for (int b = 0; b < size; ++b) {
    std::vector<float> v;
    for (const Contribution& c : all)
      if (c.bucket == b) v.push_back(c.value);
    h[b] = exact_sum(v);
  }
  return h;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const int size = argc > 2 ? std::atoi(argv[2]) : 120;
  FILE* f = std::fopen(argv[1], "rb");
  if (f == nullptr) return 2;
  long bad = 0, clouds = 0;
  int n;
  while (std::fread(&n, 4, 1, f) == 1) {
    std::vector<P3> cloud(n);
    if (n > 0 && std::fread(cloud.data(), 12, n, f) != static_cast<size_t>(n)) return 2;
    const std::vector<float> want = histogram_ref(cloud, size), got = histogram_model(cloud, size);
    ++clouds;
    for (int b = 0; b < size; ++b)
      if (exact_sum_model::bits_of(want[b]) != exact_sum_model::bits_of(got[b])) {
        ++bad;
        std::printf("cloud %ld bucket %d: want %a got %a\n", clouds, b, want[b], got[b]);
        break;
      }
  }
  std::printf("mismatches: %ld of %ld clouds; sums: %ld sequential adds of %ld (%ld unsafe chunks); introsort rounds %ld, "
              "active elements %ld of %ld; doubling levels %ld\n",
              bad, clouds, g_seq_adds, g_sum_elements, g_unsafe, g_rounds, g_active_elements, g_round_elements, g_levels);
  return bad == 0 ? 0 : 1;
}
