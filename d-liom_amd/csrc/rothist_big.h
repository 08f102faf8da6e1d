// ComputeHistogram for height slices that do not fit the LDS path of rotational_histogram.hip (more than kMaxSlice = 4096
// returns in one 0.2 m slice -- the floor of every real scan: 15 000 returns of a 0.15 m-filtered 64 x 1024 scan, 60 000
// of a 128 x 2048 one).  Included by rotational_histogram.hip INSIDE namespace dliom::rothist, after its helpers.
// tests/cpp/hist_big_model.cc is this formulation in plain C++, pinned against a direct restatement of
// rotational_scan_matcher.cc:29-123,159-170 with the host's std::sort and atan2f.
//
// One workgroup of 1024 threads per big slice, arrays in HBM (L2 resident), thread t owning a contiguous range of
// positions in every pass:
//   big_prepare_kernel   the slice's points in input order (compaction), ComputeCentroid as an exact parallel replay of
//                        the sequential float sum (exact_sum.h), angles (the restated atan2f), items (slice << 32 |
//                        ordered angle bits, position) in input order
//   hipcub radix sort    of all big slices' items at once, 38 key bits, STABLE: by (slice, angle, input position)
//   big_slice_kernel     std::sort's order of equal angles (below), the sorted points, their centroid (exact replay
//                        again), the `last_point` chain by pointer doubling, the contributions in order
// std::sort's order of EQUAL angles (SortSlice sorts by angle only; which of two equal angles comes first decides
// `last_point`): introsort's partition rounds are replayed on the input-order array as in the LDS path, but only on
// segments that still hold two tied elements -- a segment without ties has a unique sorted order, and the final insertion
// sort is stable, so all that is needed of the arrangement is where the tied elements are: every group of equal angles
// is then ordered by arrangement position.
// `last_point` (:70-80) moves to the first live point farther than kMaxDistance from it.  On a floor slice sorted by
// angle nearly every point is such a jump (13 688 of 14 460), so walking the chain costs one step per point.  Instead:
// next(i) for EVERY i in parallel, then the nodes on the path 0 -> next(0) -> ... by pointer doubling: marks spread along
// next^(2^d) while the pointers are squared (marked nodes are path nodes at every moment, so the passes need no
// snapshot).  A point's last_point is the last marked position before it; a marked point contributes nothing.

#ifdef DLIOM_EXPERIMENTS
__device__ unsigned long long dbg_big[64 * 16];
#define DLIOM_BSTAMP(k) if (threadIdx.x == 0 && blockIdx.x < 4) dbg_big[(blockIdx.x + 4 * (kernel_id)) * 16 + (k)] = __builtin_readcyclecounter()
#else
#define DLIOM_BSTAMP(k)
#endif

constexpr size_t kBigLdsBytes = static_cast<size_t>(kMaxSlice) * 8 + 2 * static_cast<size_t>(kMaxSlice + 8) * 2 + sizeof(Queue) + kMaxSlice + 64;
constexpr int kMaxBig = 63;         // big slices per cloud (slice ordinal 63 is the sort's padding key)
constexpr int kBigKeyBits = 38;     // 32 angle bits + 6 slice bits

struct BigArrays {
  // per point of the cloud (n_padded + 64 entries each; slice b works at offset begin_b + b, so that every slice has
  // room for one sentinel behind its last entry)
  float *bx, *by;                  // the slice's points in input order
  unsigned long long *key_in, *key_out;
  unsigned *val_in, *val_out;
  unsigned long long* arr;         // introsort's array: key << 32 | position in the slice
  unsigned *seg_first, *seg_last, *g, *l, *tmp_l, *tmp_r, *cut, *tpre, *pos_of, *sorted_id, *jump_a, *jump_b;
  unsigned char *act, *fl, *tied, *dead, *mark;
  float *spx, *spy;                // the sorted points
  unsigned* valid;                 // [kMaxBig]: items of slice b handed to the sort
};

struct BigSlice {
  int ordinal;     // b
  int bin;         // key + kBinOrigin
  unsigned count;  // points
  unsigned begin;  // points of all slices in front of it (its region of the contribution arrays)
  unsigned n_big;
};

// The (blockIdx.x)-th slice above kMaxSlice, from the key counts (every workgroup redoes the two scans: 4096 counts).
__device__ __forceinline__ bool find_big_slice(const unsigned* __restrict__ bin_counts, unsigned* wave_sums, BigSlice* out,
                                               unsigned* sh4) {
  unsigned my_counts[kBins / kThreads], my_sum = 0u, my_big = 0u;
#pragma unroll
  for (int k = 0; k < kBins / kThreads; ++k) {
    my_counts[k] = bin_counts[threadIdx.x * (kBins / kThreads) + k];
    my_sum += my_counts[k];
    my_big += my_counts[k] > static_cast<unsigned>(kMaxSlice) ? 1u : 0u;
  }
  unsigned total_points, total_big;
  const unsigned points_before = block_exclusive_scan(my_sum, wave_sums, &total_points);
  const unsigned big_before = block_exclusive_scan(my_big, wave_sums, &total_big);
  __syncthreads();
  if (threadIdx.x == 0) sh4[0] = 0xFFFFFFFFu;
  __syncthreads();
  {
    unsigned pb = points_before, bb = big_before;
#pragma unroll
    for (int k = 0; k < kBins / kThreads; ++k) {
      if (my_counts[k] > static_cast<unsigned>(kMaxSlice)) {
        if (bb == blockIdx.x) {
          sh4[0] = threadIdx.x * (kBins / kThreads) + k;
          sh4[1] = my_counts[k];
          sh4[2] = pb;
        }
        ++bb;
      }
      pb += my_counts[k];
    }
  }
  __syncthreads();
  out->ordinal = static_cast<int>(blockIdx.x);
  out->bin = static_cast<int>(sh4[0]);
  out->count = sh4[1];
  out->begin = sh4[2];
  out->n_big = total_big;
  const bool found = sh4[0] != 0xFFFFFFFFu;
  __syncthreads();
  return found;
}

// thread t owns positions [lo, hi) of [0, m)
__device__ __forceinline__ void owned_range(int m, int* lo, int* hi) {
  const int per = (m + kThreads - 1) / kThreads;
  *lo = min(m, static_cast<int>(threadIdx.x) * per);
  *hi = min(m, *lo + per);
}

// inclusive running maximum over the workgroup's threads in thread order (ints), returned EXCLUSIVE (threads before this one)
__device__ __forceinline__ int block_exclusive_max(int v, int identity, int* wave_part /* 16 ints of LDS */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if (lane >= d) incl = max(incl, up);
  }
  __syncthreads();
  if (lane == 63) wave_part[wave] = incl;
  __syncthreads();
  int before = identity;
  for (int w = 0; w < wave; ++w) before = max(before, wave_part[w]);
  int excl = __shfl_up(incl, 1, 64);
  if (lane == 0) excl = identity;
  return max(before, excl);
}

__device__ __forceinline__ unsigned key_of(unsigned long long item) { return static_cast<unsigned>(item >> 32); }

// ---- std::sort's order of equal keys on a slice of any size --------------------------------------------------------
// Thread t owns the positions [lo, hi) in every pass; they are visited four at a time with the loads of all four issued
// before anything depends on them: a plain loop of load -> use -> store pays the memory latency once per position, and
// with ~15 positions per thread and a dozen passes per partition round that was 1.3 ms for a floor slice of 14 500
// returns (round 4's first version).
template <class Stage1, class Stage2, class Use>
__device__ __forceinline__ void for_owned4(int lo, int hi, Stage1 stage1, Stage2 stage2, Use use) {
  for (int p0 = lo; p0 < hi; p0 += 4) {
    const int q0 = p0, q1 = min(p0 + 1, hi - 1), q2 = min(p0 + 2, hi - 1), q3 = min(p0 + 3, hi - 1);
    const auto a0 = stage1(q0);
    const auto a1 = stage1(q1);
    const auto a2 = stage1(q2);
    const auto a3 = stage1(q3);
    const auto b0 = stage2(q0, a0);
    const auto b1 = stage2(q1, a1);
    const auto b2 = stage2(q2, a2);
    const auto b3 = stage2(q3, a3);
    use(q0, a0, b0);
    if (p0 + 1 < hi) use(q1, a1, b1);
    if (p0 + 2 < hi) use(q2, a2, b2);
    if (p0 + 3 < hi) use(q3, a3, b3);
  }
}
struct U2 {
  unsigned a, b;
};
struct U4 {
  unsigned a, b, c, d;
};

// in:  sk/sv   the m items sorted by (key, input position): key in the low 32 bits of sk, position in sv (< count)
//      ik/iv   the same items in input order
// out: sorted_id[j] = position (in the slice) of the j-th element of std::sort's result
// Scratch arrays have m + 1 entries.  The partition rounds run on arrays in HBM while more than kMaxSlice elements lie in
// segments that still hold ties, then in LDS (lds_a / lds_sc: wave_sort_arrangement, the replay of the small slices).  Returns false when
// the depth limit's heap sort would have to run on a segment too large to do by one thread in global memory (flags |= 4:
// the host entry point takes the cloud).
__device__ bool big_sort_order(const unsigned long long* __restrict__ sk, const unsigned* __restrict__ sv,
                               const unsigned long long* __restrict__ ik, const unsigned* __restrict__ iv, int m, int count,
                               const BigArrays& A, unsigned off, unsigned* wave_sums, unsigned long long* lds_a,
                               const SortScratch& lds_sc) {
  unsigned long long* __restrict__ arr = A.arr + off;
  unsigned* __restrict__ seg_first = A.seg_first + off;
  unsigned* __restrict__ seg_last = A.seg_last + off;
  unsigned* __restrict__ g = A.g + off;
  unsigned* __restrict__ l = A.l + off;
  unsigned* __restrict__ tmp_l = A.tmp_l + off;
  unsigned* __restrict__ tmp_r = A.tmp_r + off;
  unsigned* __restrict__ cut = A.cut + off;
  unsigned* __restrict__ tpre = A.tpre + off;
  unsigned* __restrict__ pos_of = A.pos_of + off;
  unsigned* __restrict__ sorted_id = A.sorted_id + off;
  unsigned char* __restrict__ act = A.act + off;
  unsigned char* __restrict__ fl = A.fl + off;
  unsigned char* __restrict__ tied = A.tied + off;
  int lo, hi;
  {  // positions in the slice run over [0, count); m <= count of them are items
    int clo, chi;
    owned_range(count, &clo, &chi);
    for (int p = clo; p < chi; ++p) tied[p] = 0;
  }
  owned_range(m, &lo, &hi);
  __syncthreads();
  int t = 0;
  for_owned4(
      lo, hi, [&](int j) { return U2{static_cast<unsigned>(sk[j]), static_cast<unsigned>(sk[min(j + 1, m - 1)])}; },
      [&](int j, U2) { return U2{sv[j], sv[min(j + 1, m - 1)]}; },
      [&](int j, U2 k, U2 v) {
        if (j + 1 < m && k.a == k.b) {
          tied[v.a] = 1;
          tied[v.b] = 1;
          t = 1;
        }
      });
  const bool any_tie = __syncthreads_or(t) != 0;
  if (!any_tie) {
    for_owned4(lo, hi, [&](int j) { return sv[j]; }, [](int, unsigned) { return 0; }, [&](int j, unsigned v, int) { sorted_id[j] = v; });
    __syncthreads();
    return true;
  }
  for_owned4(
      lo, hi, [&](int p) { return U2{static_cast<unsigned>(ik[p]), iv[p]}; }, [](int, U2) { return 0; },
      [&](int p, U2 x, int) {
        arr[p] = (static_cast<unsigned long long>(x.a) << 32) | x.b;
        seg_first[p] = 0u;
        seg_last[p] = static_cast<unsigned>(m);
      });
  int depth = 0;
  for (int v = m; v > 1; v >>= 1) ++depth;
  depth *= 2;
  bool ok = true;
  __syncthreads();
  for (;;) {
    // which segments still matter: above the threshold and holding at least two tied elements
    unsigned cnt = 0u;
    for_owned4(
        lo, hi, [&](int p) { return static_cast<unsigned>(arr[p]); }, [&](int, unsigned id) { return static_cast<unsigned>(tied[id]); },
        [&](int p, unsigned, unsigned f) {
          fl[p] = static_cast<unsigned char>(f);
          cnt += f;
        });
    unsigned total;
    unsigned run = block_exclusive_scan(cnt, wave_sums, &total);
    for_owned4(
        lo, hi, [&](int p) { return static_cast<unsigned>(fl[p]); }, [](int, unsigned) { return 0; },
        [&](int p, unsigned f, int) {
          tpre[p] = run;
          run += f;
        });
    if (threadIdx.x == 0) tpre[m] = total;
    __syncthreads();
    int any = 0, heap_too_large = 0;
    unsigned active = 0u;
    for_owned4(
        lo, hi, [&](int p) { return U2{seg_first[p], seg_last[p]}; }, [&](int, U2 s) { return U2{tpre[s.a], tpre[s.b]}; },
        [&](int p, U2 s, U2 c) {
          const int a = (s.b - s.a > 16u && c.b - c.a >= 2u) ? 1 : 0;
          act[p] = static_cast<unsigned char>(a);
          any |= a;
          active += static_cast<unsigned>(a);
          if (a && depth == 0 && s.b - s.a > 8192u) heap_too_large = 1;
        });
    if (__syncthreads_or(any) == 0) break;
    unsigned active_total;
    const unsigned compact_at = block_exclusive_scan(active, wave_sums, &active_total);
    if (active_total <= static_cast<unsigned>(kMaxSlice)) {
      // ---- the rest in LDS: the active segments, packed in order (a segment is active as a whole, so segments stay
      //      contiguous), under the replay of the small slices; identities are the packed indices at this moment
      unsigned* __restrict__ cpos = g;  // packed index -> position; g, l, tpre are free from here on
      unsigned* __restrict__ cid = l;   // packed index -> position in the slice (the item's low word)
      unsigned* __restrict__ cidx = tpre;  // position -> packed index (active positions)
      {
        unsigned c = compact_at;
        for_owned4(
            lo, hi, [&](int p) { return static_cast<unsigned>(act[p]); }, [](int, unsigned) { return 0; },
            [&](int p, unsigned a, int) {
              if (a) cidx[p] = c++;
            });
      }
      __syncthreads();
      queue_init(lds_sc.queue);
      __syncthreads();
      for_owned4(
          lo, hi, [&](int p) { return U4{static_cast<unsigned>(act[p]), seg_first[p], seg_last[p], cidx[p]}; },
          [&](int p, U4 x) { return x.a ? U2{cidx[x.b], cidx[x.c - 1u]} : U2{0u, 0u}; },
          [&](int p, U4 x, U2 sfl) {
            if (x.a) {
              const unsigned long long item = arr[p];
              const unsigned c = x.d;
              lds_a[c] = (item & 0xffffffff00000000ull) | c;
              const_cast<unsigned char*>(lds_sc.tied)[c] = tied[static_cast<unsigned>(item)];
              cpos[c] = static_cast<unsigned>(p);
              cid[c] = static_cast<unsigned>(item);
              if (x.b == static_cast<unsigned>(p)) queue_push(lds_sc.queue, static_cast<int>(sfl.a), static_cast<int>(sfl.b) + 1, depth);
            }
          });
      const int T = static_cast<int>(active_total);
      __syncthreads();
      if (!wave_sort_arrangement(lds_a, lds_sc)) ok = false;
      for (int q = threadIdx.x; q < T; q += kThreads) {
        const unsigned long long item = lds_a[q];
        arr[cpos[q]] = (item & 0xffffffff00000000ull) | cid[static_cast<unsigned>(item) & 0xffffu];
      }
      __syncthreads();
      break;
    }
    if (depth == 0) {
      // std::sort's depth limit: heap sort (restated in rotational_histogram.hip) of what is left, one thread per segment
      if (__syncthreads_or(heap_too_large) != 0) {
        ok = false;
        break;
      }
      for (int p = lo; p < hi; ++p)
        if (act[p] && seg_first[p] == static_cast<unsigned>(p)) heap_sort_keys(arr + p, static_cast<int>(seg_last[p]) - p);
      __syncthreads();
      break;
    }
    --depth;
    // (a) __move_median_to_first(first, first + 1, mid, last - 1)
    for (int p = lo; p < hi; ++p)
      if (act[p] && seg_first[p] == static_cast<unsigned>(p)) {
        const int first = p, last = static_cast<int>(seg_last[p]);
        const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
        const unsigned ka = key_of(arr[ia]), kb = key_of(arr[ib]), kc = key_of(arr[ic]);
        int md;
        if (ka < kb) {
          if (kb < kc) md = ib;
          else if (ka < kc) md = ic;
          else md = ia;
        } else if (ka < kc) md = ia;
        else if (kb < kc) md = ic;
        else md = ib;
        const unsigned long long tmp = arr[first];
        arr[first] = arr[md];
        arr[md] = tmp;
      }
    __syncthreads();
    // (b) where the two pointers of __unguarded_partition stop
    unsigned gc = 0u, lc = 0u;
    for_owned4(
        lo, hi, [&](int p) { return U4{static_cast<unsigned>(act[p]), seg_first[p], key_of(arr[p]), 0u}; },
        [&](int, U4 x) { return x.a ? key_of(arr[x.b]) : 0u; },
        [&](int p, U4 x, unsigned pivot) {
          unsigned ge = 0u, le = 0u;
          if (x.a && x.b != static_cast<unsigned>(p)) {
            ge = x.c < pivot ? 0u : 1u;
            le = pivot < x.c ? 0u : 1u;
          }
          fl[p] = static_cast<unsigned char>(ge | (le << 1));
          gc += ge;
          lc += le;
        });
    unsigned gtotal, ltotal;
    unsigned gb = block_exclusive_scan(gc, wave_sums, &gtotal);
    unsigned lb = block_exclusive_scan(lc, wave_sums, &ltotal);
    for_owned4(
        lo, hi, [&](int p) { return static_cast<unsigned>(fl[p]); }, [](int, unsigned) { return 0; },
        [&](int p, unsigned f, int) {
          g[p] = gb;
          l[p] = lb;
          gb += f & 1u;
          lb += f >> 1;
        });
    if (threadIdx.x == 0) {
      g[m] = gtotal;
      l[m] = ltotal;
    }
    __syncthreads();
    for_owned4(
        lo, hi, [&](int p) { return U4{static_cast<unsigned>(act[p]) != 0u ? static_cast<unsigned>(fl[p]) : 0u, seg_first[p], seg_last[p], g[p]}; },
        [&](int p, U4 x) { return x.a != 0u ? U4{g[x.b + 1u], l[x.c], l[p + 1], 0u} : U4{0u, 0u, 0u, 0u}; },
        [&](int p, U4 x, U4 y) {
          if (x.b == static_cast<unsigned>(p)) return;  // the pivot
          if (x.a & 1u) tmp_l[x.b + 1u + (x.d - y.a)] = static_cast<unsigned>(p);
          if (x.a & 2u) tmp_r[x.b + 1u + (y.b - y.c)] = static_cast<unsigned>(p);
        });
    __syncthreads();
    // (c) the k-th pair swaps while the pointers have not crossed; the thread at the boundary knows the cut
    for_owned4(
        lo, hi, [&](int q) { return U4{static_cast<unsigned>(act[q]), seg_first[q], seg_last[q], 0u}; },
        [&](int q, U4 x) {
          if (x.a == 0u || x.b == static_cast<unsigned>(q)) return U4{0u, 0u, 0u, 0u};
          return U4{g[x.c] - g[x.b + 1u], l[x.c] - l[x.b + 1u], tmp_l[q], tmp_r[q]};
        },
        [&](int q, U4 x, U4 y) {
          if (x.a == 0u || x.b == static_cast<unsigned>(q)) return;
          const unsigned first = x.b, kk = static_cast<unsigned>(q) - (first + 1u);
          const unsigned cnt_l = y.a, cnt_r = y.b;
          const bool v = kk < cnt_l && kk < cnt_r && y.c < y.d;
          if (v) {
            const unsigned long long xl = arr[y.c], xr = arr[y.d];
            arr[y.c] = xr;
            arr[y.d] = xl;
          }
          auto valid = [&](unsigned j) { return j < cnt_l && j < cnt_r && tmp_l[first + 1u + j] < tmp_r[first + 1u + j]; };
          int K = -1;
          if (kk == 0u && !v) K = 0;
          else if (v && !valid(kk + 1u)) K = static_cast<int>(kk) + 1;
          if (K >= 0) {
            unsigned i = 0x7fffffffu;
            if (static_cast<unsigned>(K) < cnt_l) i = tmp_l[first + 1u + static_cast<unsigned>(K)];
            if (K > 0) i = min(i, tmp_r[first + 1u + static_cast<unsigned>(K) - 1u]);
            cut[first] = i;
          }
        });
    __syncthreads();
    // (d) [first, cut) and [cut, last)
    for_owned4(
        lo, hi, [&](int p) { return U2{static_cast<unsigned>(act[p]), seg_first[p]}; }, [&](int, U2 x) { return x.a ? cut[x.b] : 0u; },
        [&](int p, U2 x, unsigned c) {
          if (x.a) {
            if (static_cast<unsigned>(p) < c) seg_last[p] = c;
            else seg_first[p] = c;
          }
        });
    __syncthreads();
  }
  if (!ok) return false;
  // where the tied elements are in the arrangement
  for_owned4(
      lo, hi, [&](int q) { return static_cast<unsigned>(arr[q]); }, [&](int, unsigned id) { return static_cast<unsigned>(tied[id]); },
      [&](int q, unsigned id, unsigned f) {
        if (f) pos_of[id] = static_cast<unsigned>(q);
      });
  __syncthreads();
  // the final insertion sort is stable: a group of equal keys ends up in arrangement order
  for_owned4(
      lo, hi, [&](int j) { return sv[j]; }, [&](int, unsigned id) { return static_cast<unsigned>(tied[id]); },
      [&](int j, unsigned id, unsigned f) {
        unsigned dst = static_cast<unsigned>(j);
        if (f) {
          const unsigned key = static_cast<unsigned>(sk[j]);
          int gs = j, ge = j + 1;
          while (gs > 0 && static_cast<unsigned>(sk[gs - 1]) == key) --gs;
          while (ge < m && static_cast<unsigned>(sk[ge]) == key) ++ge;
          const unsigned mine = pos_of[id];
          unsigned r = 0u;
          for (int u = gs; u < ge; ++u) r += pos_of[sv[u]] < mine ? 1u : 0u;
          dst = static_cast<unsigned>(gs) + r;
        }
        sorted_id[dst] = id;
      });
  __syncthreads();
  return true;
}

// ---- kernel B1: compaction, centroid, items ------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void big_prepare_kernel(const float* __restrict__ rx, const float* __restrict__ ry,
                                                               const short* __restrict__ keys, int n,
                                                               const unsigned* __restrict__ bin_counts, BigArrays A,
                                                               unsigned* __restrict__ flags) {
  __shared__ unsigned wave_sums[kThreads / 64];
  __shared__ unsigned sh4[4];
  __shared__ exact_sum::Scratch<2> es;
  BigSlice s;
#ifdef DLIOM_EXPERIMENTS
  constexpr int kernel_id = 0;
#endif
  DLIOM_BSTAMP(0);
  if (!find_big_slice(bin_counts, wave_sums, &s, sh4)) return;
  if (s.n_big > static_cast<unsigned>(kMaxBig)) {
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(flags, 16u);
    return;
  }
  DLIOM_BSTAMP(1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int key = s.bin - kBinOrigin;
  const int count = static_cast<int>(s.count);
  const unsigned off = s.begin + static_cast<unsigned>(s.ordinal);
  float* bx = A.bx + off;
  float* by = A.by + off;
  // the slice's points in input order: the compaction of slice_kernel, into HBM
  {
    const int blocks512 = (n + 511) / 512;
    const int per_wave = (blocks512 + kThreads / 64 - 1) / (kThreads / 64);
    const int blk_lo = wave * per_wave, blk_hi = min(blocks512, blk_lo + per_wave);
    unsigned mine = 0u;
    for (int blk = blk_lo; blk < blk_hi; ++blk) {
      const int i0 = blk * 512 + lane * 8;
      uint4 kk = make_uint4(0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu);
      if (i0 < n) kk = *reinterpret_cast<const uint4*>(keys + i0);
      const unsigned w4[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const short kv = static_cast<short>((w4[t >> 1] >> ((t & 1) * 16)) & 0xffffu);
        mine += (i0 + t < n && kv == key) ? 1u : 0u;
      }
    }
    unsigned wave_total = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) wave_total += __shfl_xor(wave_total, d, 64);
    __syncthreads();
    if (lane == 0) wave_sums[wave] = wave_total;
    __syncthreads();
    unsigned at = 0u;
    for (int w = 0; w < wave; ++w) at += wave_sums[w];
    for (int blk = blk_lo; blk < blk_hi; ++blk) {
      const int i0 = blk * 512 + lane * 8;
      uint4 kk = make_uint4(0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu);
      if (i0 < n) kk = *reinterpret_cast<const uint4*>(keys + i0);
      const unsigned w4[4] = {kk.x, kk.y, kk.z, kk.w};
      unsigned hits = 0u, cnt = 0u;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const short kv = static_cast<short>((w4[t >> 1] >> ((t & 1) * 16)) & 0xffffu);
        if (i0 + t < n && kv == key) {
          hits |= 1u << t;
          ++cnt;
        }
      }
      unsigned incl = cnt;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
      }
      unsigned slot = at + incl - cnt;
      while (hits != 0u) {
        const int t = __builtin_ctz(hits);
        bx[slot] = rx[i0 + t];
        by[slot] = ry[i0 + t];
        ++slot;
        hits &= hits - 1u;
      }
      at += __shfl(incl, 63, 64);
    }
    __syncthreads();
  }
  DLIOM_BSTAMP(2);
  // ComputeCentroid (:52-59), x and y (the z sum is never read)
  const float* const arrays[2] = {bx, by};
  const float zero[2] = {0.f, 0.f};
  float sums[2];
  exact_sum::block_sequential_sums<2>(arrays, count, zero, sums, es);
  const float cx = sums[0] / static_cast<float>(count), cy = sums[1] / static_cast<float>(count);
  DLIOM_BSTAMP(3);
  // SortSlice's (angle, point) pairs in input order; points closer than kMinDistance to the centroid are skipped (:111-113)
  int lo, hi;
  owned_range(count, &lo, &hi);
  unsigned valid = 0u;
  for (int i = lo; i < hi; ++i) valid += norm2(bx[i] - cx, by[i] - cy) < kMinDistance ? 0u : 1u;
  unsigned total_valid;
  unsigned at = block_exclusive_scan(valid, wave_sums, &total_valid);
  for (int i = lo; i < hi; ++i) {
    const float dx = bx[i] - cx, dy = by[i] - cy;
    if (!(norm2(dx, dy) < kMinDistance)) {
      A.key_in[off + at] = (static_cast<unsigned long long>(s.ordinal) << 32) | ordered_bits(fd_atan2f(dy, dx));
      A.val_in[off + at] = static_cast<unsigned>(i);
      ++at;
    }
  }
  if (threadIdx.x == 0) A.valid[s.ordinal] = total_valid;
  DLIOM_BSTAMP(4);
}

// ---- kernel B2: everything behind the sort -------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void big_slice_kernel(const unsigned* __restrict__ bin_counts, int histogram_size,
                                                             float squared_jump, BigArrays A, unsigned char* __restrict__ c_bucket,
                                                             float* __restrict__ c_value, unsigned* __restrict__ flags) {
  // dynamic LDS: the arrays of the small slices' replay (libstdcxx_sort_arrangement) for the rounds that fit; the exact
  // sums' scratch lies over them (never in use at the same time)
  extern __shared__ __attribute__((aligned(16))) unsigned long long big_lds[];
  unsigned long long* lds_a = big_lds;
  unsigned short* u16_base = reinterpret_cast<unsigned short*>(lds_a + kMaxSlice);
  constexpr int kU16 = kMaxSlice + 8;
  const SortScratch lds_sc{u16_base, u16_base + kU16,
                           reinterpret_cast<unsigned char*>(u16_base + 2 * kU16) + sizeof(Queue),
                           reinterpret_cast<Queue*>(u16_base + 2 * kU16)};
  static_assert(sizeof(exact_sum::Scratch<2>) <= kBigLdsBytes, "the exact sums' scratch fits the replay's arrays");
  exact_sum::Scratch<2>& es = *reinterpret_cast<exact_sum::Scratch<2>*>(big_lds);
  __shared__ unsigned wave_sums[kThreads / 64];
  __shared__ int wave_max[kThreads / 64];
  __shared__ unsigned sh4[4];
  BigSlice s;
#ifdef DLIOM_EXPERIMENTS
  constexpr int kernel_id = 1;
#endif
  DLIOM_BSTAMP(0);
  if (!find_big_slice(bin_counts, wave_sums, &s, sh4)) return;
  if (s.n_big > static_cast<unsigned>(kMaxBig)) return;  // flagged by big_prepare_kernel
  const unsigned off = s.begin + static_cast<unsigned>(s.ordinal);
  const int m = static_cast<int>(A.valid[s.ordinal]);
  if (m == 0) return;
  unsigned sorted_at = 0u;  // the sort packs the slices' items back to back in slice order
  for (int b = 0; b < s.ordinal; ++b) sorted_at += A.valid[b];
  const unsigned long long* sk = A.key_out + sorted_at;
  const unsigned* sv = A.val_out + sorted_at;
  if (!big_sort_order(sk, sv, A.key_in + off, A.val_in + off, m, static_cast<int>(s.count), A, off, wave_sums, lds_a, lds_sc)) {
    if (threadIdx.x == 0) atomicOr(flags, 4u);
    return;
  }
  DLIOM_BSTAMP(1);
  const unsigned* sorted_id = A.sorted_id + off;
  const float* bx = A.bx + off;
  const float* by = A.by + off;
  float* px = A.spx + off;
  float* py = A.spy + off;
  int lo, hi;
  owned_range(m, &lo, &hi);
  for (int j = lo; j < hi; ++j) {
    const unsigned id = sorted_id[j];
    px[j] = bx[id];
    py[j] = by[id];
  }
  __syncthreads();
  DLIOM_BSTAMP(2);
  // AddPointCloudSliceToHistogram: centroid of the SORTED points (:68)
  const float* const arrays[2] = {px, py};
  const float zero[2] = {0.f, 0.f};
  float sums[2];
  exact_sum::block_sequential_sums<2>(arrays, m, zero, sums, es);
  const float cx = sums[0] / static_cast<float>(m), cy = sums[1] / static_cast<float>(m);
  DLIOM_BSTAMP(3);
  unsigned char* dead = A.dead + off;
  unsigned char* mark = A.mark + off;
  unsigned* ja = A.jump_a + off;
  unsigned* jb = A.jump_b + off;
  for (int j = lo; j < hi; ++j) {
    dead[j] = norm2(px[j] - cx, py[j] - cy) < kMinDistance ? 1 : 0;
    mark[j] = j == 0 ? 1 : 0;
  }
  if (threadIdx.x == 0) {
    mark[m] = 0;
    ja[m] = static_cast<unsigned>(m);
    jb[m] = static_cast<unsigned>(m);
  }
  __syncthreads();
  // next(i): the first live point farther than kMaxDistance from point i (squared lengths: rotational_histogram.hip)
  for (int i = lo; i < hi; ++i) {
    const float ax = px[i], ay = py[i];
    int j = i + 1;
    for (; j < m; ++j) {
      if (dead[j]) continue;
      const float dx = px[j] - ax, dy = py[j] - ay;
      if (dx * dx + dy * dy >= squared_jump) break;
    }
    ja[i] = static_cast<unsigned>(j);
  }
  __syncthreads();
  DLIOM_BSTAMP(4);
  // (positions i = tid, tid + 1024, ... here: the levels need no prefix over positions, and consecutive lanes on
  // consecutive entries with eight independent loads in flight hide the latency of the dependent gathers)
  for (int d = 0; (1 << d) < 2 * m; ++d) {
    for (int i0 = static_cast<int>(threadIdx.x); i0 < m; i0 += 8 * kThreads) {
      unsigned t[8], t2[8];
      unsigned char mk[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * kThreads;
        t[u] = i < m ? ja[i] : static_cast<unsigned>(m);
        mk[u] = i < m ? mark[i] : 0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) t2[u] = ja[t[u]];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * kThreads;
        if (i < m) {
          if (mk[u]) mark[t[u]] = 1;
          jb[i] = t2[u];
        }
      }
    }
    __syncthreads();
    unsigned* t = ja;
    ja = jb;
    jb = t;
  }
  DLIOM_BSTAMP(5);
  // last_point of point j = the last marked position before it (position 0 to begin with)
  int last_marked = -1;
  for (int j = lo; j < hi; ++j)
    if (mark[j]) last_marked = j;
  int anchor = max(0, block_exclusive_max(last_marked, -1, wave_max));
  unsigned emitted = 0u;
  // two passes over the owned points: count, then write in order
  for (int pass = 0; pass < 2; ++pass) {
    int a = anchor;
    unsigned at = 0u;
    if (pass == 1) {
      unsigned total;
      at = s.begin + block_exclusive_scan(emitted, wave_sums, &total);
    }
    for (int j = lo; j < hi; ++j) {
      const int last_point = a;
      const bool jump = mark[j] != 0 && j != 0;
      if (mark[j]) a = j;
      if (dead[j] || jump) continue;
      const float pxj = px[j], pyj = py[j];
      const float dx = pxj - px[last_point], dy = pyj - py[last_point];
      const float distance = norm2(dx, dy);
      if (distance < kMinDistance) continue;
      if (pass == 0) {
        ++emitted;
        continue;
      }
      const float ex = pxj - cx, ey = pyj - cy;
      const float direction_norm = norm2(ex, ey);
      const float dot = (dx / distance) * (ex / direction_norm) + (dy / distance) * (ey / direction_norm);
      c_bucket[at] = static_cast<unsigned char>(bucket_of(fd_atan2f(dy, dx), histogram_size));
      c_value[at] = fmaxf(0.f, 1.f - fabsf(dot));
      ++at;
    }
  }
  DLIOM_BSTAMP(6);
#ifdef DLIOM_EXPERIMENTS
  if (threadIdx.x == 0 && blockIdx.x < 4) {
    dbg_big[(blockIdx.x + 4) * 16 + 10] = s.count;
    dbg_big[(blockIdx.x + 4) * 16 + 11] = static_cast<unsigned long long>(m);
  }
#endif
}

// dliom_diag_std_sort_order for more than kMaxSlice keys: the sorted (key, position) pairs come from the radix sort
__global__ __launch_bounds__(kThreads) void big_sort_order_kernel(int n, BigArrays A, int* __restrict__ order, int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long big_lds[];
  unsigned long long* lds_a = big_lds;
  unsigned short* u16_base = reinterpret_cast<unsigned short*>(lds_a + kMaxSlice);
  constexpr int kU16 = kMaxSlice + 8;
  const SortScratch lds_sc{u16_base, u16_base + kU16,
                           reinterpret_cast<unsigned char*>(u16_base + 2 * kU16) + sizeof(Queue),
                           reinterpret_cast<Queue*>(u16_base + 2 * kU16)};
  __shared__ unsigned wave_sums[kThreads / 64];
  const bool ok = big_sort_order(A.key_out, A.val_out, A.key_in, A.val_in, n, n, A, 0u, wave_sums, lds_a, lds_sc);
  int lo, hi;
  owned_range(n, &lo, &hi);
  if (ok)
    for (int j = lo; j < hi; ++j) order[j] = static_cast<int>(A.sorted_id[j]);
  if (threadIdx.x == 0) *status = ok ? 0 : 1;
}
__global__ void big_sort_items_kernel(const float* __restrict__ keys, int n, unsigned long long* __restrict__ key_in,
                                      unsigned* __restrict__ val_in) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    key_in[i] = ordered_bits(keys[i]);
    val_in[i] = static_cast<unsigned>(i);
  }
}
