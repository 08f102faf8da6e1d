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

// dynamic LDS of big_slice_kernel: the replay's arrays (61 KB) or the exact sums' scratch (50 KB) or the chain's arrays
// (10 bytes per point) -- one after the other; one workgroup of 1024 threads per CU anyway
constexpr size_t kBigReplayBytes = static_cast<size_t>(kMaxSlice) * 8 + 2 * static_cast<size_t>(kMaxSlice + 8) * 2 + sizeof(Queue) + kMaxSlice + 64;
constexpr size_t kBigLdsBytes = 150 * 1024;
static_assert(kBigReplayBytes <= kBigLdsBytes, "the replay's arrays fit");
constexpr int kWorkListCap = 256;    // segments of a big slice's replay that still hold ties, before they fit LDS together
constexpr int kMaxBig = 63;         // big slices per cloud (slice ordinal 63 is the sort's padding key)
constexpr int kBigKeyBits = 38;     // 32 angle bits + 6 slice bits

#ifdef DLIOM_EXPERIMENTS
__device__ int dbg_coop_min = 4096;
#endif
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
#ifdef DLIOM_EXPERIMENTS
#define DLIOM_SSTAMP(k) if (threadIdx.x == 0 && blockIdx.x < 4) dbg_big[(blockIdx.x + 8) * 16 + (k)] = __builtin_readcyclecounter()
#else
#define DLIOM_SSTAMP(k)
#endif
  __shared__ int wl_first[kWorkListCap], wl_last[kWorkListCap], wl_depth[kWorkListCap];
  __shared__ int wl_n, wl_pick, wl_total, wl_overflow, wl_mode;
  __shared__ int wl_base[kWorkListCap + 1];
  __shared__ unsigned co_cnt[2][kThreads / 64];
  __shared__ unsigned co_k, co_tied[2];
  __shared__ int co_cut;
  // ---- slices up to ~16 000 points (the floor of a filtered 64- or 128-beam scan): the whole replay in LDS.  An item is
  //      32 bits there -- the angle's dense RANK among the slice's angles (the radix sort's order gives it; equal angles
  //      share it, so comparisons come out as on the angles) and the position in the slice, 16 bits each -- and the two
  //      pointers' stops are 16-bit positions: 9 bytes per point with the tie flags.  (Until this, seven workgroup-wide
  //      partitions on arrays in HBM at ~9.5 us each -- nine dependent L2 round trips a round -- then a copy into LDS
  //      for the rest: 150 us of a 10 000-point floor slice's 300.)
  {
    const size_t arr_bytes = (static_cast<size_t>(m) + 2) / 2 * 8;               // u32 [m + 1], 8-byte aligned end
    const size_t stop_bytes = (static_cast<size_t>(m) + 8 + 3) / 4 * 8;          // u16 [m + 8]
    const size_t tied_bytes = (static_cast<size_t>(count) + 8 + 15) / 16 * 16;   // u8 [count + 8]
    // the ring of the wave-per-segment stage: the segments of a level are disjoint and have more than 16 elements each
    const size_t queue_entries = 2 * static_cast<size_t>(m) / 17 + 64;
    const size_t need = arr_bytes + 2 * stop_bytes + tied_bytes + 16 + queue_entries * sizeof(uint2) + 16;
    if (need <= kBigLdsBytes && count < 65536 && 2 * stop_bytes >= 2 * static_cast<size_t>(count)) {
      char* base = reinterpret_cast<char*>(lds_a);
      unsigned* arr = reinterpret_cast<unsigned*>(base);
      unsigned short* tl = reinterpret_cast<unsigned short*>(base + arr_bytes);
      unsigned short* tr = reinterpret_cast<unsigned short*>(base + arr_bytes + stop_bytes);
      unsigned char* tied = reinterpret_cast<unsigned char*>(base + arr_bytes + 2 * stop_bytes);
      Queue* queue = reinterpret_cast<Queue*>(base + arr_bytes + 2 * stop_bytes + tied_bytes);
      unsigned short* rank_by_pos = tl;  // [count], before the partitions need the stops' arrays
      unsigned* __restrict__ sorted_id = A.sorted_id + off;
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      DLIOM_SSTAMP(0);
      for (int p = threadIdx.x; p < count + 8; p += kThreads) tied[p] = 0;
      __syncthreads();
      int t = 0;
      for (int j0 = static_cast<int>(threadIdx.x); j0 < m; j0 += 8 * kThreads) {  // (coalesced, eight loads in flight)
        unsigned ka[8], kb[8], ia[8], ib[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = min(j0 + u * kThreads, m - 1), j1 = min(j + 1, m - 1);
          ka[u] = static_cast<unsigned>(sk[j]);
          kb[u] = static_cast<unsigned>(sk[j1]);
          ia[u] = sv[j];
          ib[u] = sv[j1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + u * kThreads;
          if (j + 1 < m && ka[u] == kb[u]) {
            tied[ia[u]] = 1;
            tied[ib[u]] = 1;
            t = 1;
          }
        }
      }
      const bool any_tie = __syncthreads_or(t) != 0;
      if (!any_tie) {
        for (int j = threadIdx.x; j < m; j += kThreads) sorted_id[j] = sv[j];
        __syncthreads();
        return true;
      }
      DLIOM_SSTAMP(1);
      // the angles' dense ranks: thread t owns the sorted positions [lo, hi)
      int lo, hi;
      owned_range(m, &lo, &hi);
      {
        constexpr int kOwnMax = 16;  // (m <= 16 384 here)
        unsigned kk[kOwnMax + 1], vv[kOwnMax];
#pragma unroll
        for (int u = 0; u <= kOwnMax; ++u) kk[u] = static_cast<unsigned>(sk[min(max(lo - 1 + u, 0), m - 1)]);
#pragma unroll
        for (int u = 0; u < kOwnMax; ++u) vv[u] = sv[min(lo + u, m - 1)];
        unsigned mine = 0u;
#pragma unroll
        for (int u = 0; u < kOwnMax; ++u)
          if (lo + u < hi && lo + u > 0 && kk[u + 1] != kk[u]) ++mine;
        unsigned total;
        unsigned rank = block_exclusive_scan(mine, wave_sums, &total);
#pragma unroll
        for (int u = 0; u < kOwnMax; ++u)
          if (lo + u < hi) {
            if (lo + u > 0 && kk[u + 1] != kk[u]) ++rank;
            rank_by_pos[vv[u]] = static_cast<unsigned short>(rank);
          }
      }
      __syncthreads();
      for (int q0 = static_cast<int>(threadIdx.x); q0 < m; q0 += 8 * kThreads) {  // std::sort's input: the items in input order
        unsigned v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = iv[min(q0 + u * kThreads, m - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (q0 + u * kThreads < m) arr[q0 + u * kThreads] = (static_cast<unsigned>(rank_by_pos[v8[u]]) << 16) | v8[u];
      }
      int depth0 = 0;
      for (int v = m; v > 1; v >>= 1) ++depth0;
      depth0 *= 2;
      if (threadIdx.x == 0) {
        wl_first[0] = 0;
        wl_last[0] = m;
        wl_depth[0] = depth0;
        wl_n = m > 16 ? 1 : 0;
        wl_overflow = 0;
      }
      queue_init(queue, static_cast<unsigned>(queue_entries));
      __syncthreads();  // (rank_by_pos is dead: the stops' arrays are free)
      DLIOM_SSTAMP(2);
#ifdef DLIOM_EXPERIMENTS
      int dbg_rounds_lds = 0;
#endif
      bool ok = true;
#ifdef DLIOM_EXPERIMENTS
      const int kCoopMin = dbg_coop_min;
#else
      // larger segments are partitioned by the whole workgroup, smaller ones by a wave each (a workgroup-wide partition
      // costs ~18 000 cycles whatever the size -- nine barriers --, a wave takes ~30 cycles per element but sixteen of
      // them work side by side; measured on a 10 462-point floor slice: 239 000 cycles with 1024, 206 000 with 2048,
      // 175 000 with 4096, 230 000 with 8192)
      constexpr int kCoopMin = 4096;
#endif
      for (int guard = 0; guard < (1 << 16); ++guard) {
        if (threadIdx.x == 0) {
          int pick = -1, best = kCoopMin;
          for (int e = 0; e < wl_n; ++e) {
            const int len = wl_last[e] - wl_first[e];
            if (len > best) {
              best = len;
              pick = e;
            }
          }
          wl_pick = pick;
        }
        __syncthreads();
        if (wl_pick < 0 || wl_overflow != 0) break;
#ifdef DLIOM_EXPERIMENTS
        ++dbg_rounds_lds;
#endif
        const int first = wl_first[wl_pick], last = wl_last[wl_pick], depth = wl_depth[wl_pick];
        if (depth == 0) {  // std::sort's depth limit on a segment that large: its heap sort, sequential -- refused
          ok = false;
          break;
        }
        // (a) __move_median_to_first(first, first + 1, mid, last - 1)
        if (threadIdx.x == 0) {
          const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
          const unsigned ka = item_key(arr[ia]), kb = item_key(arr[ib]), kc = item_key(arr[ic]);
          int md;
          if (ka < kb) {
            if (kb < kc) md = ib;
            else if (ka < kc) md = ic;
            else md = ia;
          } else if (ka < kc) md = ia;
          else if (kb < kc) md = ic;
          else md = ib;
          const unsigned tmp = arr[first];
          arr[first] = arr[md];
          arr[md] = tmp;
        }
        __syncthreads();
        const unsigned pivot = item_key(arr[first]);
        // (b) the stops of the two pointers, both lists in ascending order of position: wave w takes a contiguous share
        const int n_in = last - (first + 1);
        const int per_wave = ((n_in + (kThreads / 64) * 64 - 1) / ((kThreads / 64) * 64)) * 64;
        const int w_lo = first + 1 + wave * per_wave, w_hi = min(last, w_lo + per_wave);
        unsigned cl = 0u, cr = 0u;
        for (int base2 = w_lo; base2 < w_hi; base2 += 64) {
          const int p = base2 + lane;
          const bool in = p < w_hi;
          const unsigned x = in ? item_key(arr[p]) : 0u;
          cl += __builtin_popcountll(__builtin_amdgcn_ballot_w64(in && !(x < pivot)));
          cr += __builtin_popcountll(__builtin_amdgcn_ballot_w64(in && !(pivot < x)));
        }
        if (lane == 0) {
          co_cnt[0][wave] = cl;
          co_cnt[1][wave] = cr;
        }
        __syncthreads();
        unsigned at_l = 0u, at_r = 0u, cnt_l = 0u, cnt_r = 0u;
        for (int w = 0; w < kThreads / 64; ++w) {
          if (w < wave) {
            at_l += co_cnt[0][w];
            at_r += co_cnt[1][w];
          }
          cnt_l += co_cnt[0][w];
          cnt_r += co_cnt[1][w];
        }
        unsigned short* stops_l = tl + first + 1;
        unsigned short* stops_r = tr + first + 1;
        for (int base2 = w_lo; base2 < w_hi; base2 += 64) {
          const int p = base2 + lane;
          const bool in = p < w_hi;
          const unsigned x = in ? item_key(arr[p]) : 0u;
          const bool ge = in && !(x < pivot), le = in && !(pivot < x);
          const unsigned long long ml = __builtin_amdgcn_ballot_w64(ge), mr = __builtin_amdgcn_ballot_w64(le);
          if (ge) stops_l[at_l + __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(ml >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(ml), 0u))] = static_cast<unsigned short>(p);
          if (le) stops_r[at_r + __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mr >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mr), 0u))] = static_cast<unsigned short>(p);
          at_l += __builtin_popcountll(ml);
          at_r += __builtin_popcountll(mr);
        }
        if (threadIdx.x == 0) {
          co_k = min(cnt_l, cnt_r);
          co_tied[0] = co_tied[1] = 0u;
        }
        __syncthreads();
        // (c) the k-th stop from the left swaps with the k-th from the right while they have not crossed
        const unsigned lim = min(cnt_l, cnt_r);
        {
          unsigned first_invalid = lim;
          for (unsigned k = threadIdx.x; k < lim; k += kThreads)
            if (!(stops_l[k] < stops_r[cnt_r - 1u - k])) {
              first_invalid = k;
              break;  // (the valid k are a prefix: this thread's later ones are invalid as well)
            }
          if (first_invalid < lim) atomicMin(&co_k, first_invalid);
        }
        __syncthreads();
        const unsigned K = co_k;
        for (unsigned k = threadIdx.x; k < K; k += kThreads) {
          const unsigned il = stops_l[k], ir = stops_r[cnt_r - 1u - k];
          const unsigned xl = arr[il], xr = arr[ir];
          arr[il] = xr;
          arr[ir] = xl;
        }
        if (threadIdx.x == 0) {
          unsigned c = 0x7fffffffu;  // where the left pointer stops next
          if (K < cnt_l) c = stops_l[K];
          if (K > 0u) c = min(c, static_cast<unsigned>(stops_r[cnt_r - K]));
          co_cut = static_cast<int>(c);
        }
        __syncthreads();
        const int cut = co_cut;
        // (d) [first, cut) and [cut, last): on the list if they are above the threshold and hold two tied elements
        {
          unsigned tl2 = 0u, tr2 = 0u;
          for (int p = first + static_cast<int>(threadIdx.x); p < last; p += kThreads) {
            const bool td = tied[item_id(arr[p])] != 0;
            tl2 += (td && p < cut) ? 1u : 0u;
            tr2 += (td && p >= cut) ? 1u : 0u;
          }
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) {
            tl2 += __shfl_xor(tl2, d, 64);
            tr2 += __shfl_xor(tr2, d, 64);
          }
          if (lane == 0) {
            if (tl2 != 0u) atomicAdd(&co_tied[0], tl2);
            if (tr2 != 0u) atomicAdd(&co_tied[1], tr2);
          }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
          int n = wl_n;
          wl_first[wl_pick] = wl_first[n - 1];
          wl_last[wl_pick] = wl_last[n - 1];
          wl_depth[wl_pick] = wl_depth[n - 1];
          --n;
          const int cf[2] = {first, cut}, cl2[2] = {cut, last};
          for (int c = 0; c < 2; ++c)
            if (cl2[c] - cf[c] > 16 && co_tied[c] >= 2u) {
              if (n < kWorkListCap) {
                wl_first[n] = cf[c];
                wl_last[n] = cl2[c];
                wl_depth[n] = depth - 1;
                ++n;
              } else {
                wl_overflow = 1;
              }
            }
          wl_n = n;
        }
        __syncthreads();
      }
      if (!ok || wl_overflow != 0) return false;
      DLIOM_SSTAMP(3);
#ifdef DLIOM_EXPERIMENTS
      if (threadIdx.x == 0 && blockIdx.x < 4) dbg_big[(blockIdx.x + 8) * 16 + 15] = static_cast<unsigned long long>(dbg_rounds_lds) | (static_cast<unsigned long long>(wl_n) << 32);
#endif
      // what is left: one wave per segment, level by level (wave_sort_arrangement), in place
      if (static_cast<int>(threadIdx.x) < wl_n) queue_push(queue, wl_first[threadIdx.x], wl_last[threadIdx.x], wl_depth[threadIdx.x]);
      __syncthreads();
      {
        const SortScratch sc{tl, tr, tied, queue};
        if (!wave_sort_arrangement(arr, sc)) return false;
      }
      DLIOM_SSTAMP(4);
      // where the tied elements are in the arrangement; a group of equal keys ends up in arrangement order (the final
      // insertion sort is stable)
      unsigned short* pos_of = tl;  // by position in the slice (the stops' arrays are free again)
      for (int q = threadIdx.x; q < m; q += kThreads) {
        const unsigned id = item_id(arr[q]);
        if (tied[id]) pos_of[id] = static_cast<unsigned short>(q);
      }
      __syncthreads();
      for (int j0 = static_cast<int>(threadIdx.x); j0 < m; j0 += 8 * kThreads) {
        unsigned id8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) id8[u] = sv[min(j0 + u * kThreads, m - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + u * kThreads;
          if (j >= m) continue;
          const unsigned id = id8[u];
          unsigned dst = static_cast<unsigned>(j);
          if (tied[id]) {
            const unsigned key = static_cast<unsigned>(sk[j]);
            int gs = j, ge = j + 1;
            while (gs > 0 && static_cast<unsigned>(sk[gs - 1]) == key) --gs;
            while (ge < m && static_cast<unsigned>(sk[ge]) == key) ++ge;
            const unsigned mine = pos_of[id];
            unsigned r = 0u;
            for (int w2 = gs; w2 < ge; ++w2) r += pos_of[sv[w2]] < mine ? 1u : 0u;
            dst = static_cast<unsigned>(gs) + r;
          }
          sorted_id[dst] = id;
        }
      }
      __syncthreads();
      DLIOM_SSTAMP(5);
      return true;
    }
  }
  unsigned long long* __restrict__ arr = A.arr + off;
  unsigned* __restrict__ l = A.l + off;
  unsigned* __restrict__ tmp_l = A.tmp_l + off;
  unsigned* __restrict__ tmp_r = A.tmp_r + off;
  unsigned* __restrict__ pos_of = A.pos_of + off;
  unsigned* __restrict__ sorted_id = A.sorted_id + off;
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
  for (int j0 = static_cast<int>(threadIdx.x); j0 < m; j0 += 8 * kThreads) {  // (coalesced, eight loads in flight)
    unsigned ka[8], kb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = min(j0 + u * kThreads, m - 1);
      ka[u] = static_cast<unsigned>(sk[j]);
      kb[u] = static_cast<unsigned>(sk[min(j + 1, m - 1)]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + u * kThreads;
      if (j + 1 < m && ka[u] == kb[u]) {
        tied[sv[j]] = 1;
        tied[sv[j + 1]] = 1;
        t = 1;
      }
    }
  }
  const bool any_tie = __syncthreads_or(t) != 0;
#ifdef DLIOM_EXPERIMENTS
  int dbg_round = 0;
#endif
  DLIOM_SSTAMP(0);
  if (!any_tie) {
    for (int j = threadIdx.x; j < m; j += kThreads) sorted_id[j] = sv[j];
    __syncthreads();
    return true;
  }
  for (int p0 = static_cast<int>(threadIdx.x); p0 < m; p0 += 8 * kThreads) {
    unsigned k8[8], v8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int p = min(p0 + u * kThreads, m - 1);
      k8[u] = static_cast<unsigned>(ik[p]);
      v8[u] = iv[p];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (p0 + u * kThreads < m) arr[p0 + u * kThreads] = (static_cast<unsigned long long>(k8[u]) << 32) | v8[u];
  }
  int depth0 = 0;
  for (int v = m; v > 1; v >>= 1) ++depth0;
  depth0 *= 2;
  bool ok = true;
  // ---- introsort's partitions, only where two tied elements still share a segment.  A work list of such segments
  //      (disjoint, in HBM coordinates); while they hold more than kMaxSlice elements together the LARGEST one is
  //      partitioned by the whole workgroup (coop_partition: four coalesced passes), then everything left moves to LDS
  //      and wave_sort_arrangement finishes it.  (Round 4's first version ran all segments through block-wide rounds over
  //      thread-owned positions: ~90 us per round for a floor slice of 10 000 returns, four or five rounds.)
  if (threadIdx.x == 0) {
    wl_first[0] = 0;
    wl_last[0] = m;
    wl_depth[0] = depth0;
    wl_n = m > 16 ? 1 : 0;
    wl_overflow = 0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int guard = 0; guard < (1 << 20); ++guard) {
    // What next (thread 0 decides, everybody follows): the largest segment is partitioned by the whole workgroup in HBM,
    // or as many segments as fit LDS together are finished there (wave_sort_arrangement) and leave the list.
    if (threadIdx.x == 0) {
      int pick = -1, best = 0, total = 0;
      for (int e = 0; e < wl_n; ++e) {
        const int len = wl_last[e] - wl_first[e];
        total += len;
        if (len > best) {
          best = len;
          pick = e;
        }
      }
      wl_pick = pick;
      // With a handful of tied pairs the segments that matter halve with every partition: a few workgroup-wide
      // partitions (~9 us each) until everything left fits LDS at once beat several LDS batches (~40 us each).  With ties
      // everywhere nothing shrinks: then the segments go through LDS in batches as soon as they are small enough.
      // A segment at std::sort's depth limit is heap-sorted, which wave_sort_arrangement does (one lane) for what fits LDS:
      // such a segment goes with a batch however much else is waiting.  (Round 6's soak, 18 865 keys with a fifth of
      // them tied: a path of lopsided partitions reached the limit on 1 203 elements while 5 000 others waited, the
      // workgroup-wide branch was chosen for it and refused.)
      int mode = 0;
      if (wl_n > 0) {
        const bool by_workgroup = best > kMaxSlice || (total > kMaxSlice && best > kMaxSlice / 4 && wl_depth[pick] > 0);
        mode = (by_workgroup && wl_n < kWorkListCap - 2) ? 1 : 2;
      }
      if (mode == 2) {  // the batch: entries in list order while they fit; the chosen ones move to the END of the list
        int at = 0, taken = 0, n = wl_n;
        for (int e = 0; e < n - taken;) {
          const int len = wl_last[e] - wl_first[e];
          if (len <= kMaxSlice && at + len <= kMaxSlice && taken < kWorkListCap) {
            const int last_free = n - taken - 1;
            const int f = wl_first[e], l2 = wl_last[e], d = wl_depth[e];
            wl_first[e] = wl_first[last_free];
            wl_last[e] = wl_last[last_free];
            wl_depth[e] = wl_depth[last_free];
            wl_first[last_free] = f;
            wl_last[last_free] = l2;
            wl_depth[last_free] = d;
            at += len;
            ++taken;
          } else {
            ++e;
          }
        }
        wl_total = taken;  // the batch is the last `taken` entries of the list
        if (taken == 0) wl_overflow = 1;  // a list full of segments above kMaxSlice: cannot happen below 2^18 elements
      }
      wl_mode = mode;
    }
    __syncthreads();
    if (wl_mode == 0 || wl_overflow != 0) break;
    if (wl_mode == 2) {
      // ---- a batch in LDS: the chosen segments packed one behind the other; an item's identity there is its packed index
      const int n_batch = wl_total, e0 = wl_n - n_batch;
      if (threadIdx.x == 0) {
        int at = 0;
        for (int e = 0; e < n_batch; ++e) {
          wl_base[e] = at;
          at += wl_last[e0 + e] - wl_first[e0 + e];
        }
        wl_base[n_batch] = at;
      }
      queue_init(lds_sc.queue);
      __syncthreads();
      const int T = wl_base[n_batch];
      unsigned* __restrict__ cid = l;  // packed index -> position in the slice (the item's low word)
      for (int c = threadIdx.x; c < T; c += kThreads) {
        int e = 0;
        while (c >= wl_base[e + 1]) ++e;
        const int p = wl_first[e0 + e] + (c - wl_base[e]);
        const unsigned long long item = arr[p];
        lds_a[c] = (item & 0xffffffff00000000ull) | static_cast<unsigned>(c);
        const_cast<unsigned char*>(lds_sc.tied)[c] = tied[static_cast<unsigned>(item)];
        cid[c] = static_cast<unsigned>(item);
      }
      if (static_cast<int>(threadIdx.x) < n_batch)
        queue_push(lds_sc.queue, wl_base[threadIdx.x], wl_base[threadIdx.x + 1], wl_depth[e0 + threadIdx.x]);
      __syncthreads();
      DLIOM_SSTAMP(11);
      if (!wave_sort_arrangement(lds_a, lds_sc)) ok = false;
      DLIOM_SSTAMP(12);
#ifdef DLIOM_EXPERIMENTS
      if (threadIdx.x == 0 && blockIdx.x < 4) dbg_big[(blockIdx.x + 8) * 16 + 15] = static_cast<unsigned long long>(T) | (static_cast<unsigned long long>(dbg_round) << 32);
#endif
      for (int q = threadIdx.x; q < T; q += kThreads) {
        int e = 0;
        while (q >= wl_base[e + 1]) ++e;
        const unsigned long long item = lds_a[q];
        arr[wl_first[e0 + e] + (q - wl_base[e])] = (item & 0xffffffff00000000ull) | cid[static_cast<unsigned>(item) & 0xffffu];
      }
      __syncthreads();
      if (threadIdx.x == 0) wl_n = e0;
      __syncthreads();
      if (!ok) break;
      continue;
    }
#ifdef DLIOM_EXPERIMENTS
    if (dbg_round < 9) DLIOM_SSTAMP(1 + dbg_round);
    ++dbg_round;
#endif
    const int first = wl_first[wl_pick], last = wl_last[wl_pick], depth = wl_depth[wl_pick];
    if (depth == 0) {
      // std::sort's depth limit on a segment that large: its heap sort, sequential in HBM -- refused (the host takes the cloud)
      ok = false;
      break;
    }
    // (a) __move_median_to_first(first, first + 1, mid, last - 1)
    if (threadIdx.x == 0) {
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
    const unsigned pivot = key_of(arr[first]);
    // (b) the stops of the two pointers, both lists in ascending order of position: wave w takes a contiguous share of
    //     (first, last) in steps of 64 positions, four steps' loads in flight
    const int n_in = last - (first + 1);
    const int per_wave = ((n_in + (kThreads / 64) * 64 - 1) / ((kThreads / 64) * 64)) * 64;
    const int w_lo = first + 1 + wave * per_wave, w_hi = min(last, w_lo + per_wave);
    unsigned cl = 0u, cr = 0u;
    for (int base = w_lo; base < w_hi; base += 256) {
      unsigned x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = base + 64 * u + lane;
        x[u] = p < w_hi ? key_of(arr[p]) : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool in = base + 64 * u + lane < w_hi;
        cl += __builtin_popcountll(__builtin_amdgcn_ballot_w64(in && !(x[u] < pivot)));
        cr += __builtin_popcountll(__builtin_amdgcn_ballot_w64(in && !(pivot < x[u])));
      }
    }
    if (lane == 0) {
      co_cnt[0][wave] = cl;
      co_cnt[1][wave] = cr;
    }
    __syncthreads();
    unsigned at_l = 0u, at_r = 0u, cnt_l = 0u, cnt_r = 0u;
    for (int w = 0; w < kThreads / 64; ++w) {
      if (w < wave) {
        at_l += co_cnt[0][w];
        at_r += co_cnt[1][w];
      }
      cnt_l += co_cnt[0][w];
      cnt_r += co_cnt[1][w];
    }
    unsigned* stops_l = tmp_l + first + 1;
    unsigned* stops_r = tmp_r + first + 1;
    for (int base = w_lo; base < w_hi; base += 256) {
      unsigned x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = base + 64 * u + lane;
        x[u] = p < w_hi ? key_of(arr[p]) : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = base + 64 * u + lane;
        const bool in = p < w_hi;
        const bool ge = in && !(x[u] < pivot), le = in && !(pivot < x[u]);
        const unsigned long long ml = __builtin_amdgcn_ballot_w64(ge), mr = __builtin_amdgcn_ballot_w64(le);
        if (ge) stops_l[at_l + __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(ml >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(ml), 0u))] = static_cast<unsigned>(p);
        if (le) stops_r[at_r + __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mr >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mr), 0u))] = static_cast<unsigned>(p);
        at_l += __builtin_popcountll(ml);
        at_r += __builtin_popcountll(mr);
      }
    }
    if (threadIdx.x == 0) {
      co_k = min(cnt_l, cnt_r);
      co_tied[0] = co_tied[1] = 0u;
    }
    __syncthreads();
    // (c) the k-th stop from the left swaps with the k-th from the right while they have not crossed: K = the first k
    //     that does not (the valid k are 0 .. K - 1: positions from the left grow with k, from the right they fall)
    const unsigned lim = min(cnt_l, cnt_r);
    {
      unsigned first_invalid = lim;
      for (unsigned k0 = threadIdx.x; k0 < lim; k0 += 4 * kThreads) {
        unsigned a4[4], b4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned k = k0 + u * kThreads;
          a4[u] = k < lim ? stops_l[k] : 0u;
          b4[u] = k < lim ? stops_r[cnt_r - 1u - k] : 1u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned k = k0 + u * kThreads;
          if (k < lim && !(a4[u] < b4[u])) first_invalid = min(first_invalid, k);
        }
      }
      if (first_invalid < lim) atomicMin(&co_k, first_invalid);
    }
    __syncthreads();
    const unsigned K = co_k;
    for (unsigned k0 = threadIdx.x; k0 < K; k0 += 4 * kThreads) {
      unsigned il[4], ir[4];
      unsigned long long xl[4], xr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned k = k0 + u * kThreads;
        il[u] = k < K ? stops_l[k] : 0u;
        ir[u] = k < K ? stops_r[cnt_r - 1u - k] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xl[u] = arr[il[u]];
        xr[u] = arr[ir[u]];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (k0 + u * kThreads < K) {
          arr[il[u]] = xr[u];
          arr[ir[u]] = xl[u];
        }
    }
    if (threadIdx.x == 0) {
      unsigned c = 0x7fffffffu;  // where the left pointer stops next
      if (K < cnt_l) c = stops_l[K];
      if (K > 0u) c = min(c, stops_r[cnt_r - K]);
      co_cut = static_cast<int>(c);
    }
    __syncthreads();
    const int cut = co_cut;
    // (d) [first, cut) and [cut, last): on the list if they are above the threshold and hold two tied elements
    {
      const int n_all = last - first;
      const int pw = ((n_all + (kThreads / 64) * 64 - 1) / ((kThreads / 64) * 64)) * 64;
      const int lo_w = first + wave * pw, hi_w = min(last, lo_w + pw);
      unsigned tl = 0u, tr = 0u;
      for (int base = lo_w; base < hi_w; base += 256) {
        unsigned id4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int p = base + 64 * u + lane;
          id4[u] = p < hi_w ? static_cast<unsigned>(arr[p]) : 0u;
        }
        unsigned char t4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t4[u] = base + 64 * u + lane < hi_w ? tied[id4[u]] : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int p = base + 64 * u + lane;
          tl += __builtin_popcountll(__builtin_amdgcn_ballot_w64(t4[u] != 0 && p < cut));
          tr += __builtin_popcountll(__builtin_amdgcn_ballot_w64(t4[u] != 0 && p >= cut));
        }
      }
      if (lane == 0) {
        atomicAdd(&co_tied[0], tl);
        atomicAdd(&co_tied[1], tr);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      // replace the entry by its children that still matter
      int n = wl_n;
      wl_first[wl_pick] = wl_first[n - 1];
      wl_last[wl_pick] = wl_last[n - 1];
      wl_depth[wl_pick] = wl_depth[n - 1];
      --n;
      const int cf[2] = {first, cut}, cl2[2] = {cut, last};
      for (int c = 0; c < 2; ++c)
        if (cl2[c] - cf[c] > 16 && co_tied[c] >= 2u) {
          if (n < kWorkListCap) {
            wl_first[n] = cf[c];
            wl_last[n] = cl2[c];
            wl_depth[n] = depth - 1;
            ++n;
          } else {
            wl_overflow = 1;
          }
        }
      wl_n = n;
    }
    __syncthreads();
  }
  if (wl_overflow != 0) ok = false;
  if (!ok) return false;
  DLIOM_SSTAMP(13);
  // where the tied elements are in the arrangement
  for (int q0 = static_cast<int>(threadIdx.x); q0 < m; q0 += 8 * kThreads) {
    unsigned id8[8];
    unsigned char f8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) id8[u] = static_cast<unsigned>(arr[min(q0 + u * kThreads, m - 1)]);
#pragma unroll
    for (int u = 0; u < 8; ++u) f8[u] = tied[id8[u]];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (q0 + u * kThreads < m && f8[u]) pos_of[id8[u]] = static_cast<unsigned>(q0 + u * kThreads);
  }
  __syncthreads();
  // the final insertion sort is stable: a group of equal keys ends up in arrangement order
  for (int j0 = static_cast<int>(threadIdx.x); j0 < m; j0 += 8 * kThreads) {
    unsigned id8[8];
    unsigned char f8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) id8[u] = sv[min(j0 + u * kThreads, m - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) f8[u] = tied[id8[u]];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + u * kThreads;
      if (j >= m) continue;
      const unsigned id = id8[u];
      {
        unsigned dst = static_cast<unsigned>(j);
        if (f8[u]) {
          const unsigned key = static_cast<unsigned>(sk[j]);
          int gs = j, ge = j + 1;
          while (gs > 0 && static_cast<unsigned>(sk[gs - 1]) == key) --gs;
          while (ge < m && static_cast<unsigned>(sk[ge]) == key) ++ge;
          const unsigned mine = pos_of[id];
          unsigned r = 0u;
          for (int w2 = gs; w2 < ge; ++w2) r += pos_of[sv[w2]] < mine ? 1u : 0u;
          dst = static_cast<unsigned>(gs) + r;
        }
        sorted_id[dst] = id;
      }
    }
  }
  __syncthreads();
  DLIOM_SSTAMP(14);
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
  // (eight positions per thread at a time, position = tid + 1024 u: consecutive lanes on consecutive entries and all loads
  // of a stage in flight together -- a plain loop over the thread's positions pays the memory latency once per position)
  for (int j0 = static_cast<int>(threadIdx.x); j0 < m; j0 += 8 * kThreads) {
    unsigned id[8];
    float x[8], y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) id[u] = j0 + u * kThreads < m ? sorted_id[j0 + u * kThreads] : 0u;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      x[u] = bx[id[u]];
      y[u] = by[id[u]];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (j0 + u * kThreads < m) {
        px[j0 + u * kThreads] = x[u];
        py[j0 + u * kThreads] = y[u];
      }
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
  // the chain's arrays (next pointers twice, dead and mark bytes) live in LDS when the slice is small enough for that
  // (15 000 points: the floor of a filtered 64-beam scan), else in HBM
  // (19 000 points: next and x1 of the blocked walk below as 16-bit positions, dead and mark bytes -- 6 bytes a point,
  // 8 reserved)
  const bool chain_in_lds = (static_cast<size_t>(m) + 4) * 8 + 64 <= kBigLdsBytes;
  const int m4 = (m + 4) & ~3;
  unsigned short* nx16 = reinterpret_cast<unsigned short*>(big_lds);  // LDS: next(i)
  unsigned short* x1 = nx16 + m4;
  unsigned* ja = A.jump_a + off;  // HBM: next(i), squared level by level
  unsigned* jb = A.jump_b + off;
  unsigned char* dead = chain_in_lds ? reinterpret_cast<unsigned char*>(x1 + m4) : A.dead + off;
  unsigned char* mark = chain_in_lds ? dead + m4 : A.mark + off;
  __shared__ unsigned short walk_entry_t[kThreads];  // where the path enters a block of the walk (m: it does not)
  __syncthreads();  // (the exact sums' scratch lies under these arrays)
  for (int j0 = static_cast<int>(threadIdx.x); j0 < m; j0 += 8 * kThreads) {
    float x[8], y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = min(j0 + u * kThreads, m - 1);
      x[u] = px[j];
      y[u] = py[j];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + u * kThreads;
      if (j < m) {
        dead[j] = norm2(x[u] - cx, y[u] - cy) < kMinDistance ? 1 : 0;
        mark[j] = (j == 0 && !chain_in_lds) ? 1 : 0;
      }
    }
  }
  if (threadIdx.x == 0) {
    mark[m] = 0;
    if (chain_in_lds) {
      nx16[m] = static_cast<unsigned short>(m);
      x1[m] = static_cast<unsigned short>(m);
    } else {
      ja[m] = static_cast<unsigned>(m);
      jb[m] = static_cast<unsigned>(m);
    }
  }
  walk_entry_t[threadIdx.x] = static_cast<unsigned short>(m);
  __syncthreads();
  // next(i): the first live point farther than kMaxDistance from point i (squared lengths: rotational_histogram.hip).
  // On a floor nearly every point's answer is i + 1: that candidate is tested for eight positions at once, the others
  // walk on alone.
  for (int i0 = static_cast<int>(threadIdx.x); i0 < m; i0 += 8 * kThreads) {
    float ax[8], ay[8], nx[8], ny[8];
    unsigned char nd[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(i0 + u * kThreads, m - 1), j = min(i + 1, m - 1);
      ax[u] = px[i];
      ay[u] = py[i];
      nx[u] = px[j];
      ny[u] = py[j];
      nd[u] = dead[j];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * kThreads;
      if (i < m) {
        int j = i + 1;
        const float dx0 = nx[u] - ax[u], dy0 = ny[u] - ay[u];
        if (!(j < m && nd[u] == 0 && dx0 * dx0 + dy0 * dy0 >= squared_jump)) {
          for (j = min(i + 1, m); j < m; ++j) {
            if (dead[j]) continue;
            const float dx = px[j] - ax[u], dy = py[j] - ay[u];
            if (dx * dx + dy * dy >= squared_jump) break;
          }
        }
        if (chain_in_lds) nx16[i] = static_cast<unsigned short>(j);
        else ja[i] = static_cast<unsigned>(j);
      }
    }
  }
  __syncthreads();
  DLIOM_BSTAMP(4);
  if (chain_in_lds) {
    // ---- the nodes of the path 0 -> next(0) -> ... by a BLOCKED WALK (tests/cpp/chain_walk_model.cc is this in plain
    //      C++ against the plain walk).  next(i) > i: the path only moves forward.  Squaring all m pointers
    //      ceil(lg m) times (pointer doubling, below: what the arrays in HBM still get) was 50 us of a 10 000-point floor
    //      slice -- fifteen passes over every position, bound by instruction issue.  Here the positions are cut into
    //      blocks of kWalkBlock; (1) one thread per block, positions from the last to the first: x1(i) = the first path
    //      node at or behind the block's end (next(i) if that already is, else x1(next(i))); (2) ONE thread hops from
    //      block to block with x1 and notes where the path enters each; (3) one thread per block marks from there with
    //      next.  ~2 * 64 + m / 64 dependent LDS accesses instead of 15 passes.  (A wave level in between -- x2(i), the
    //      first node behind the WAVE's 64 blocks, thread blocks of ceil(m / 1024) positions -- was built first and was
    //      slower than the doubling: its sweep has one lane of every wave active per step, and sixteen waves issuing
    //      64 * 11 predicated steps are 90 000 cycles of a compute unit's instruction issue.)
    constexpr int kWalkBlock = 64;
    static_assert(kThreads * kWalkBlock >= 19200, "one thread per block");
    const int b_lo = min(m, static_cast<int>(threadIdx.x) * kWalkBlock), b_hi = min(m, b_lo + kWalkBlock);
    for (int i = b_hi - 1; i >= b_lo; --i) {
      const unsigned v = nx16[i];
      x1[i] = static_cast<unsigned short>(static_cast<int>(v) >= b_hi ? v : x1[v]);
    }
    __syncthreads();
    if (threadIdx.x == 0)
      for (int e = 0; e < m; e = x1[e]) walk_entry_t[e / kWalkBlock] = static_cast<unsigned short>(e);
    __syncthreads();
    for (int i = walk_entry_t[threadIdx.x]; i < b_hi; i = nx16[i]) mark[i] = 1;
    __syncthreads();
  }
  // marks spread along next^(2^d) while the pointers are squared (arrays in HBM)
  for (int d = 0; !chain_in_lds && (1 << d) < 2 * m; ++d) {
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
  // last_point of point j = the last marked position before it (position 0 to begin with); thread t owns [lo, hi) here:
  // the contributions keep the order of the points
  int last_marked = -1;
  for (int j = lo; j < hi; ++j)
    if (mark[j]) last_marked = j;
  const int anchor0 = max(0, block_exclusive_max(last_marked, -1, wave_max));
  constexpr int kOwn = 16;  // owned positions handled in registers (slices up to 16 384 points); more: the plain loop
  unsigned emitted = 0u;
  // (the choice must be the same for every thread: both branches hold the workgroup's prefix scan, and the last thread
  // with points owns fewer than the others -- chosen per thread, its wave ran both branches, barriers and all, one after
  // the other and that thread's contributions landed on top of its wave's first ones: slices above 16 384 points whose
  // last owner had any, found in round 4 on the 20 090-point floor of a noise-free 128 x 2048 scan)
  if ((m + kThreads - 1) / kThreads <= kOwn) {
    int anchor_of[kOwn];
    bool live[kOwn];
    {
      int a = anchor0;
#pragma unroll
      for (int u = 0; u < kOwn; ++u) {
        const int j = lo + u;
        anchor_of[u] = a;
        live[u] = false;
        if (j < hi) {
          const bool mk = mark[j] != 0;
          live[u] = dead[j] == 0 && !(mk && j != 0);
          if (mk) a = j;
        }
      }
    }
    float xj[kOwn], yj[kOwn], xa[kOwn], ya[kOwn];
#pragma unroll
    for (int u = 0; u < kOwn; ++u) {
      const int j = min(lo + u, m - 1);
      xj[u] = px[j];
      yj[u] = py[j];
      xa[u] = px[anchor_of[u]];
      ya[u] = py[anchor_of[u]];
    }
    float dist[kOwn];
#pragma unroll
    for (int u = 0; u < kOwn; ++u) {
      dist[u] = norm2(xj[u] - xa[u], yj[u] - ya[u]);
      live[u] = live[u] && !(dist[u] < kMinDistance);
      emitted += live[u] ? 1u : 0u;
    }
    unsigned total;
    unsigned at = s.begin + block_exclusive_scan(emitted, wave_sums, &total);
#pragma unroll
    for (int u = 0; u < kOwn; ++u)
      if (live[u]) {
        const float dx = xj[u] - xa[u], dy = yj[u] - ya[u];
        const float ex = xj[u] - cx, ey = yj[u] - cy;
        const float direction_norm = norm2(ex, ey);
        const float dot = (dx / dist[u]) * (ex / direction_norm) + (dy / dist[u]) * (ey / direction_norm);
        c_bucket[at] = static_cast<unsigned char>(bucket_of(fd_atan2f(dy, dx), histogram_size));
        c_value[at] = fmaxf(0.f, 1.f - fabsf(dot));
        ++at;
      }
  } else {
    // two passes over the owned points: count, then write in order
    for (int pass = 0; pass < 2; ++pass) {
      int a = anchor0;
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
