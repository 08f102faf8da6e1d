// Device-resident HybridGrid and range-data insertion for gfx950.
//
// Layout in HBM (DESIGN.md "grid"):
//   table : (8<<bits)^3 uint32, z-major over LEAF coordinates shifted by 4<<bits;
//           entry = pool slot of that 8x8x8 leaf, 0 = not allocated.
//   pool  : slot * 1 KiB, each leaf 512 uint16 in the reference's z-major order
//           (mapping/3d/hybrid_grid.h:40-43); slot 0 is a permanent all-zero
//           leaf so lookups need no "missing" branch.
// This flattens the reference's DynamicGrid -> NestedGrid pointer levels
// (hybrid_grid.h:143-409) into one dependent load while keeping its 1 KiB
// leaves, its index range [-32<<bits, 32<<bits) and its growth rule.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "device_common.h"

namespace dliom {

constexpr uint32_t kSlotLocked = 0xFFFFFFFFu;
constexpr int kMaxFlatBits = 8;  // DynamicGrid's own limit (hybrid_grid.h:389); (8<<8)^3 entries = 2^33: 64-bit index math
constexpr int kMaxDenseBits = 4; // dense mirror: (64<<4 + 2)^3 * 2 B = 2.0 GiB

__device__ __forceinline__ bool leaf_table_index(int ix, int iy, int iz, int half, unsigned gsize,
                                                 unsigned L, size_t* tidx, unsigned* cell) {
  const unsigned sx = static_cast<unsigned>(ix + half);
  const unsigned sy = static_cast<unsigned>(iy + half);
  const unsigned sz = static_cast<unsigned>(iz + half);
  if (!((sx < gsize) & (sy < gsize) & (sz < gsize))) return false;
  *tidx = (static_cast<size_t>(sz >> 3) * L + (sy >> 3)) * L + (sx >> 3);  // up to 2^33 entries at bits = 8
  *cell = ((sz & 7u) << 6) | ((sy & 7u) << 3) | (sx & 7u);
  return true;
}

// Makes sure the leaf that holds (ix,iy,iz) exists.  Exactly one thread wins the
// CAS and takes the next pool slot; the others need nothing from this kernel
// (later kernels read the table after the kernel boundary).
__device__ __forceinline__ void ensure_leaf(uint32_t* table, int32_t* slot_coord, uint32_t* count,
                                            int ix, int iy, int iz, int half, unsigned gsize,
                                            unsigned L) {
  size_t tidx;
  unsigned cell;
  if (!leaf_table_index(ix, iy, iz, half, gsize, L, &tidx, &cell)) return;
  if (table[tidx] != 0u) return;
  if (atomicCAS(&table[tidx], 0u, kSlotLocked) == 0u) {
    const uint32_t slot = atomicAdd(count, 1u);
    slot_coord[3 * static_cast<size_t>(slot) + 0] = ix >> 3;  // arithmetic shift == floor(/8)
    slot_coord[3 * static_cast<size_t>(slot) + 1] = iy >> 3;
    slot_coord[3 * static_cast<size_t>(slot) + 2] = iz >> 3;
    atomicExch(&table[tidx], slot);
  }
}

// HybridGrid::ApplyLookupTable (hybrid_grid.h:509-520) on one 16-bit cell: cells that already carry the update
// marker are left alone, so the result does not depend on which thread gets there first.
// Returns the value written (with the marker bit) or 0 when the cell was already updated.
__device__ __forceinline__ uint32_t apply_table(uint32_t* pool32, size_t value_index,
                                                const uint16_t* __restrict__ lut) {
  // No atomic is needed: a thread that finds the cell unmarked writes lut[old] (marker included), and every thread that
  // finds it unmarked -- racing threads, threads on another XCD whose L2 still holds the old line -- writes that SAME
  // value from the SAME old value; a thread that finds the marker leaves the cell alone.  The 16-bit store touches only
  // this cell's two bytes, so the neighbour in the same dword is not involved (the 32-bit compare-and-swap this replaces
  // was: device-scope atomics execute on the memory side at ~30 G/s, and they were the whole cost of the three passes).
  volatile uint16_t* cell = reinterpret_cast<volatile uint16_t*>(pool32) + value_index;
  const uint32_t v = *cell;
  if (v >= 0x8000u) return 0u;
  const uint32_t nv = lut[v];
  *cell = static_cast<uint16_t>(nv);
  return nv;
}
// FinishUpdate for one cell (hybrid_grid.h:494-500): drops the marker; idempotent, plain 16-bit accesses as above.
__device__ __forceinline__ void clear_marker(uint32_t* pool32, size_t value_index) {
  volatile uint16_t* cell = reinterpret_cast<volatile uint16_t*>(pool32) + value_index;
  const uint16_t v = *cell;
  if (v & 0x8000u) *cell = v & 0x7FFFu;
}

struct InsertArgs {
  const float* returns;  // packed xyz
  int64_t n;             // upper bound of the number of returns
  const unsigned* n_dev; // if non-null the actual number (<= n) lives on the device
  float ox, oy, oz;      // origin
  float resolution;
  int num_free;
  int half;
  unsigned gsize;
  unsigned L;
  uint16_t* dense;  // write-through dense mirror of the grid (may be null), see ensure_dense()
  int dense_stride; // cells per axis of the mirror
  int dense_off[3]; // mirror coordinate = cell index + dense_off (a windowed mirror does not hold every cell)
};

// Index into the dense mirror: mirror coordinate = cell index + off per axis (the whole-grid mirror: half + 1, one guard
// cell); 4 x 4 x 4 bricks of 64 cells (one 128-byte cache line), bricks z-major with B = ceil(stride / 4) per axis.
// false: the cell lies outside the mirror (a windowed mirror; the matcher never reads there).
__device__ __forceinline__ bool dense_index(int ix, int iy, int iz, const int (&off)[3], int stride, size_t* index) {
  const unsigned x = static_cast<unsigned>(ix + off[0]), y = static_cast<unsigned>(iy + off[1]), z = static_cast<unsigned>(iz + off[2]);
  const unsigned S = static_cast<unsigned>(stride);
  if (x >= S || y >= S || z >= S) return false;
  const size_t B = static_cast<size_t>((stride + 3) >> 2);
  *index = (((z >> 2) * B + (y >> 2)) * B + (x >> 2)) * 64 + (((z & 3u) << 4) | ((y & 3u) << 2) | (x & 3u));
  return true;
}

// range_data_inserter_3d.cc:36-50: the k-th sample on the ray origin->hit in
// Array3i arithmetic (int multiply, truncating int division).
__device__ __forceinline__ void miss_cell(int ocx, int ocy, int ocz, int dx, int dy, int dz,
                                          int position, int num_samples, int* mx, int* my,
                                          int* mz) {
  *mx = ocx + dx * position / num_samples;
  *my = ocy + dy * position / num_samples;
  *mz = ocz + dz * position / num_samples;
}

__device__ __forceinline__ int bits_for(int c) {
  // smallest b >= 1 with -(32<<b) <= c < (32<<b)
  int b = 1;
  while (b < 9 && !(c >= -(32 << b) && c < (32 << b))) ++b;
  return b;
}

// Pass 0: the DynamicGrid bits the reference would end up with, and the
// CHECK_LT(num_samples, 1<<15) condition.  out[0] = max needed bits, out[1] = ray too long.
__global__ void insert_scan_kernel(InsertArgs a, int* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= (a.n_dev != nullptr ? static_cast<int64_t>(*a.n_dev) : a.n)) return;
  const int hx = cell_of(a.returns[3 * i], a.resolution);
  const int hy = cell_of(a.returns[3 * i + 1], a.resolution);
  const int hz = cell_of(a.returns[3 * i + 2], a.resolution);
  int need = max(bits_for(hx), max(bits_for(hy), bits_for(hz)));
  const int ocx = cell_of(a.ox, a.resolution), ocy = cell_of(a.oy, a.resolution),
            ocz = cell_of(a.oz, a.resolution);
  const int dx = hx - ocx, dy = hy - ocy, dz = hz - ocz;
  const int num_samples = max(abs(dx), max(abs(dy), abs(dz)));
  if (num_samples >= (1 << 15)) atomicMax(&out[1], 1);
  for (int position = max(0, num_samples - a.num_free); position < num_samples; ++position) {
    int mx, my, mz;
    miss_cell(ocx, ocy, ocz, dx, dy, dz, position, num_samples, &mx, &my, &mz);
    need = max(need, max(bits_for(mx), max(bits_for(my), bits_for(mz))));
  }
  atomicMax(&out[0], need);
}

// Pass 1: allocate every leaf that a hit or miss cell touches (what
// mutable_value() does lazily, hybrid_grid.h:285-301,167-177).
__global__ void insert_alloc_kernel(InsertArgs a, uint32_t* table, int32_t* slot_coord,
                                    uint32_t* count) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= (a.n_dev != nullptr ? static_cast<int64_t>(*a.n_dev) : a.n)) return;
  const int hx = cell_of(a.returns[3 * i], a.resolution);
  const int hy = cell_of(a.returns[3 * i + 1], a.resolution);
  const int hz = cell_of(a.returns[3 * i + 2], a.resolution);
  ensure_leaf(table, slot_coord, count, hx, hy, hz, a.half, a.gsize, a.L);
  const int ocx = cell_of(a.ox, a.resolution), ocy = cell_of(a.oy, a.resolution),
            ocz = cell_of(a.oz, a.resolution);
  const int dx = hx - ocx, dy = hy - ocy, dz = hz - ocz;
  const int num_samples = max(abs(dx), max(abs(dy), abs(dz)));
  for (int position = max(0, num_samples - a.num_free); position < num_samples; ++position) {
    int mx, my, mz;
    miss_cell(ocx, ocy, ocz, dx, dy, dz, position, num_samples, &mx, &my, &mz);
    ensure_leaf(table, slot_coord, count, mx, my, mz, a.half, a.gsize, a.L);
  }
}

// Pass 2 (MODE 0): hits.  Pass 3 (MODE 1): misses.  Pass 4 (MODE 2): FinishUpdate
// -- clear the marker on every touched cell (hybrid_grid.h:494-500).
template <int MODE>
__global__ void insert_apply_kernel(InsertArgs a, const uint32_t* __restrict__ table,
                                    uint32_t* pool32, const uint16_t* __restrict__ lut) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= (a.n_dev != nullptr ? static_cast<int64_t>(*a.n_dev) : a.n)) return;
  const int hx = cell_of(a.returns[3 * i], a.resolution);
  const int hy = cell_of(a.returns[3 * i + 1], a.resolution);
  const int hz = cell_of(a.returns[3 * i + 2], a.resolution);
  size_t tidx;
  unsigned cell;
  if (MODE == 0 || MODE == 2) {
    if (leaf_table_index(hx, hy, hz, a.half, a.gsize, a.L, &tidx, &cell)) {
      const size_t vi = static_cast<size_t>(table[tidx]) * 512u + cell;
      if (MODE == 0) {
        const uint32_t nv = apply_table(pool32, vi, lut);
        // each cell is written at most once per Insert: the mirror takes the final value
        if (nv != 0u && a.dense != nullptr) {
            size_t di;
            if (dense_index(hx, hy, hz, a.dense_off, a.dense_stride, &di)) a.dense[di] = static_cast<uint16_t>(nv & 0x7FFFu);
          }
      } else {
        clear_marker(pool32, vi);
      }
    }
  }
  if (MODE == 1 || MODE == 2) {
    const int ocx = cell_of(a.ox, a.resolution), ocy = cell_of(a.oy, a.resolution),
              ocz = cell_of(a.oz, a.resolution);
    const int dx = hx - ocx, dy = hy - ocy, dz = hz - ocz;
    const int num_samples = max(abs(dx), max(abs(dy), abs(dz)));
    for (int position = max(0, num_samples - a.num_free); position < num_samples; ++position) {
      int mx, my, mz;
      miss_cell(ocx, ocy, ocz, dx, dy, dz, position, num_samples, &mx, &my, &mz);
      if (!leaf_table_index(mx, my, mz, a.half, a.gsize, a.L, &tidx, &cell)) continue;
      const size_t vi = static_cast<size_t>(table[tidx]) * 512u + cell;
      if (MODE == 1) {
        const uint32_t nv = apply_table(pool32, vi, lut);
        if (nv != 0u && a.dense != nullptr) {
            size_t di;
            if (dense_index(mx, my, mz, a.dense_off, a.dense_stride, &di)) a.dense[di] = static_cast<uint16_t>(nv & 0x7FFFu);
          }
      } else {
        clear_marker(pool32, vi);
      }
    }
  }
}

// Rebuilds the leaf table for a new extent from the per-slot leaf coordinates
// (the counterpart of DynamicGrid::Grow, hybrid_grid.h:387-405).
__global__ void rebuild_table_kernel(const int32_t* __restrict__ slot_coord, uint32_t count,
                                     uint32_t* table, int leaf_half, unsigned L) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x + 1u;  // slot 0 is the null leaf
  if (s >= count) return;
  const unsigned lx = static_cast<unsigned>(slot_coord[3 * static_cast<size_t>(s)] + leaf_half);
  const unsigned ly = static_cast<unsigned>(slot_coord[3 * static_cast<size_t>(s) + 1] + leaf_half);
  const unsigned lz = static_cast<unsigned>(slot_coord[3 * static_cast<size_t>(s) + 2] + leaf_half);
  table[(static_cast<size_t>(lz) * L + ly) * L + lx] = s;
}

// Dense mirror (re)build: one workgroup per allocated leaf copies its 512 cells.
struct DenseOff {
  int v[3];
};
__global__ void dense_fill_kernel(const int32_t* __restrict__ slot_coord, const uint16_t* __restrict__ pool,
                                  uint16_t* __restrict__ dense, DenseOff off, int stride) {
  const size_t s = static_cast<size_t>(blockIdx.x) + 1;  // slot 0 is the null leaf
  const int bx = slot_coord[3 * s] * 8, by = slot_coord[3 * s + 1] * 8, bz = slot_coord[3 * s + 2] * 8;
  for (int c = threadIdx.x; c < 512; c += blockDim.x) {
    const uint16_t v = pool[s * 512 + c] & 0x7FFFu;
    size_t di;
    if (dense_index(bx + (c & 7), by + ((c >> 3) & 7), bz + (c >> 6), off.v, stride, &di)) dense[di] = v > 0 ? v : 1;
  }
}

__global__ void upload_alloc_kernel(const int32_t* __restrict__ origins, int64_t n, uint32_t* table,
                                    int32_t* slot_coord, uint32_t* count, int half, unsigned gsize,
                                    unsigned L) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ensure_leaf(table, slot_coord, count, origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], half,
              gsize, L);
}

__global__ void upload_copy_kernel(const int32_t* __restrict__ origins,
                                   const uint16_t* __restrict__ values, const uint32_t* __restrict__ table,
                                   uint16_t* pool, int half, unsigned gsize, unsigned L) {
  const int64_t b = blockIdx.x;
  size_t tidx;
  unsigned cell;
  if (!leaf_table_index(origins[3 * b], origins[3 * b + 1], origins[3 * b + 2], half, gsize, L, &tidx,
                        &cell))
    return;
  const size_t slot = table[tidx];
  for (int k = threadIdx.x; k < 512; k += blockDim.x)
    pool[slot * 512u + k] = values[static_cast<size_t>(b) * 512u + k];
}

__global__ void set_alloc_kernel(const int32_t* __restrict__ cells, int64_t n, uint32_t* table,
                                 int32_t* slot_coord, uint32_t* count, int half, unsigned gsize,
                                 unsigned L) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ensure_leaf(table, slot_coord, count, cells[3 * i], cells[3 * i + 1], cells[3 * i + 2], half, gsize, L);
}

__global__ void set_values_kernel(const int32_t* __restrict__ cells, const uint16_t* __restrict__ values,
                                  int64_t n, const uint32_t* __restrict__ table, uint16_t* pool, int half,
                                  unsigned gsize, unsigned L) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  size_t tidx;
  unsigned cell;
  if (!leaf_table_index(cells[3 * i], cells[3 * i + 1], cells[3 * i + 2], half, gsize, L, &tidx, &cell)) return;
  pool[static_cast<size_t>(table[tidx]) * 512u + cell] = values[i];
}

__global__ void get_values_kernel(GridView g, const int32_t* __restrict__ cells, int64_t n,
                                  uint16_t* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = static_cast<uint16_t>(grid_value(g, cells[3 * i], cells[3 * i + 1], cells[3 * i + 2]));
}

// sensor::TransformRangeData (sensor/range_data.cc:25-33) through up to two float poses applied
// in sequence, then FilterRangeDataByMaxRange (mapping/3d/submap_3d.cc:42-51): keeps returns
// with ||hit - origin|| <= max_range (Eigen norm order x*x + (y*y + z*z)).  Survivors are
// compacted in arbitrary order -- insertion is order independent.
struct TransformArgs {
  Quat4 q[2];
  float t[2][3];
  int num_poses;
  float ox, oy, oz;   // origin AFTER the transforms
  float max_range;    // <= 0: keep everything
};
__global__ void transform_filter_kernel(const float* __restrict__ px, const float* __restrict__ py,
                                        const float* __restrict__ pz, int64_t n, TransformArgs a,
                                        float* __restrict__ out, unsigned* __restrict__ count) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = px[i], y = py[i], z = pz[i];
  for (int k = 0; k < a.num_poses; ++k) {
    float rx, ry, rz;
    rotate_point(a.q[k], x, y, z, rx, ry, rz);
    x = rx + a.t[k][0];
    y = ry + a.t[k][1];
    z = rz + a.t[k][2];
  }
  if (a.max_range > 0.f) {
    const float dx = x - a.ox, dy = y - a.oy, dz = z - a.oz;
    const float norm = sqrtf(dx * dx + (dy * dy + dz * dz));
    if (!(norm <= a.max_range)) return;
  }
  const unsigned k = atomicAdd(count, 1u);
  out[3 * static_cast<size_t>(k)] = x;
  out[3 * static_cast<size_t>(k) + 1] = y;
  out[3 * static_cast<size_t>(k) + 2] = z;
}

// ---- fused insertion of one HBM-resident cloud into up to 4 grids ---------------------------------
// (Submap3D::InsertRangeData for both grids of both active submaps, mapping/3d/submap_3d.cc:264-279,
// 303-309.)  Every kernel recomputes a target's transform chain and range filter per point instead of
// materialising transformed copies; blockIdx.y selects the target.  The host launches all five passes
// back to back and synchronises ONCE: passes 1-4 skip a target whose pass-0 status says it needs the
// host first (extent growth or CHECK failure); the host then grows and re-runs just that target.
constexpr int kMaxInsertTargets = 4;
struct InsertTarget {
  Quat4 q[2];
  float t[2][3];
  int num_poses;
  float ox, oy, oz;   // range_data.origin after the transforms
  float max_range;    // <= 0: no range filter
  float resolution;
  int num_free;
  int bits;
  int half;
  unsigned gsize, L;
  uint32_t* table;
  int32_t* slot_coord;
  uint32_t* count;
  uint32_t* pool32;
  uint16_t* dense;
  int dense_stride;
  int dense_off[3];
  unsigned* count_slot;  // pinned [seq, count] of the target's grid (null: none), written by the last pass
  unsigned count_seq;
};
struct MultiInsertArgs {
  InsertTarget tg[kMaxInsertTargets];
  unsigned run_mask;  // bit k: process target k in this launch
  const float* px;
  const float* py;
  const float* pz;
  int n;
  const uint16_t* hit;
  const uint16_t* miss;
  int* status;  // per target: [needed bits, ray too long]
  int* host_status;  // pinned host copy of `status`, written by pass 1 (null: not wanted) -- saves a copy dispatch
  unsigned* done_word;  // ... followed by this completion word (pinned; null: none), which the host polls
  unsigned done_seq;
};

// Transformed + range-filtered hit of point i for target tg; false if filtered out.
__device__ __forceinline__ bool target_hit(const MultiInsertArgs& a, const InsertTarget& tg, int i, int* hx,
                                           int* hy, int* hz) {
  float x = a.px[i], y = a.py[i], z = a.pz[i];
  for (int k = 0; k < tg.num_poses; ++k) {
    float rx, ry, rz;
    rotate_point(tg.q[k], x, y, z, rx, ry, rz);
    x = rx + tg.t[k][0];
    y = ry + tg.t[k][1];
    z = rz + tg.t[k][2];
  }
  if (tg.max_range > 0.f) {
    const float dx = x - tg.ox, dy = y - tg.oy, dz = z - tg.oz;
    if (!(sqrtf(dx * dx + (dy * dy + dz * dz)) <= tg.max_range)) return false;
  }
  *hx = cell_of(x, tg.resolution);
  *hy = cell_of(y, tg.resolution);
  *hz = cell_of(z, tg.resolution);
  return true;
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}

template <int PASS>  // 0 scan, 1 alloc, 2 hits, 3 misses, 4 finish
__global__ void multi_insert_kernel(MultiInsertArgs a) {
  const int tgi = blockIdx.y;
  if (!((a.run_mask >> tgi) & 1u)) return;
  const InsertTarget& tg = a.tg[tgi];
  // pass 0 has finished: its verdict goes to the host from here (the passes that follow are skipped per target)
  if (PASS == 1 && a.host_status != nullptr && blockIdx.x == 0 && tgi == __ffs(a.run_mask) - 1) {  // uniform over the block
    if (threadIdx.x < 2 * kMaxInsertTargets) {
      a.host_status[threadIdx.x] = a.status[threadIdx.x];
      __threadfence_system();
    }
    __syncthreads();
    if (threadIdx.x == 0 && a.done_word != nullptr) {
      __threadfence_system();
      *reinterpret_cast<volatile unsigned*>(a.done_word) = a.done_seq;
    }
  }
  if (PASS > 0 && (a.status[2 * tgi] > tg.bits || a.status[2 * tgi + 1] != 0)) return;  // host first
  if (PASS == 4 && tg.count_slot != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
    // the leaf count is final since pass 1: (count, then sequence number) for the host's next ensure_capacity
    volatile unsigned* slot = tg.count_slot;
    slot[1] = *tg.count;
    __threadfence_system();
    slot[0] = tg.count_seq;
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int hx = 0, hy = 0, hz = 0;
  const bool valid = i < a.n && target_hit(a, tg, i, &hx, &hy, &hz);
  const int ocx = cell_of(tg.ox, tg.resolution), ocy = cell_of(tg.oy, tg.resolution),
            ocz = cell_of(tg.oz, tg.resolution);
  const int dx = hx - ocx, dy = hy - ocy, dz = hz - ocz;
  const int num_samples = max(abs(dx), max(abs(dy), abs(dz)));
  const int first = max(0, num_samples - tg.num_free);
  if (PASS == 0) {
    int need = 0, too_long = 0;
    if (valid) {
      need = max(bits_for(hx), max(bits_for(hy), bits_for(hz)));
      too_long = num_samples >= (1 << 15) ? 1 : 0;
      for (int position = first; position < num_samples; ++position) {
        int mx, my, mz;
        miss_cell(ocx, ocy, ocz, dx, dy, dz, position, num_samples, &mx, &my, &mz);
        need = max(need, max(bits_for(mx), max(bits_for(my), bits_for(mz))));
      }
    }
    need = wave_max_i32(need);
    too_long = wave_max_i32(too_long);
    // the host only asks "more bits than the grid has?": in the steady state no wavefront reports anything (every
    // wavefront's atomicMax on this one word used to be ~85 % of the pass: same-address atomics serialise in L2)
    if ((threadIdx.x & 63) == 0) {
      if (need > tg.bits) atomicMax(&a.status[2 * tgi], need);
      if (too_long) atomicMax(&a.status[2 * tgi + 1], 1);
    }
    return;
  }
  if (!valid) return;
  size_t tidx;
  unsigned cell;
  if (PASS == 1) {
    ensure_leaf(tg.table, tg.slot_coord, tg.count, hx, hy, hz, tg.half, tg.gsize, tg.L);
    for (int position = first; position < num_samples; ++position) {
      int mx, my, mz;
      miss_cell(ocx, ocy, ocz, dx, dy, dz, position, num_samples, &mx, &my, &mz);
      ensure_leaf(tg.table, tg.slot_coord, tg.count, mx, my, mz, tg.half, tg.gsize, tg.L);
    }
    return;
  }
  if (PASS == 2 || PASS == 4) {
    if (leaf_table_index(hx, hy, hz, tg.half, tg.gsize, tg.L, &tidx, &cell)) {
      const size_t vi = static_cast<size_t>(tg.table[tidx]) * 512u + cell;
      if (PASS == 2) {
        const uint32_t nv = apply_table(tg.pool32, vi, a.hit);
        if (nv != 0u && tg.dense != nullptr) {
            size_t di;
            if (dense_index(hx, hy, hz, tg.dense_off, tg.dense_stride, &di)) tg.dense[di] = static_cast<uint16_t>(nv & 0x7FFFu);
          }
      } else {
        clear_marker(tg.pool32, vi);
      }
    }
  }
  if (PASS == 3 || PASS == 4) {
    for (int position = first; position < num_samples; ++position) {
      int mx, my, mz;
      miss_cell(ocx, ocy, ocz, dx, dy, dz, position, num_samples, &mx, &my, &mz);
      if (!leaf_table_index(mx, my, mz, tg.half, tg.gsize, tg.L, &tidx, &cell)) continue;
      const size_t vi = static_cast<size_t>(tg.table[tidx]) * 512u + cell;
      if (PASS == 3) {
        const uint32_t nv = apply_table(tg.pool32, vi, a.miss);
        if (nv != 0u && tg.dense != nullptr) {
            size_t di;
            if (dense_index(mx, my, mz, tg.dense_off, tg.dense_stride, &di)) tg.dense[di] = static_cast<uint16_t>(nv & 0x7FFFu);
          }
      } else {
        clear_marker(tg.pool32, vi);
      }
    }
  }
}

static inline unsigned blocks_for(int64_t n, int threads) {
  return static_cast<unsigned>((n + threads - 1) / threads);
}

}  // namespace dliom

using namespace dliom;

GridView dliom_grid::view() const {
  GridView v;
  v.table = d_table;
  v.pool = d_pool;
  v.half = 32 << bits;
  v.leaves_per_axis = 8 << bits;
  v.grid_size = 64u << bits;
  v.resolution = resolution;
  v.inv_resolution = 1.f / resolution;
  v.log2_leaves = bits + 3;
  v.dense = d_dense;
  v.dense_stride = dense_stride;
  v.dense_bricks = dense_bricks;
  for (int a = 0; a < 3; ++a) v.dense_off[a] = dense_off[a];
  return v;
}

// L^3 leaf slots plus one trailing sentinel entry that is always 0: kernels send out-of-extent
// lookups there and read the null leaf without a second select.
static size_t table_entries(int bits) {
  const size_t L = static_cast<size_t>(8) << bits;
  return L * L * L + 1;
}

void dliom_grid::book() {
  if (!ledger) return;
  const int64_t table = d_table != nullptr ? static_cast<int64_t>(table_entries(bits) * sizeof(uint32_t)) : 0;
  const int64_t pool = d_pool != nullptr ? capacity * (1024 + 12) : 0;
  const int64_t mirror = mirror_bytes();
  ledger->leaf_table_bytes += table - booked_table;
  ledger->leaf_pool_bytes += pool - booked_pool;
  ledger->mirror_bytes += mirror - booked_mirror;
  booked_table = table;
  booked_pool = pool;
  booked_mirror = mirror;
}

int dliom_grid::refresh_count(int64_t* count) {
  uint32_t c = 0;
  DLIOM_HIP_TRY(hipMemcpyAsync(&c, d_count, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  used_upper = c;
  applied_seq = insert_seq;  // exact as of now: the pinned count of the last fused insertion is superseded
  if (count != nullptr) *count = c;
  return DLIOM_OK;
}

int dliom_grid::ensure_bits(int needed_bits) {
  if (needed_bits <= bits) return DLIOM_OK;
  if (needed_bits > 8) return DLIOM_ERR_GRID_EXTENT;
  if (needed_bits > kMaxFlatBits) return DLIOM_ERR_GRID_EXTENT;  // flat table limit, DESIGN.md
  uint32_t* new_table = nullptr;
  const size_t entries = table_entries(needed_bits);
  DLIOM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&new_table), entries * sizeof(uint32_t)));
  DLIOM_HIP_TRY(hipMemsetAsync(new_table, 0, entries * sizeof(uint32_t), ctx->stream));
  int64_t count = 0;
  DLIOM_TRY(refresh_count(&count));
  if (count > 1) {
    hipLaunchKernelGGL(rebuild_table_kernel, dim3(blocks_for(count - 1, 256)), dim3(256), 0,
                       ctx->stream, d_slot_coord, static_cast<uint32_t>(count), new_table,
                       4 << needed_bits, 8u << needed_bits);
    DLIOM_HIP_TRY(hipGetLastError());
  }
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  DLIOM_HIP_TRY(hipFree(d_table));
  d_table = new_table;
  bits = needed_bits;
  drop_dense();  // wrong extent now; rebuilt lazily by the next match
  book();
  return DLIOM_OK;
}

void dliom_grid::drop_dense() {
  if (d_dense != nullptr) (void)hipFree(d_dense);
  d_dense = nullptr;
  dense_stride = 0;
  dense_bricks = 0;
  dense_windowed = false;
  book();
}

// Dense mirror of the grid for the correlative matcher: (grid_size + 2)^3 uint16 in 4x4x4 bricks
// (dense_index), one guard cell per side, holding what the matcher SUMS: max(value & 0x7fff, 1) -- unknown cells,
// guard cells and everything outside read 1 (= kMinProbability's value, rtcsm3d.hip), known cells
// their marker-free value (always >= 1 once a lookup table was applied).  Spends HBM capacity (258 MiB at
// bits = 3, 2.0 GiB at bits = 4) to make a voxel lookup ONE load at a linear address instead of
// leaf-table load + leaf load; kept in sync by the insertion kernels (write-through) and rebuilt
// from the leaf pool after uploads or growth.  Grids beyond bits = 4 stay on the leaf path.
static int build_mirror(dliom_grid* g, const int off[3], int stride, bool windowed) {
  const size_t bricks = static_cast<size_t>((stride + 3) >> 2);
  const size_t cells = bricks * bricks * bricks * 64;
  // A window rebuilt around a new pose has the size of the old one (the radius is the scan's, clamped): the allocation
  // is kept -- a 4 GB hipFree synchronises the device and the hipMalloc behind it is no cheaper -- and only refilled.
  // Every failure below leaves the grid WITHOUT a mirror (d_dense null), never with a half-built one.
  const bool reuse = g->d_dense != nullptr && g->dense_stride == stride;
  if (g->d_dense != nullptr && !reuse) g->drop_dense();
  g->dense_stride = 0;  // not usable until it is filled
  g->dense_bricks = 0;
  g->dense_windowed = false;
  if (!reuse) {
    // the caller's cap on mirror memory (dliom_ctx_set_mirror_budget): refused like an allocation that failed -- the
    // matcher falls back to the leaf-table kernels on this grid
    if (g->ledger && g->ledger->mirror_budget > 0 &&
        g->ledger->mirror_bytes + static_cast<int64_t>(cells * sizeof(uint16_t)) > g->ledger->mirror_budget) {
      ++g->ledger->mirrors_refused;
      return DLIOM_ERR_GRID_EXTENT;
    }
    if (cells * sizeof(uint16_t) > (size_t{1} << 30)) {  // windows and bits = 4 mirrors: is the memory there at all?
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < cells * sizeof(uint16_t) + (size_t{1} << 30)) {
        g->d_dense = nullptr;
        return DLIOM_ERR_GRID_EXTENT;  // the caller falls back to the leaf-table kernels
      }
    }
    if (hipMalloc(reinterpret_cast<void**>(&g->d_dense), cells * sizeof(uint16_t)) != hipSuccess) {
      (void)hipGetLastError();
      g->d_dense = nullptr;
      return DLIOM_ERR_GRID_EXTENT;
    }
  }
  int64_t count = 0;
  int st = hipMemsetD16Async(reinterpret_cast<hipDeviceptr_t>(g->d_dense), 1, cells, g->ctx->stream) == hipSuccess ? DLIOM_OK : DLIOM_ERR_HIP;
  if (st == DLIOM_OK) st = g->refresh_count(&count);
  DenseOff o{{off[0], off[1], off[2]}};
  if (st == DLIOM_OK && count > 1) {
    hipLaunchKernelGGL(dense_fill_kernel, dim3(static_cast<unsigned>(count - 1)), dim3(256), 0, g->ctx->stream,
                       g->d_slot_coord, g->d_pool, g->d_dense, o, stride);
    if (hipGetLastError() != hipSuccess) st = DLIOM_ERR_HIP;
  }
  if (st != DLIOM_OK) {
    g->drop_dense();
    return st;
  }
  ++g->dense_rebuilds;
  g->dense_stride = stride;
  g->dense_bricks = static_cast<int>(bricks);
  for (int a = 0; a < 3; ++a) g->dense_off[a] = off[a];
  g->dense_windowed = windowed;
  g->book();
  return DLIOM_OK;
}

int dliom_grid::ensure_dense() {
  if (d_dense != nullptr && !dense_windowed) return DLIOM_OK;
  if (bits > kMaxDenseBits) return DLIOM_ERR_GRID_EXTENT;
  const int h1 = (32 << bits) + 1;
  const int off[3] = {h1, h1, h1};
  return build_mirror(this, off, (64 << bits) + 2, false);
}

// Grids beyond bits = 4 (a 10 cm grid with more than 51 m of extent: every outdoor submap whose high-resolution range
// is not cut at 20 m) cannot be mirrored whole -- 17.6 GB at bits = 5, and the LDS-box kernel addresses the mirror with
// 32-bit byte offsets.  The matcher only ever reads within (farthest point + search window) of its initial pose:
// a WINDOW of the grid around that pose is mirrored instead, at most 1264 cells a side (4.04 GB).  The insertion kernels
// write through to the cells that lie inside it; the window is kept while the next match still fits and rebuilt around
// the new pose, with a margin of 48 cells, when it does not (memset + one pass over the leaves: about a millisecond per
// gigabyte, once every few metres of travel).
int dliom_grid::ensure_dense_for(const int centre[3], int radius_cells) {
  if (bits <= kMaxDenseBits) return ensure_dense();
  constexpr int kMaxSide = 1264, kMargin = 48;
  if (radius_cells < 1 || 2 * (radius_cells + 2) + 4 > kMaxSide) return DLIOM_ERR_GRID_EXTENT;
  if (d_dense != nullptr && dense_windowed) {
    bool inside = true;
    for (int a = 0; a < 3; ++a) {
      const int lo = centre[a] - radius_cells + dense_off[a], hi = centre[a] + radius_cells + dense_off[a];
      inside = inside && lo >= 1 && hi <= dense_stride - 2;  // one guard cell per side, like the whole-grid mirror
    }
    if (inside) return DLIOM_OK;
  }
  const int r = std::min(radius_cells + kMargin, (kMaxSide - 8) / 2);
  const int stride = (2 * r + 2 + 3) & ~3;
  int off[3];
  for (int a = 0; a < 3; ++a) off[a] = -(centre[a] - r) + 1;  // cell centre - r -> mirror coordinate 1
  return build_mirror(this, off, stride, true);
}

int dliom_grid::ensure_capacity(int64_t additional_slots) {
  if (h_count_slot != nullptr && insert_seq != 0 && applied_seq != insert_seq &&
      __atomic_load_n(h_count_slot, __ATOMIC_ACQUIRE) == insert_seq) {
    // the last fused insertion has finished and left its exact count: tighten the bound without touching the stream
    used_upper = static_cast<int64_t>(__atomic_load_n(h_count_slot + 1, __ATOMIC_RELAXED)) + (used_upper - upper_at_insert);
    applied_seq = insert_seq;
  }
  if (used_upper + additional_slots <= capacity) return DLIOM_OK;
  int64_t count = 0;
  DLIOM_TRY(refresh_count(&count));  // tighten the pessimistic bound first
  if (count + additional_slots <= capacity) return DLIOM_OK;
  int64_t new_cap = std::max<int64_t>(capacity * 2, count + additional_slots);
  new_cap = std::max<int64_t>(new_cap, 4096);
  // the score kernel addresses the pool with 32-bit byte offsets: <= 4 Mi leaves (4 GiB)
  new_cap = std::min<int64_t>(new_cap, int64_t{1} << 22);
  if (count + additional_slots > new_cap) return DLIOM_ERR_GRID_EXTENT;
  uint16_t* new_pool = nullptr;
  int32_t* new_coord = nullptr;
  DLIOM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&new_pool), static_cast<size_t>(new_cap) * 1024));
  DLIOM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&new_coord), static_cast<size_t>(new_cap) * 12));
  // invariant: slots >= count are zero
  DLIOM_HIP_TRY(hipMemsetAsync(new_pool, 0, static_cast<size_t>(new_cap) * 1024, ctx->stream));
  if (d_pool != nullptr) {
    DLIOM_HIP_TRY(hipMemcpyAsync(new_pool, d_pool, static_cast<size_t>(count) * 1024,
                                 hipMemcpyDeviceToDevice, ctx->stream));
    DLIOM_HIP_TRY(hipMemcpyAsync(new_coord, d_slot_coord, static_cast<size_t>(count) * 12,
                                 hipMemcpyDeviceToDevice, ctx->stream));
  }
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (d_pool != nullptr) {
    DLIOM_HIP_TRY(hipFree(d_pool));
    DLIOM_HIP_TRY(hipFree(d_slot_coord));
  }
  d_pool = new_pool;
  d_slot_coord = new_coord;
  capacity = new_cap;
  book();
  return DLIOM_OK;
}

// Reallocates the leaf pool to the slots in use (rounded up to 256): a finished submap is never inserted into
// again, and insertion sizes the pool for the worst case (every return and free-space voxel in a new leaf).
int dliom_grid::shrink_to_fit() {
  int64_t count = 0;
  DLIOM_TRY(refresh_count(&count));
  const int64_t want = std::max<int64_t>(256, (count + 255) & ~int64_t{255});
  if (d_pool == nullptr || want >= capacity) return DLIOM_OK;
  uint16_t* new_pool = nullptr;
  int32_t* new_coord = nullptr;
  // every failure below leaves the grid exactly as it was (old pool, old capacity) and frees what was allocated
  if (hipMalloc(reinterpret_cast<void**>(&new_pool), static_cast<size_t>(want) * 1024) != hipSuccess) {
    (void)hipGetLastError();
    return DLIOM_ERR_HIP;
  }
  if (hipMalloc(reinterpret_cast<void**>(&new_coord), static_cast<size_t>(want) * 12) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(new_pool);
    return DLIOM_ERR_HIP;
  }
  const bool ok = hipMemsetAsync(new_pool, 0, static_cast<size_t>(want) * 1024, ctx->stream) == hipSuccess &&
                  hipMemcpyAsync(new_pool, d_pool, static_cast<size_t>(count) * 1024, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
                  hipMemcpyAsync(new_coord, d_slot_coord, static_cast<size_t>(count) * 12, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
                  hipStreamSynchronize(ctx->stream) == hipSuccess;
  if (!ok) {
    (void)hipFree(new_pool);
    (void)hipFree(new_coord);
    return DLIOM_ERR_HIP;
  }
  (void)hipFree(d_pool);
  (void)hipFree(d_slot_coord);
  d_pool = new_pool;
  d_slot_coord = new_coord;
  capacity = want;
  used_upper = count;
  book();
  return DLIOM_OK;
}

extern "C" {

int dliom_grid_create(dliom_ctx* ctx, float resolution, dliom_grid** out) {
  if (ctx == nullptr || out == nullptr || !(resolution > 0.f)) return DLIOM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  dliom_grid* g = new dliom_grid;
  g->ctx = ctx;
  g->ledger = ctx->ledger;
  g->resolution = resolution;
  g->bits = 1;
  int s = DLIOM_OK;
  do {
    if (hipMalloc(reinterpret_cast<void**>(&g->d_table), table_entries(1) * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&g->d_count), sizeof(uint32_t)) != hipSuccess) {
      s = DLIOM_ERR_HIP;
      break;
    }
    if (hipMemsetAsync(g->d_table, 0, table_entries(1) * sizeof(uint32_t), ctx->stream) != hipSuccess) {
      s = DLIOM_ERR_HIP;
      break;
    }
    const uint32_t one = 1;  // slot 0 is the reserved null leaf
    if (hipMemcpyAsync(g->d_count, &one, sizeof(one), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
      s = DLIOM_ERR_HIP;
      break;
    }
    g->used_upper = 1;
    {
      void* slot = nullptr;  // optional: without it the leaf count is read back the old way (memcpy + synchronise)
      if (hipHostMalloc(&slot, 64, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess) {
        g->h_count_slot = static_cast<unsigned*>(slot);
        g->h_count_slot[0] = g->h_count_slot[1] = 0u;
      } else {
        (void)hipGetLastError();
      }
    }
    s = g->ensure_capacity(1024);
  } while (false);
  ++g->ledger->grids;  // (dliom_grid_destroy takes it back, also on the failure path below)
  g->book();
  if (s != DLIOM_OK) {
    dliom_grid_destroy(g);
    return s;
  }
  *out = g;
  return DLIOM_OK;
}

int dliom_grid_destroy(dliom_grid* g) {
  if (g == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  // Objects may be torn down in any order (e.g. at interpreter exit): do not touch g->ctx.
  (void)hipDeviceSynchronize();
  if (g->d_table) (void)hipFree(g->d_table);
  if (g->d_pool) (void)hipFree(g->d_pool);
  if (g->d_slot_coord) (void)hipFree(g->d_slot_coord);
  if (g->d_count) (void)hipFree(g->d_count);
  if (g->d_dense) (void)hipFree(g->d_dense);
  if (g->h_count_slot) (void)hipHostFree(g->h_count_slot);
  if (g->ledger) {  // the ledger outlives whichever of (context, grid) goes first
    g->ledger->leaf_table_bytes -= g->booked_table;
    g->ledger->leaf_pool_bytes -= g->booked_pool;
    g->ledger->mirror_bytes -= g->booked_mirror;
    --g->ledger->grids;
  }
  delete g;
  return DLIOM_OK;
}

int dliom_grid_memory_stats(const dliom_grid* g, dliom_memory_stats* out) {
  if (g == nullptr || out == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  std::memset(out, 0, sizeof *out);
  out->grids = 1;
  out->leaf_table_bytes = g->booked_table;
  out->leaf_pool_bytes = g->booked_pool;
  out->mirror_bytes = g->booked_mirror;
  out->mirror_windowed = g->d_dense != nullptr && g->dense_windowed ? 1 : 0;
  out->leaf_capacity = g->capacity;
  out->leaf_slots_upper_bound = g->used_upper;
  if (g->ledger) {
    out->mirror_budget_bytes = g->ledger->mirror_budget;
    out->mirrors_refused = g->ledger->mirrors_refused;
  }
  return DLIOM_OK;
}

int dliom_grid_resolution(const dliom_grid* g, float* resolution) {
  if (g == nullptr || resolution == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  *resolution = g->resolution;
  return DLIOM_OK;
}

int dliom_grid_mirror_stats(const dliom_grid* g, int64_t* rebuilds, int64_t* bytes, int* windowed) {
  if (g == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  if (rebuilds != nullptr) *rebuilds = g->dense_rebuilds;
  if (bytes != nullptr)
    *bytes = g->d_dense != nullptr ? static_cast<int64_t>(g->dense_bricks) * g->dense_bricks * g->dense_bricks * 128 : 0;
  if (windowed != nullptr) *windowed = g->d_dense != nullptr && g->dense_windowed ? 1 : 0;
  return DLIOM_OK;
}

int dliom_grid_bits(const dliom_grid* g, int* bits) {
  if (g == nullptr || bits == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  *bits = g->bits;
  return DLIOM_OK;
}

int dliom_grid_upload_blocks(dliom_grid* g, const int32_t* origins, const uint16_t* values512,
                             int64_t n) {
  if (g == nullptr || n < 0 || (n > 0 && (origins == nullptr || values512 == nullptr)))
    return DLIOM_ERR_INVALID_ARGUMENT;
  if (n == 0) return DLIOM_OK;
  dliom_ctx* ctx = g->ctx;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  int lo = 0, hi = 0;
  for (int64_t i = 0; i < 3 * n; ++i) {
    if ((origins[i] & 7) != 0) return DLIOM_ERR_INVALID_ARGUMENT;
    lo = std::min(lo, origins[i]);
    hi = std::max(hi, origins[i] + 7);
  }
  DLIOM_TRY(g->ensure_bits(needed_bits_for_cell_range(lo, hi)));
  DLIOM_TRY(g->ensure_capacity(n));
  const size_t obytes = static_cast<size_t>(n) * 12, vbytes = static_cast<size_t>(n) * 1024;
  DLIOM_TRY(ctx->misc.reserve(obytes + 256 + vbytes));
  int32_t* d_orig = ctx->misc.as<int32_t>();
  uint16_t* d_vals = reinterpret_cast<uint16_t*>(static_cast<char*>(ctx->misc.p) +
                                                  ((obytes + 255) & ~static_cast<size_t>(255)));
  DLIOM_HIP_TRY(hipMemcpyAsync(d_orig, origins, obytes, hipMemcpyHostToDevice, ctx->stream));
  DLIOM_HIP_TRY(hipMemcpyAsync(d_vals, values512, vbytes, hipMemcpyHostToDevice, ctx->stream));
  const GridView v = g->view();
  hipLaunchKernelGGL(upload_alloc_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, d_orig,
                     n, g->d_table, g->d_slot_coord, g->d_count, v.half, v.grid_size,
                     static_cast<unsigned>(v.leaves_per_axis));
  DLIOM_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(upload_copy_kernel, dim3(static_cast<unsigned>(n)), dim3(256), 0, ctx->stream,
                     d_orig, d_vals, g->d_table, g->d_pool, v.half, v.grid_size,
                     static_cast<unsigned>(v.leaves_per_axis));
  DLIOM_HIP_TRY(hipGetLastError());
  g->used_upper += n;
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));  // host buffers may be reused by the caller
  g->drop_dense();
  return DLIOM_OK;
}

int dliom_grid_num_blocks(const dliom_grid* g, int64_t* num_blocks) {
  if (g == nullptr || num_blocks == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  int64_t c = 0;
  DLIOM_TRY(const_cast<dliom_grid*>(g)->refresh_count(&c));
  *num_blocks = c - 1;
  return DLIOM_OK;
}

int dliom_grid_download_blocks(const dliom_grid* g, int32_t* origins, uint16_t* values512,
                               int64_t capacity, int64_t* num_blocks) {
  if (g == nullptr || num_blocks == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  int64_t c = 0;
  DLIOM_TRY(const_cast<dliom_grid*>(g)->refresh_count(&c));
  const int64_t n = c - 1;
  *num_blocks = n;
  if (n == 0) return DLIOM_OK;
  if (capacity < n) return DLIOM_ERR_CAPACITY;
  if (origins == nullptr || values512 == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  dliom_ctx* ctx = g->ctx;
  DLIOM_HIP_TRY(hipMemcpyAsync(origins, g->d_slot_coord + 3, static_cast<size_t>(n) * 12,
                               hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipMemcpyAsync(values512, g->d_pool + 512, static_cast<size_t>(n) * 1024,
                               hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (int64_t i = 0; i < 3 * n; ++i) origins[i] *= 8;  // leaf coordinate -> corner voxel index
  return DLIOM_OK;
}

int dliom_grid_set_values(dliom_grid* g, const int32_t* cells, const uint16_t* values, int64_t n) {
  if (g == nullptr || n < 0 || (n > 0 && (cells == nullptr || values == nullptr)))
    return DLIOM_ERR_INVALID_ARGUMENT;
  if (n == 0) return DLIOM_OK;
  dliom_ctx* ctx = g->ctx;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  int lo = 0, hi = 0;
  for (int64_t i = 0; i < 3 * n; ++i) {
    lo = std::min(lo, cells[i]);
    hi = std::max(hi, cells[i]);
  }
  DLIOM_TRY(g->ensure_bits(needed_bits_for_cell_range(lo, hi)));
  DLIOM_TRY(g->ensure_capacity(n));
  const size_t cbytes = (static_cast<size_t>(n) * 12 + 255) & ~static_cast<size_t>(255);
  DLIOM_TRY(ctx->misc.reserve(cbytes + static_cast<size_t>(n) * 2));
  int32_t* d_cells = ctx->misc.as<int32_t>();
  uint16_t* d_vals = reinterpret_cast<uint16_t*>(static_cast<char*>(ctx->misc.p) + cbytes);
  DLIOM_HIP_TRY(hipMemcpyAsync(d_cells, cells, static_cast<size_t>(n) * 12, hipMemcpyHostToDevice, ctx->stream));
  DLIOM_HIP_TRY(hipMemcpyAsync(d_vals, values, static_cast<size_t>(n) * 2, hipMemcpyHostToDevice, ctx->stream));
  const GridView v = g->view();
  hipLaunchKernelGGL(set_alloc_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, d_cells, n,
                     g->d_table, g->d_slot_coord, g->d_count, v.half, v.grid_size,
                     static_cast<unsigned>(v.leaves_per_axis));
  hipLaunchKernelGGL(set_values_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, d_cells, d_vals,
                     n, g->d_table, g->d_pool, v.half, v.grid_size, static_cast<unsigned>(v.leaves_per_axis));
  DLIOM_HIP_TRY(hipGetLastError());
  g->used_upper += n;
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  g->drop_dense();
  return DLIOM_OK;
}

int dliom_grid_get_values(const dliom_grid* g, const int32_t* cells, int64_t n, uint16_t* values) {
  if (g == nullptr || n < 0 || (n > 0 && (cells == nullptr || values == nullptr)))
    return DLIOM_ERR_INVALID_ARGUMENT;
  if (n == 0) return DLIOM_OK;
  dliom_ctx* ctx = g->ctx;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  const size_t cbytes = static_cast<size_t>(n) * 12;
  const size_t off = (cbytes + 255) & ~static_cast<size_t>(255);
  DLIOM_TRY(ctx->misc.reserve(off + static_cast<size_t>(n) * 2));
  int32_t* d_cells = ctx->misc.as<int32_t>();
  uint16_t* d_out = reinterpret_cast<uint16_t*>(static_cast<char*>(ctx->misc.p) + off);
  DLIOM_HIP_TRY(hipMemcpyAsync(d_cells, cells, cbytes, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(get_values_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, g->view(),
                     d_cells, n, d_out);
  DLIOM_HIP_TRY(hipGetLastError());
  DLIOM_HIP_TRY(hipMemcpyAsync(values, d_out, static_cast<size_t>(n) * 2, hipMemcpyDeviceToHost,
                               ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return DLIOM_OK;
}

// Shared tail of the insertion entry points.  d_returns: packed xyz on the device (n upper
// bound, n_dev optional exact count), d_hit/d_miss: 32768-entry tables on the device,
// d_scan: 2 ints of scratch.
static int insert_device(dliom_grid* g, const float origin[3], const float* d_returns, int64_t n,
                         const unsigned* n_dev, const uint16_t* d_hit, const uint16_t* d_miss,
                         int* d_scan, int num_free_space_voxels) {
  dliom_ctx* ctx = g->ctx;
  DLIOM_HIP_TRY(hipMemsetAsync(d_scan, 0, 8, ctx->stream));
  InsertArgs a;
  a.returns = d_returns;
  a.n = n;
  a.n_dev = n_dev;
  a.ox = origin[0];
  a.oy = origin[1];
  a.oz = origin[2];
  a.resolution = g->resolution;
  a.num_free = num_free_space_voxels;
  a.half = 0;
  a.gsize = 0;
  a.L = 0;
  a.dense = nullptr;
  a.dense_stride = 0;
  const dim3 grid(blocks_for(n, 256)), block(256);
  hipLaunchKernelGGL(insert_scan_kernel, grid, block, 0, ctx->stream, a, d_scan);
  DLIOM_HIP_TRY(hipGetLastError());
  int scan[2] = {0, 0};
  DLIOM_HIP_TRY(hipMemcpyAsync(scan, d_scan, 8, hipMemcpyDeviceToHost, ctx->stream));
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (scan[1] != 0) return DLIOM_ERR_RAY_TOO_LONG;
  if (scan[0] > 8) return DLIOM_ERR_GRID_EXTENT;
  DLIOM_TRY(g->ensure_bits(scan[0]));
  DLIOM_TRY(g->ensure_capacity(n * (1 + static_cast<int64_t>(num_free_space_voxels))));

  const GridView v = g->view();
  a.half = v.half;
  a.gsize = v.grid_size;
  a.L = static_cast<unsigned>(v.leaves_per_axis);
  a.dense = g->d_dense;  // write-through when the mirror exists
  a.dense_stride = g->dense_stride;
  for (int k = 0; k < 3; ++k) a.dense_off[k] = g->dense_off[k];
  uint32_t* pool32 = reinterpret_cast<uint32_t*>(g->d_pool);
  hipLaunchKernelGGL(insert_alloc_kernel, grid, block, 0, ctx->stream, a, g->d_table, g->d_slot_coord,
                     g->d_count);
  hipLaunchKernelGGL(insert_apply_kernel<0>, grid, block, 0, ctx->stream, a, g->d_table, pool32, d_hit);
  if (num_free_space_voxels > 0)
    hipLaunchKernelGGL(insert_apply_kernel<1>, grid, block, 0, ctx->stream, a, g->d_table, pool32,
                       d_miss);
  hipLaunchKernelGGL(insert_apply_kernel<2>, grid, block, 0, ctx->stream, a, g->d_table, pool32, d_hit);
  DLIOM_HIP_TRY(hipGetLastError());
  g->used_upper += n * (1 + static_cast<int64_t>(num_free_space_voxels));
  return DLIOM_OK;
}

int dliom_grid_insert(dliom_grid* g, const float origin[3], const float* returns_xyz, int64_t n,
                      const uint16_t* hit_table, const uint16_t* miss_table,
                      int num_free_space_voxels) {
  if (g == nullptr || origin == nullptr || n < 0 || hit_table == nullptr || miss_table == nullptr ||
      (n > 0 && returns_xyz == nullptr) || num_free_space_voxels < 0)
    return DLIOM_ERR_INVALID_ARGUMENT;
  if (n == 0) return DLIOM_OK;  // nothing to insert; FinishUpdate on nothing
  dliom_ctx* ctx = g->ctx;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  // scratch: [returns | hit lut | miss lut | scan out]
  const size_t pbytes = (static_cast<size_t>(n) * 12 + 255) & ~static_cast<size_t>(255);
  DLIOM_TRY(ctx->misc.reserve(pbytes + 65536 * 2 + 256));
  char* base = static_cast<char*>(ctx->misc.p);
  float* d_returns = reinterpret_cast<float*>(base);
  uint16_t* d_hit = reinterpret_cast<uint16_t*>(base + pbytes);
  uint16_t* d_miss = d_hit + 32768;
  int* d_scan = reinterpret_cast<int*>(base + pbytes + 65536 * 2);
  DLIOM_HIP_TRY(hipMemcpyAsync(d_returns, returns_xyz, static_cast<size_t>(n) * 12,
                               hipMemcpyHostToDevice, ctx->stream));
  DLIOM_HIP_TRY(hipMemcpyAsync(d_hit, hit_table, 65536, hipMemcpyHostToDevice, ctx->stream));
  DLIOM_HIP_TRY(hipMemcpyAsync(d_miss, miss_table, 65536, hipMemcpyHostToDevice, ctx->stream));
  const int span = ctx->begin_span(DLIOM_KERNEL_INSERT);
  const int s = insert_device(g, origin, d_returns, n, nullptr, d_hit, d_miss, d_scan,
                              num_free_space_voxels);
  ctx->end_span(span);
  // ctx->misc is reused by later calls on this context: finish before returning.
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return s;
}

int dliom_inserter_create(dliom_ctx* ctx, double hit_probability, double miss_probability,
                          int num_free_space_voxels, dliom_inserter** out) {
  if (ctx == nullptr || out == nullptr || num_free_space_voxels < 0) return DLIOM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  // CHECK_GT(hit, 0.5), CHECK_LT(miss, 0.5): range_data_inserter_3d.cc:64-65
  if (!(hit_probability > 0.5) || !(miss_probability < 0.5)) return DLIOM_ERR_INVALID_ARGUMENT;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  dliom_inserter* ins = new dliom_inserter;
  ins->ctx = ctx;
  ins->num_free_space_voxels = num_free_space_voxels;
  ins->hit_table.resize(32768);
  ins->miss_table.resize(32768);
  // RangeDataInserter3D ctor: ComputeLookupTableToApplyOdds(Odds(float(p)))
  dliom_compute_lookup_table_to_apply_odds(dliom_odds(static_cast<float>(hit_probability)),
                                           ins->hit_table.data());
  dliom_compute_lookup_table_to_apply_odds(dliom_odds(static_cast<float>(miss_probability)),
                                           ins->miss_table.data());
  if (hipMalloc(reinterpret_cast<void**>(&ins->d_tables), 2 * 65536 + 256) != hipSuccess) {
    delete ins;
    return DLIOM_ERR_HIP;
  }
  if (hipMemcpy(ins->d_tables, ins->hit_table.data(), 65536, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(ins->d_tables + 32768, ins->miss_table.data(), 65536, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(ins->d_tables);
    delete ins;
    return DLIOM_ERR_HIP;
  }
  *out = ins;
  return DLIOM_OK;
}

int dliom_inserter_destroy(dliom_inserter* ins) {
  if (ins == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  if (ins->d_tables != nullptr) (void)hipFree(ins->d_tables);
  delete ins;
  return DLIOM_OK;
}

int dliom_inserter_tables(const dliom_inserter* ins, uint16_t* hit_table, uint16_t* miss_table) {
  if (ins == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  if (hit_table != nullptr) std::memcpy(hit_table, ins->hit_table.data(), 65536);
  if (miss_table != nullptr) std::memcpy(miss_table, ins->miss_table.data(), 65536);
  return DLIOM_OK;
}

int dliom_inserter_insert(const dliom_inserter* ins, dliom_grid* g, const float origin[3],
                          const float* returns_xyz, int64_t n) {
  if (ins == nullptr || g == nullptr || origin == nullptr || n < 0 || (n > 0 && returns_xyz == nullptr))
    return DLIOM_ERR_INVALID_ARGUMENT;
  if (n == 0) return DLIOM_OK;
  dliom_ctx* ctx = g->ctx;
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  const size_t pbytes = (static_cast<size_t>(n) * 12 + 255) & ~static_cast<size_t>(255);
  DLIOM_TRY(ctx->misc.reserve(pbytes + 256));
  char* base = static_cast<char*>(ctx->misc.p);
  float* d_returns = reinterpret_cast<float*>(base);
  int* d_scan = reinterpret_cast<int*>(base + pbytes);
  DLIOM_HIP_TRY(hipMemcpyAsync(d_returns, returns_xyz, static_cast<size_t>(n) * 12,
                               hipMemcpyHostToDevice, ctx->stream));
  const int span = ctx->begin_span(DLIOM_KERNEL_INSERT);
  const int s = insert_device(g, origin, d_returns, n, nullptr, ins->d_tables, ins->d_tables + 32768,
                              d_scan, ins->num_free_space_voxels);
  ctx->end_span(span);
  DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return s;
}

// Applies a pose chain to the origin on the host with the device's float operation order.
static void transform_origin(const float* poses7, int num_poses, const float origin[3], float out[3]) {
  float o[3] = {origin[0], origin[1], origin[2]};
  for (int k = 0; k < num_poses; ++k) {
    const float* p = poses7 + 7 * k;
    const float qw = p[3], qx = p[4], qy = p[5], qz = p[6];
    float uvx = qy * o[2] - qz * o[1], uvy = qz * o[0] - qx * o[2], uvz = qx * o[1] - qy * o[0];
    uvx = uvx + uvx;
    uvy = uvy + uvy;
    uvz = uvz + uvz;
    const float cx = qy * uvz - qz * uvy, cy = qz * uvx - qx * uvz, cz = qx * uvy - qy * uvx;
    const float nx = ((o[0] + qw * uvx) + cx) + p[0];
    const float ny = ((o[1] + qw * uvy) + cy) + p[1];
    const float nz = ((o[2] + qw * uvz) + cz) + p[2];
    o[0] = nx;
    o[1] = ny;
    o[2] = nz;
  }
  out[0] = o[0];
  out[1] = o[1];
  out[2] = o[2];
}

// Host-side proof that pass 0 (the extent scan) of an insertion would report nothing for this target -- no cell beyond
// the grid's current extent, no ray of 2^15 cells.  Every hit is the image of a cloud point under the target's rigid
// chain: it lies in the image of the cloud's bounding box [-e, e] (e = max |x|, |y|, |z| where the cloud's producer
// recorded them, max ||p|| per axis otherwise) -- centre c <- R c + t, half extents e <- |R| e per pose -- and, when the
// range filter is on, within max_range of the chain's image of the origin; every miss cell lies in the box spanned by
// the origin's cell and a hit's cell (range_data_inserter_3d.cc:36-50).  Float rounding of the chains and quaternions a
// few 1e-7 off unit norm are covered by the relative slack; anything else (non-finite input, quaternions far from unit)
// is "not proven" and the caller runs the scan as before.
static bool insertion_provably_inside(const InsertTarget& tg, const float origin_local[3], const dliom_cloud& cloud) {
  if (!(cloud.max_norm >= 0.f) || !std::isfinite(cloud.max_norm)) return false;
  double c[3] = {0.0, 0.0, 0.0}, e[3];
  for (int a = 0; a < 3; ++a) {
    e[a] = cloud.abs_max[a] >= 0.f ? std::min<double>(cloud.abs_max[a], cloud.max_norm) : static_cast<double>(cloud.max_norm);
    if (!std::isfinite(e[a])) return false;
  }
  for (int j = 0; j < tg.num_poses; ++j) {
    const Quat4& q = tg.q[j];
    const double w = q.w, x = q.x, y = q.y, z = q.z;
    const double n2 = w * w + x * x + y * y + z * z;
    if (!(n2 > 0.999 && n2 < 1.001)) return false;
    // v + 2 w (u x v) + 2 u x (u x v) as a matrix (what rotate_point computes, for any q)
    const double M[3][3] = {{1.0 - 2.0 * (y * y + z * z), 2.0 * (x * y - w * z), 2.0 * (x * z + w * y)},
                            {2.0 * (x * y + w * z), 1.0 - 2.0 * (x * x + z * z), 2.0 * (y * z - w * x)},
                            {2.0 * (x * z - w * y), 2.0 * (y * z + w * x), 1.0 - 2.0 * (x * x + y * y)}};
    double nc[3], ne[3];
    for (int a = 0; a < 3; ++a) {
      nc[a] = M[a][0] * c[0] + M[a][1] * c[1] + M[a][2] * c[2] + tg.t[j][a];
      ne[a] = std::fabs(M[a][0]) * e[0] + std::fabs(M[a][1]) * e[1] + std::fabs(M[a][2]) * e[2];
    }
    for (int a = 0; a < 3; ++a) {
      c[a] = nc[a];
      e[a] = ne[a];
    }
  }
  const double on = std::sqrt(double(origin_local[0]) * origin_local[0] + double(origin_local[1]) * origin_local[1] +
                              double(origin_local[2]) * origin_local[2]);
  double reach = (static_cast<double>(cloud.max_norm) + on) * 1.001;  // |hit - origin| in the world frame
  if (tg.max_range > 0.f) reach = std::min(reach, static_cast<double>(tg.max_range));
  reach = reach * (1.0 + 1e-4) + 1e-3;
  const double res = tg.resolution;
  if (!std::isfinite(reach) || !(res > 0.0)) return false;
  if (reach / res + 4.0 >= 32768.0) return false;  // num_samples < 1 << 15
  const double half = static_cast<double>(tg.half);
  const float o[3] = {tg.ox, tg.oy, tg.oz};
  for (int a = 0; a < 3; ++a) {
    if (!std::isfinite(o[a]) || !std::isfinite(c[a]) || !std::isfinite(e[a])) return false;
    const double ao = std::fabs(static_cast<double>(o[a]));
    const double by_box = std::fabs(c[a]) + e[a] * (1.0 + 1e-4) + 1e-3;  // |hit coordinate|, from the cloud's box
    const double by_range = ao + reach;                                   // ... from the distance to the origin
    const double far = std::max(ao, std::min(by_box, by_range));          // hits and the origin itself
    if (far / res + 3.0 >= half) return false;  // cells in [-half, half)
  }
  return true;
}

static void refresh_target(InsertTarget* tg, dliom_grid* g) {
  const GridView v = g->view();
  tg->bits = g->bits;
  tg->half = v.half;
  tg->gsize = v.grid_size;
  tg->L = static_cast<unsigned>(v.leaves_per_axis);
  tg->table = g->d_table;
  tg->slot_coord = g->d_slot_coord;
  tg->count = g->d_count;
  tg->pool32 = reinterpret_cast<uint32_t*>(g->d_pool);
  tg->dense = g->d_dense;
  tg->dense_stride = g->dense_stride;
  for (int k = 0; k < 3; ++k) tg->dense_off[k] = g->dense_off[k];
  tg->count_slot = g->h_count_slot;
}

// Accounts for an insertion of n returns into the grids on the host: the pessimistic bound now, the exact count when the
// last pass has written it (dliom_grid::ensure_capacity).  Call BEFORE the launches (the sequence number rides in them).
static void account_insertion_one(MultiInsertArgs* a, dliom_grid* g, int k, int64_t n, int F) {
  g->used_upper += n * (1 + static_cast<int64_t>(F));
  g->insert_seq = g->insert_seq + 1u == 0u ? 1u : g->insert_seq + 1u;
  g->upper_at_insert = g->used_upper;
  a->tg[k].count_seq = g->insert_seq;
}
static void account_insertion(MultiInsertArgs* a, dliom_grid* const* grids, int num_targets, int64_t n, int F) {
  for (int k = 0; k < num_targets; ++k) account_insertion_one(a, grids[k], k, n, F);
}

int dliom_inserter_insert_cloud_multi(const dliom_inserter* ins, int num_targets, dliom_grid* const* grids,
                                      const float* poses7, const int* num_poses, const float origin[3],
                                      const dliom_cloud* cloud, const float* max_range) {
  if (ins == nullptr || grids == nullptr || origin == nullptr || cloud == nullptr || num_poses == nullptr ||
      max_range == nullptr || num_targets < 1 || num_targets > kMaxInsertTargets)
    return DLIOM_ERR_INVALID_ARGUMENT;
  const int64_t n = cloud->n;
  if (n == 0) return DLIOM_OK;
  if (n > (int64_t{1} << 30)) return DLIOM_ERR_INVALID_ARGUMENT;
  dliom_ctx* ctx = grids[0]->ctx;
#ifdef DLIOM_EXPERIMENTS
  static const int timing = tuning_int("DLIOM_TIMING", 0);
  const auto t_begin = std::chrono::steady_clock::now();
  struct TimingGuard {
    std::chrono::steady_clock::time_point t0;
    int on;
    const char* path = "scan";
    ~TimingGuard() {
      if (on) std::fprintf(stderr, "TIMING insert (%s): %.1f us on the host\n", path,
                           std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
  } timing_guard{t_begin, timing};
#endif
  DLIOM_HIP_TRY(hipSetDevice(ctx->device));
  MultiInsertArgs a;
  std::memset(&a, 0, sizeof(a));
  const int F = ins->num_free_space_voxels;
  for (int k = 0; k < num_targets; ++k) {
    if (grids[k] == nullptr || num_poses[k] < 0 || num_poses[k] > 2 || (num_poses[k] > 0 && poses7 == nullptr))
      return DLIOM_ERR_INVALID_ARGUMENT;
    InsertTarget& tg = a.tg[k];
    const float* p = poses7 + 14 * k;  // two pose slots per target
    tg.num_poses = num_poses[k];
    for (int j = 0; j < num_poses[k]; ++j) {
      tg.q[j] = Quat4{p[7 * j + 3], p[7 * j + 4], p[7 * j + 5], p[7 * j + 6]};
      tg.t[j][0] = p[7 * j];
      tg.t[j][1] = p[7 * j + 1];
      tg.t[j][2] = p[7 * j + 2];
    }
    float o[3];
    transform_origin(p, num_poses[k], origin, o);
    tg.ox = o[0];
    tg.oy = o[1];
    tg.oz = o[2];
    tg.max_range = max_range[k];
    tg.resolution = grids[k]->resolution;
    tg.num_free = F;
    DLIOM_TRY(grids[k]->ensure_capacity(n * (1 + static_cast<int64_t>(F))));
    refresh_target(&tg, grids[k]);
  }
  a.px = cloud->d_x;
  a.py = cloud->d_y;
  a.pz = cloud->d_z;
  a.n = static_cast<int>(n);
  a.hit = ins->d_tables;
  a.miss = ins->d_tables + 32768;
  a.run_mask = (1u << num_targets) - 1u;
  const dim3 grid_dim(blocks_for(n, 256), num_targets), block(256);
  // Steady state (round 5): the scan's range is known on the host (max ||p|| travels with the cloud), so "no target
  // needs more bits, no ray is too long" is usually PROVEN before anything is launched -- then there is no extent scan,
  // no status fill and, above all, no verdict to wait for: the four update passes are enqueued and the call returns
  // (the caller's next step is ordered behind them by the stream).  The wait was a polled round trip in the middle of
  // the chain -- fill, pass 0, [host], passes 1-4 -- that kept host and device idle in turn.
  bool proven = true;
  for (int k = 0; k < num_targets && proven; ++k) proven = insertion_provably_inside(a.tg[k], origin, *cloud);
  if (proven) {
    unsigned* zeros = nullptr;
    DLIOM_TRY(zero_words(ctx, &zeros));
    a.status = reinterpret_cast<int*>(zeros);  // read only: "nothing to report" for every target
    a.host_status = nullptr;
    a.done_word = nullptr;
    account_insertion(&a, grids, num_targets, n, F);
    const int span = ctx->begin_span(DLIOM_KERNEL_INSERT);
    hipLaunchKernelGGL(multi_insert_kernel<1>, grid_dim, block, 0, ctx->stream, a);
    hipLaunchKernelGGL(multi_insert_kernel<2>, grid_dim, block, 0, ctx->stream, a);
    if (F > 0) hipLaunchKernelGGL(multi_insert_kernel<3>, grid_dim, block, 0, ctx->stream, a);
    hipLaunchKernelGGL(multi_insert_kernel<4>, grid_dim, block, 0, ctx->stream, a);
    ctx->end_span(span);
    DLIOM_HIP_TRY(hipGetLastError());
#ifdef DLIOM_EXPERIMENTS
    timing_guard.path = "proven";
#endif
    return DLIOM_OK;
  }
  DLIOM_TRY(ctx->misc.reserve(256));
  a.status = ctx->misc.as<int>();
  DLIOM_HIP_TRY(hipMemsetAsync(a.status, 0, 8 * kMaxInsertTargets, ctx->stream));
  a.host_status = static_cast<int*>(ctx->pinned);  // device-visible; written by pass 1 only
  // The host needs pass 0's verdict, not the end of the update passes: everything that follows on this context is
  // ordered behind them by the stream.  Pass 1 ends its copy of the verdict with a completion word; the call returns
  // when that arrives (the update passes may still be running -- 50 us the caller's next step no longer waits for).
  a.done_word = ctx->done_word;
  a.done_seq = ctx->done_word != nullptr ? (++ctx->done_seq == 0u ? ++ctx->done_seq : ctx->done_seq) : 0u;
  account_insertion(&a, grids, num_targets, n, F);
  const int span = ctx->begin_span(DLIOM_KERNEL_INSERT);
  hipLaunchKernelGGL(multi_insert_kernel<0>, grid_dim, block, 0, ctx->stream, a);
  int status[2 * kMaxInsertTargets];
  for (int attempt = 0; attempt < 2; ++attempt) {
    hipLaunchKernelGGL(multi_insert_kernel<1>, grid_dim, block, 0, ctx->stream, a);
    hipLaunchKernelGGL(multi_insert_kernel<2>, grid_dim, block, 0, ctx->stream, a);
    if (F > 0) hipLaunchKernelGGL(multi_insert_kernel<3>, grid_dim, block, 0, ctx->stream, a);
    hipLaunchKernelGGL(multi_insert_kernel<4>, grid_dim, block, 0, ctx->stream, a);
    DLIOM_HIP_TRY(hipGetLastError());
    if (attempt == 0) {
      if (a.done_word != nullptr)
        DLIOM_TRY(wait_done(ctx, ctx->stream, a.done_word, a.done_seq));
      else
        DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
      std::memcpy(status, a.host_status, sizeof(status));
      a.host_status = nullptr;
      a.done_word = nullptr;
      unsigned redo = 0;
      for (int k = 0; k < num_targets; ++k) {
        if (status[2 * k + 1] != 0) {
          ctx->end_span(span);
          return DLIOM_ERR_RAY_TOO_LONG;  // NB: targets before k may already be updated, like a CHECK mid-way
        }
        if (status[2 * k] > 8) {
          ctx->end_span(span);
          return DLIOM_ERR_GRID_EXTENT;
        }
        if (status[2 * k] > grids[k]->bits) {  // skipped on the device: grow, then run it alone
          DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));  // (the other targets' passes are done before anything is reallocated)
          DLIOM_TRY(grids[k]->ensure_bits(status[2 * k]));
          // ensure_bits() read the exact count back (refresh_count: used_upper = the count BEFORE this target's passes,
          // applied_seq = insert_seq), which drops the pessimistic share account_insertion() added for this insertion --
          // and the redo passes are about to allocate up to n (1 + F) leaves.  Account for them again, under a new
          // sequence number, so that used_upper stays an upper bound and pass 4's exact count of the REDO is the one
          // ensure_capacity() applies later (the invariant was broken until the next insertion).
          DLIOM_TRY(grids[k]->ensure_capacity(n * (1 + static_cast<int64_t>(F))));
          refresh_target(&a.tg[k], grids[k]);
          account_insertion_one(&a, grids[k], k, n, F);
          redo |= 1u << k;
        }
      }
      if (redo == 0) break;
      a.run_mask = redo;
    } else {
      DLIOM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
  }
  ctx->end_span(span);
  return DLIOM_OK;
}

int dliom_inserter_insert_cloud(const dliom_inserter* ins, dliom_grid* g, const float* poses7,
                                int num_poses, const float origin[3], const dliom_cloud* cloud,
                                float max_range) {
  if (g == nullptr || num_poses < 0 || num_poses > 2) return DLIOM_ERR_INVALID_ARGUMENT;
  float slots[14] = {0};
  if (num_poses > 0 && poses7 == nullptr) return DLIOM_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < 7 * num_poses; ++i) slots[i] = poses7[i];
  return dliom_inserter_insert_cloud_multi(ins, 1, &g, slots, &num_poses, origin, cloud, &max_range);
}

}  // extern "C"
