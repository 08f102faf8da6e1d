// ORACLE (test infrastructure, NOT product code).
//
// The reference's sparse occupancy voxel grid, restated as one concrete class
// (the reference composes it from three templates:
//  HybridGrid = DynamicGrid<NestedGrid<FlatGrid<uint16,3>,3>>,
//  mapping/3d/hybrid_grid.h:411-412).
//
//   leaf  : 8x8x8 uint16, z-major flat index ((z<<3)+y<<3)+x   hybrid_grid.h:40-43,66-138
//   mid   : 8x8x8 lazily allocated leaves (64^3 voxels)        hybrid_grid.h:143-244
//   top   : (2^bits)^3 lazily allocated mids, bits starts at 1, index shifted
//           by grid_size()/2, Grow() doubles the extent         hybrid_grid.h:250-409
//   cell index = lround(p / resolution) per axis                hybrid_grid.h:430-435
//   ApplyLookupTable / FinishUpdate (marker bit 1<<15)          hybrid_grid.h:494-520
//   iteration order (used by export/ToProto)                    hybrid_grid.h:93-127,187-231,303-371
#ifndef ORACLE_OM_HYBRID_GRID_H_
#define ORACLE_OM_HYBRID_GRID_H_

#include <array>
#include <cstdlib>
#include <memory>
#include <utility>
#include <vector>

#include "om_probability_values.h"

namespace oracle {

class HybridGrid {
 public:
  static constexpr int kLeafBits = 3;
  static constexpr int kMidBits = 3;
  static constexpr int kLeafSize = 1 << kLeafBits;           // 8 voxels
  static constexpr int kMidSize = kLeafSize << kMidBits;     // 64 voxels
  static constexpr int kMaxBits = 8;                         // hybrid_grid.h:389

  struct Leaf {
    std::array<uint16, 512> cells;
    Leaf() { cells.fill(0); }
  };
  struct Mid {
    std::array<std::unique_ptr<Leaf>, 512> leaves;
  };

  explicit HybridGrid(float resolution)
      : resolution_(resolution), bits_(1), top_(8) {}

  float resolution() const { return resolution_; }
  int bits() const { return bits_; }
  int grid_size() const { return kMidSize << bits_; }

  // hybrid_grid.h:430-435 -- true float division, then lround.
  Vec3i GetCellIndex(const Vec3f& point) const {
    return Vec3i(RoundToInt(point.x / resolution_),
                 RoundToInt(point.y / resolution_),
                 RoundToInt(point.z / resolution_));
  }
  // hybrid_grid.h:446-448
  Vec3f GetCenterOfCell(const Vec3i& index) const {
    return Vec3f(static_cast<float>(index.x) * resolution_,
                 static_cast<float>(index.y) * resolution_,
                 static_cast<float>(index.z) * resolution_);
  }

  // hybrid_grid.h:263-281 (+153-163, 88-90)
  uint16 value(const Vec3i& index) const {
    const int half = grid_size() >> 1;
    const int sx = index.x + half, sy = index.y + half, sz = index.z + half;
    const unsigned g = static_cast<unsigned>(grid_size());
    if (static_cast<unsigned>(sx) >= g || static_cast<unsigned>(sy) >= g ||
        static_cast<unsigned>(sz) >= g) {
      return 0;
    }
    const Mid* mid =
        top_[Flat(sx / kMidSize, sy / kMidSize, sz / kMidSize, bits_)].get();
    if (mid == nullptr) return 0;
    const int mx = sx % kMidSize, my = sy % kMidSize, mz = sz % kMidSize;
    const Leaf* leaf =
        mid->leaves[Flat(mx / kLeafSize, my / kLeafSize, mz / kLeafSize, kMidBits)]
            .get();
    if (leaf == nullptr) return 0;
    return leaf->cells[Flat(mx % kLeafSize, my % kLeafSize, mz % kLeafSize,
                            kLeafBits)];
  }

  // hybrid_grid.h:285-301 (+167-177): grows / allocates on demand.
  uint16* mutable_value(const Vec3i& index) {
    for (;;) {
      const int half = grid_size() >> 1;
      const int sx = index.x + half, sy = index.y + half, sz = index.z + half;
      const unsigned g = static_cast<unsigned>(grid_size());
      if (static_cast<unsigned>(sx) >= g || static_cast<unsigned>(sy) >= g ||
          static_cast<unsigned>(sz) >= g) {
        Grow();
        continue;
      }
      std::unique_ptr<Mid>& mid =
          top_[Flat(sx / kMidSize, sy / kMidSize, sz / kMidSize, bits_)];
      if (mid == nullptr) mid.reset(new Mid);
      const int mx = sx % kMidSize, my = sy % kMidSize, mz = sz % kMidSize;
      std::unique_ptr<Leaf>& leaf = mid->leaves[Flat(
          mx / kLeafSize, my / kLeafSize, mz / kLeafSize, kMidBits)];
      if (leaf == nullptr) leaf.reset(new Leaf);
      return &leaf->cells[Flat(mx % kLeafSize, my % kLeafSize, mz % kLeafSize,
                               kLeafBits)];
    }
  }

  // hybrid_grid.h:489-491
  void SetProbability(const Vec3i& index, float probability) {
    *mutable_value(index) = ProbabilityToValue(probability);
  }
  // hybrid_grid.h:522-524
  float GetProbability(const Vec3i& index) const {
    return ValueToProbability(value(index));
  }
  // hybrid_grid.h:527
  bool IsKnown(const Vec3i& index) const { return value(index) != 0; }

  // hybrid_grid.h:509-520
  bool ApplyLookupTable(const Vec3i& index, const std::vector<uint16>& table) {
    uint16* const cell = mutable_value(index);
    if (*cell >= kUpdateMarker) return false;
    update_indices_.push_back(cell);
    *cell = table[*cell];
    return true;
  }
  // hybrid_grid.h:494-500
  void FinishUpdate() {
    while (!update_indices_.empty()) {
      *update_indices_.back() -= kUpdateMarker;
      update_indices_.pop_back();
    }
  }

  // Visits every non-zero cell in the reference iterator's order: top cells
  // in flat z-major order, then leaves in flat order, then cells in flat order.
  template <typename F>
  void ForEachCell(F&& f) const {
    const int n_top = 1 << bits_;
    for (int t = 0; t < n_top * n_top * n_top; ++t) {
      const Mid* mid = top_[t].get();
      if (mid == nullptr) continue;
      const Vec3i tb = Unflat(t, bits_);
      for (int l = 0; l < 512; ++l) {
        const Leaf* leaf = mid->leaves[l].get();
        if (leaf == nullptr) continue;
        const Vec3i lb = Unflat(l, kMidBits);
        for (int c = 0; c < 512; ++c) {
          const uint16 v = leaf->cells[c];
          if (v == 0) continue;
          const Vec3i cb = Unflat(c, kLeafBits);
          const int off = (1 << (bits_ - 1)) * kMidSize;  // hybrid_grid.h:337
          f(Vec3i(tb.x * kMidSize + lb.x * kLeafSize + cb.x - off,
                  tb.y * kMidSize + lb.y * kLeafSize + cb.y - off,
                  tb.z * kMidSize + lb.z * kLeafSize + cb.z - off),
            v);
        }
      }
    }
  }

  // Visits every allocated leaf: (leaf-origin voxel index, 512 values).
  template <typename F>
  void ForEachLeaf(F&& f) const {
    const int n_top = 1 << bits_;
    const int off = (1 << (bits_ - 1)) * kMidSize;
    for (int t = 0; t < n_top * n_top * n_top; ++t) {
      const Mid* mid = top_[t].get();
      if (mid == nullptr) continue;
      const Vec3i tb = Unflat(t, bits_);
      for (int l = 0; l < 512; ++l) {
        const Leaf* leaf = mid->leaves[l].get();
        if (leaf == nullptr) continue;
        const Vec3i lb = Unflat(l, kMidBits);
        f(Vec3i(tb.x * kMidSize + lb.x * kLeafSize - off,
                tb.y * kMidSize + lb.y * kLeafSize - off,
                tb.z * kMidSize + lb.z * kLeafSize - off),
          leaf->cells.data());
      }
    }
  }

 private:
  // hybrid_grid.h:40-52
  static int Flat(int x, int y, int z, int bits) {
    return (((z << bits) + y) << bits) + x;
  }
  static Vec3i Unflat(int index, int bits) {
    const int mask = (1 << bits) - 1;
    return Vec3i(index & mask, (index >> bits) & mask, (index >> bits) >> bits);
  }

  // hybrid_grid.h:387-405: re-centre every mid by +2^(bits-1).
  void Grow() {
    const int new_bits = bits_ + 1;
    if (new_bits > kMaxBits) std::abort();  // CHECK_LE(new_bits, 8)
    std::vector<std::unique_ptr<Mid>> grown(8 * top_.size());
    const int n = 1 << bits_;
    const int shift = 1 << (bits_ - 1);
    for (int z = 0; z != n; ++z)
      for (int y = 0; y != n; ++y)
        for (int x = 0; x != n; ++x)
          grown[Flat(x + shift, y + shift, z + shift, new_bits)] =
              std::move(top_[Flat(x, y, z, bits_)]);
    top_ = std::move(grown);
    bits_ = new_bits;
  }

  const float resolution_;
  int bits_;
  std::vector<std::unique_ptr<Mid>> top_;
  std::vector<uint16*> update_indices_;
};

}  // namespace oracle

#endif  // ORACLE_OM_HYBRID_GRID_H_
