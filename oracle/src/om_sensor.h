// ORACLE (test infrastructure, NOT product code).
//
// Point-cloud helpers and the range-data inserter of the reference.
//   sensor/point_cloud.h:32, point_cloud.cc:25-33       PointCloud, TransformPointCloud
//   sensor/range_data.h:32-36, range_data.cc:25-33      RangeData, TransformRangeData
//   sensor/internal/voxel_filter.cc:28-37               FilterByMaxRange
//   sensor/internal/voxel_filter.cc:39-77               AdaptivelyVoxelFiltered
//   sensor/internal/voxel_filter.cc:81-90,119-131       VoxelFilter::Filter / GetCellIndex
//   mapping/3d/range_data_inserter_3d.cc:27-52,78-92    InsertMissesIntoGrid, Insert
//   mapping/3d/submap_3d.cc:42-51                       FilterRangeDataByMaxRange
#ifndef ORACLE_OM_SENSOR_H_
#define ORACLE_OM_SENSOR_H_

#include <algorithm>
#include <cstddef>
#include <unordered_set>
#include <vector>

#include "om_hybrid_grid.h"

namespace oracle {

using PointCloud = std::vector<Vec3f>;

struct RangeData {
  Vec3f origin;
  PointCloud returns;
  PointCloud misses;
};

inline PointCloud TransformPointCloud(const PointCloud& cloud,
                                      const Rigid3f& transform) {
  PointCloud result;
  result.reserve(cloud.size());
  for (const Vec3f& p : cloud) result.push_back(transform * p);
  return result;
}

inline RangeData TransformRangeData(const RangeData& rd, const Rigid3f& t) {
  return RangeData{t * rd.origin, TransformPointCloud(rd.returns, t),
                   TransformPointCloud(rd.misses, t)};
}

// Keeps the first point that falls into each voxel.  The reference keys an
// unordered_set by a 96-bit concatenation of the three uint32 indices
// (voxel_filter.cc:119-124); any exact 3-int key gives the same output.
class VoxelFilter {
 public:
  explicit VoxelFilter(float size) : resolution_(size) {}

  PointCloud Filter(const PointCloud& cloud) {
    PointCloud kept;
    for (const Vec3f& p : cloud) {
      if (seen_.insert(Key(p)).second) kept.push_back(p);
    }
    return kept;
  }
  // Index form: which input points survive (same rule).
  std::vector<int> FilterIndices(const PointCloud& cloud) {
    std::vector<int> kept;
    for (size_t i = 0; i < cloud.size(); ++i) {
      if (seen_.insert(Key(cloud[i])).second) kept.push_back(static_cast<int>(i));
    }
    return kept;
  }

 private:
  struct K {
    int x, y, z;
    bool operator==(const K& o) const { return x == o.x && y == o.y && z == o.z; }
  };
  struct H {
    size_t operator()(const K& k) const {
      uint64_t h = static_cast<uint32_t>(k.x);
      h = h * 0x9E3779B97F4A7C15ull + static_cast<uint32_t>(k.y);
      h = h * 0x9E3779B97F4A7C15ull + static_cast<uint32_t>(k.z);
      return static_cast<size_t>(h ^ (h >> 29));
    }
  };
  K Key(const Vec3f& p) const {  // voxel_filter.cc:126-131
    return K{RoundToInt(p.x / resolution_), RoundToInt(p.y / resolution_),
             RoundToInt(p.z / resolution_)};
  }
  float resolution_;
  std::unordered_set<K, H> seen_;
};

struct AdaptiveVoxelFilterOptions {
  float max_length;   // proto field is float: adaptive_voxel_filter_options.proto
  float min_num_points;
  float max_range;
};

inline PointCloud FilterByMaxRange(const PointCloud& cloud, float max_range) {
  PointCloud result;
  for (const Vec3f& p : cloud) {
    if (p.norm() <= max_range) result.push_back(p);
  }
  return result;
}

// voxel_filter.cc:39-77
inline PointCloud AdaptivelyVoxelFiltered(const AdaptiveVoxelFilterOptions& o,
                                          const PointCloud& cloud) {
  if (cloud.size() <= o.min_num_points) return cloud;
  PointCloud result = VoxelFilter(o.max_length).Filter(cloud);
  if (result.size() >= o.min_num_points) return result;
  for (float high_length = o.max_length; high_length > 1e-2f * o.max_length;
       high_length /= 2.f) {
    float low_length = high_length / 2.f;
    result = VoxelFilter(low_length).Filter(cloud);
    if (result.size() >= o.min_num_points) {
      while ((high_length - low_length) / low_length > 1e-1f) {
        const float mid_length = (low_length + high_length) / 2.f;
        const PointCloud candidate = VoxelFilter(mid_length).Filter(cloud);
        if (candidate.size() >= o.min_num_points) {
          low_length = mid_length;
          result = candidate;
        } else {
          high_length = mid_length;
        }
      }
      return result;
    }
  }
  return result;
}

inline PointCloud AdaptiveVoxelFilter(const AdaptiveVoxelFilterOptions& o,
                                      const PointCloud& cloud) {
  return AdaptivelyVoxelFiltered(o, FilterByMaxRange(cloud, o.max_range));
}

// mapping/3d/range_data_inserter_3d.cc
class RangeDataInserter3D {
 public:
  RangeDataInserter3D(float hit_probability, float miss_probability,
                      int num_free_space_voxels)
      : num_free_space_voxels_(num_free_space_voxels),
        hit_table_(ComputeLookupTableToApplyOdds(Odds(hit_probability))),
        miss_table_(ComputeLookupTableToApplyOdds(Odds(miss_probability))) {}

  const std::vector<uint16>& hit_table() const { return hit_table_; }
  const std::vector<uint16>& miss_table() const { return miss_table_; }

  // range_data_inserter_3d.cc:78-92: all hits, then all misses, one marker epoch.
  void Insert(const RangeData& range_data, HybridGrid* grid) const {
    for (const Vec3f& hit : range_data.returns) {
      grid->ApplyLookupTable(grid->GetCellIndex(hit), hit_table_);
    }
    InsertMisses(range_data.origin, range_data.returns, grid);
    grid->FinishUpdate();
  }

 private:
  // range_data_inserter_3d.cc:27-52.  `delta * position / num_samples` is
  // Eigen Array3i arithmetic: int multiply, then C++ int division (truncating
  // toward zero) per component.
  void InsertMisses(const Vec3f& origin, const PointCloud& returns,
                    HybridGrid* grid) const {
    const Vec3i origin_cell = grid->GetCellIndex(origin);
    for (const Vec3f& hit : returns) {
      const Vec3i hit_cell = grid->GetCellIndex(hit);
      const Vec3i delta = hit_cell - origin_cell;
      const int num_samples =
          std::max(std::abs(delta.x), std::max(std::abs(delta.y), std::abs(delta.z)));
      if (!(num_samples < (1 << 15))) std::abort();  // CHECK_LT
      for (int position = std::max(0, num_samples - num_free_space_voxels_);
           position < num_samples; ++position) {
        const Vec3i miss_cell(origin_cell.x + delta.x * position / num_samples,
                              origin_cell.y + delta.y * position / num_samples,
                              origin_cell.z + delta.z * position / num_samples);
        grid->ApplyLookupTable(miss_cell, miss_table_);
      }
    }
  }

  const int num_free_space_voxels_;
  const std::vector<uint16> hit_table_;
  const std::vector<uint16> miss_table_;
};

// submap_3d.cc:42-51 (drops misses)
inline RangeData FilterRangeDataByMaxRange(const RangeData& rd, float max_range) {
  RangeData result{rd.origin, {}, {}};
  for (const Vec3f& hit : rd.returns) {
    if ((hit - rd.origin).norm() <= max_range) result.returns.push_back(hit);
  }
  return result;
}

}  // namespace oracle

#endif  // ORACLE_OM_SENSOR_H_
