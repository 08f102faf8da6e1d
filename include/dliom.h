/* dliom.h -- C ABI of the MI355X-native scan-to-submap front end.
 *
 * Drop-in boundary for the hot path of peterWon/D-LIOM's
 * cartographer::mapping::LocalTrajectoryBuilder3D.
 * Every entry point names the reference interface it replaces; paths are
 * relative to /root/reference/src/cartographer/cartographer.  The library is
 * libdliom.so (d-liom_amd/csrc, hand-written HIP for gfx950); there is no CPU
 * fallback behind these calls -- without a GPU they return DLIOM_ERR_NO_DEVICE.
 *
 * Conventions
 *   pose      double[7] = [tx,ty,tz,qw,qx,qy,qz]  (order of CeresPose::Data,
 *             mapping/internal/optimization/ceres_pose.h:47-51)
 *   points    packed float xyz, 12-byte stride == std::vector<Eigen::Vector3f>::data()
 *             (sensor/point_cloud.h:32)
 *   block     one 8x8x8 uint16 leaf of the reference HybridGrid, z-major
 *             ((z<<3)+y<<3)+x  (mapping/3d/hybrid_grid.h:40-43,66-138), addressed
 *             by the voxel index of its (0,0,0) corner (a multiple of 8 per axis)
 *   status    every function returns int: 0 ok, negative = the reference's
 *             CHECK condition that would have aborted (glog CHECK -> error code,
 *             SURVEY.md 8b "error convention"); nothing aborts or throws.
 *   threads   a dliom_ctx owns one HIP stream and its scratch memory; calls on
 *             different contexts are concurrency-safe (CeresScanMatcher3D::Match
 *             is re-entered from pool threads, constraint_builder_3d.cc:320).
 *             Grids are not thread-safe, like HybridGrid.
 */
#ifndef DLIOM_H_
#define DLIOM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DLIOM_OK 0
#define DLIOM_ERR_INVALID_ARGUMENT (-1) /* CHECK_NOTNULL (rtcsm_3d.cc:38, range_data_inserter_3d.cc:80) */
#define DLIOM_ERR_HIP (-2)              /* a HIP runtime call failed; see dliom_last_error() */
#define DLIOM_ERR_NO_DEVICE (-3)        /* no gfx950 device visible */
#define DLIOM_ERR_SCORE_NOT_POSITIVE (-4) /* CHECK_GT(score, 0.f) rtcsm_3d.cc:111 */
#define DLIOM_ERR_WEIGHTS (-5)          /* CHECK_EQ / CHECK_GT ceres_scan_matcher_3d.cc:89-92 */
#define DLIOM_ERR_GRID_EXTENT (-6)      /* CHECK_LE(new_bits, 8) hybrid_grid.h:389 */
#define DLIOM_ERR_RAY_TOO_LONG (-7)     /* CHECK_LT(num_samples, 1 << 15) range_data_inserter_3d.cc:38 */
#define DLIOM_ERR_EMPTY_CLOUD (-8)      /* division by size()==0 in ScoreCandidate / sqrt(N) scaling */
#define DLIOM_ERR_CAPACITY (-9)         /* caller-provided output buffer too small */
#define DLIOM_ERR_SOLVER (-10)          /* Ceres would report FAILURE (evaluation or invalid steps) */
#define DLIOM_ERR_PEER_FAILED (-12)     /* sharded match: another rank failed before the exchange (it still took part) */

typedef struct dliom_ctx dliom_ctx;
typedef struct dliom_grid dliom_grid;
typedef struct dliom_cloud dliom_cloud;
typedef struct dliom_inserter dliom_inserter;

const char* dliom_status_string(int status);
/* Text of the last HIP failure on the calling thread ("" if none). */
const char* dliom_last_error(void);
/* Number of visible HIP devices (0 when there is none). */
int dliom_device_count(void);

/* ---- context ------------------------------------------------------------ */
int dliom_ctx_create(int device_id, dliom_ctx** out);
/* Same, but work is enqueued on a caller-owned hipStream_t (e.g. torch's). */
int dliom_ctx_create_on_stream(int device_id, void* hip_stream, dliom_ctx** out);
int dliom_ctx_destroy(dliom_ctx* ctx);
int dliom_ctx_synchronize(dliom_ctx* ctx);
int dliom_ctx_device(const dliom_ctx* ctx); /* the device id it was created on (-1 for NULL) */

/* ---- probability value tables (host; mapping/probability_values.cc:73-83) --
 * table[v] = ProbabilityToValue(ProbabilityFromOdds(odds * Odds(p(v)))) + 32768,
 * what RangeDataInserter3D's constructor builds for hit and miss
 * (mapping/3d/range_data_inserter_3d.cc:70-76 with odds = Odds(float(p))). */
int dliom_compute_lookup_table_to_apply_odds(float odds, uint16_t* table32768);
float dliom_odds(float probability);
/* ProbabilityToValue (mapping/probability_values.h:91-93). */
uint16_t dliom_probability_to_value(float probability);
/* kValueToProbability (probability_values.cc:67-68), 65536 floats. */
int dliom_value_to_probability_table(float* table65536);

/* ---- device HybridGrid (replaces mapping/3d/hybrid_grid.h:470-547) -------- */
int dliom_grid_create(dliom_ctx* ctx, float resolution, dliom_grid** out);
int dliom_grid_destroy(dliom_grid* grid);
int dliom_grid_resolution(const dliom_grid* grid, float* resolution);
/* DynamicGrid::bits_ the reference would have after the same writes
 * (hybrid_grid.h:255,387-405). */
int dliom_grid_bits(const dliom_grid* grid, int* bits);
/* The correlative matcher's dense mirror of this grid: *rebuilds = how often it was (re)built since the
 * grid was created -- once for a whole-grid mirror, once per excursion of the search out of the mirrored window for
 * grids beyond bits = 4 --, *bytes = its current size (0: none), *windowed = 1 if it covers a window of the grid. */
int dliom_grid_mirror_stats(const dliom_grid* grid, int64_t* rebuilds, int64_t* bytes, int* windowed);
/* HBM accounting (no reference counterpart: the reference's grids live in host memory).  What a context's grids hold --
 * leaf tables ((8 << bits)^3 words), leaf pools (1 KiB + 12 B a leaf slot), the correlative matcher's dense mirrors
 * (272 MB at bits = 3, 2.2 GB at bits = 4, up to 4.04 GB for a window beyond that) -- and the
 * context's scratch buffers; nothing here synchronises with the device.  dliom_grid_memory_stats: one grid's share
 * (grids = 1, scratch_bytes = 0) with its leaf capacity and the host's upper bound of the slots in use.
 * dliom_ctx_set_mirror_budget: a cap on the sum of the context's mirrors (0 = none, the default).  A mirror that would
 * exceed it is not built and RealTimeCorrelativeScanMatcher3D runs its leaf-table kernel on that grid -- the same
 * results, several times slower (dliom_rtcsm_stats.box_kernel_status = DLIOM_BOX_REFUSED_NO_MIRROR, mirrors_refused
 * counts); mirrors that exist stay. */
typedef struct dliom_memory_stats {
  int64_t grids;
  int64_t leaf_table_bytes, leaf_pool_bytes, mirror_bytes;
  int64_t mirror_budget_bytes, mirrors_refused;
  int64_t scratch_bytes;           /* context only */
  int64_t leaf_capacity;           /* grid only: slots of the pool */
  int64_t leaf_slots_upper_bound;  /* grid only: >= slots in use (exact after dliom_grid_num_blocks) */
  int mirror_windowed;             /* grid only */
} dliom_memory_stats;
int dliom_ctx_memory_stats(const dliom_ctx* ctx, dliom_memory_stats* out);
int dliom_ctx_set_mirror_budget(dliom_ctx* ctx, int64_t bytes);
int dliom_grid_memory_stats(const dliom_grid* grid, dliom_memory_stats* out);
/* Overwrites whole leaves; grows like mutable_value()/Grow(). */
int dliom_grid_upload_blocks(dliom_grid* grid, const int32_t* block_origin_xyz,
                             const uint16_t* values512, int64_t num_blocks);
int dliom_grid_num_blocks(const dliom_grid* grid, int64_t* num_blocks);
/* Allocated leaves in unspecified order; *num_blocks <= capacity. */
int dliom_grid_download_blocks(const dliom_grid* grid, int32_t* block_origin_xyz,
                               uint16_t* values512, int64_t capacity, int64_t* num_blocks);
/* *mutable_value(cell) = value for n cells (HybridGrid::SetProbability with
 * value = ProbabilityToValue(p), hybrid_grid.h:489-491; also how HybridGrid(proto) loads cells,
 * hybrid_grid.h:475-486).  Grows / allocates like the reference. */
int dliom_grid_set_values(dliom_grid* grid, const int32_t* cell_xyz, const uint16_t* values, int64_t n);
/* mapping::proto::HybridGrid wire bytes (mapping/proto/3d/hybrid_grid.proto) of the grid -- what
 * HybridGrid::ToProto().SerializeAsString() yields (hybrid_grid.h:530-542: cells in iterator order) --
 * and back (the proto constructor, hybrid_grid.h:475-486: SetProbability(ValueToProbability(v)) per
 * entry).  buffer == NULL queries *size.  This is how finished submaps leave / enter the device for
 * pbstream files and Submap3D::ToProto / UpdateFromProto (submap_3d.cc:217-250). */
int dliom_grid_to_proto(const dliom_grid* grid, uint8_t* buffer, int64_t capacity, int64_t* size);
int dliom_grid_from_proto(dliom_ctx* ctx, const uint8_t* buffer, int64_t size, dliom_grid** out);
/* mapping::proto::Submap3D (mapping/proto/submap.proto) around two serialized HybridGrid messages -- the message
 * Submap3D::ToProto fills (submap_3d.cc:217-230: local_pose, num_range_data, finished, the grids when
 * include_probability_grid_data) and the Submap3D(proto) constructor / UpdateFromProto read (:207-215,232-250).
 * Host only; the grid bytes come from / go to dliom_grid_to_proto / dliom_grid_from_proto.
 *   to_proto    *_grid_proto == NULL: that grid is absent (include_probability_grid_data == false);
 *               wrap_in_submap != 0: the bytes of proto::Submap{submap_3d = ...} as Submap3D::ToProto(proto::Submap*) emits;
 *               scalar fields equal to zero are omitted as the C++ proto3 runtime does; buffer == NULL queries *size
 *   from_proto  the grids are returned as (offset, size) into `buffer`, size -1 = absent */
int dliom_submap3d_to_proto(const double local_pose7[7], int32_t num_range_data, int finished,
                            const uint8_t* high_resolution_grid_proto, int64_t high_size,
                            const uint8_t* low_resolution_grid_proto, int64_t low_size, int wrap_in_submap,
                            uint8_t* buffer, int64_t capacity, int64_t* size);
int dliom_submap3d_from_proto(const uint8_t* buffer, int64_t size, int wrapped_in_submap, double local_pose7[7],
                              int32_t* num_range_data, int* finished, int64_t* high_offset, int64_t* high_size,
                              int64_t* low_offset, int64_t* low_size);
/* HybridGrid::value() for n cell indices (0 outside / unallocated). */
int dliom_grid_get_values(const dliom_grid* grid, const int32_t* cell_xyz, int64_t n,
                          uint16_t* values);
/* RangeDataInserter3D::Insert(range_data{origin, returns, {}}, grid)
 * (mapping/3d/range_data_inserter_3d.cc:78-92): hits with hit_table, then for
 * each ray the last num_free_space_voxels cells with miss_table, each cell at
 * most once per call, then FinishUpdate(). */
int dliom_grid_insert(dliom_grid* grid, const float origin[3], const float* returns_xyz,
                      int64_t num_returns, const uint16_t* hit_table32768,
                      const uint16_t* miss_table32768, int num_free_space_voxels);

/* ---- RangeDataInserter3D object (mapping/3d/range_data_inserter_3d.h:35-47) ---
 * The constructor builds the hit and miss odds tables once
 * (range_data_inserter_3d.cc:70-76) and keeps them in HBM. */
int dliom_inserter_create(dliom_ctx* ctx, double hit_probability, double miss_probability,
                          int num_free_space_voxels, dliom_inserter** out);
int dliom_inserter_destroy(dliom_inserter* inserter);
/* Host copies of hit_table_ / miss_table_ (32768 entries each; either may be NULL). */
int dliom_inserter_tables(const dliom_inserter* inserter, uint16_t* hit_table32768,
                          uint16_t* miss_table32768);
/* RangeDataInserter3D::Insert(RangeData{origin, returns, {}}, grid). */
int dliom_inserter_insert(const dliom_inserter* inserter, dliom_grid* grid, const float origin[3],
                          const float* returns_xyz, int64_t num_returns);
/* Submap3D::InsertRangeData's data path on the device (mapping/3d/submap_3d.cc:264-279):
 * sensor::TransformRangeData through num_poses (0..2) float poses applied in sequence
 * (poses7[k] = [tx,ty,tz,qw,qx,qy,qz]; e.g. tracking->local then local->submap), then
 * FilterRangeDataByMaxRange(max_range) when max_range > 0 (submap_3d.cc:42-51), then Insert.
 * `origin` is given in the cloud's own frame and is transformed the same way. */
int dliom_inserter_insert_cloud(const dliom_inserter* inserter, dliom_grid* grid,
                                const float* poses7, int num_poses, const float origin[3],
                                const dliom_cloud* cloud, float max_range);

/* The same for up to 4 grids in one set of launches -- the four insertions of
 * ActiveSubmaps3D::InsertRangeData (both grids of both active submaps, submap_3d.cc:303-309).
 * Target k uses poses7[14*k .. 14*k + 7*num_poses[k]) and max_range[k].
 * The insertion calls (this one, _insert_cloud, dliom_front_end_insert*) return as soon as the device has told the
 * host whether the grids can hold the scan (status, growth); the update passes themselves are still on the context's
 * stream then.  Every later call ON THE SAME CONTEXT is ordered behind them; the cloud may be destroyed right away
 * (dliom_cloud_destroy waits for the device).  Code that reads a grid through ANOTHER context (a loop-closure thread
 * building a matcher on an active submap) calls dliom_ctx_synchronize on the inserting context first -- finished
 * submaps handed out by dliom_front_end_take_finished_submap need nothing, they have been synchronised. */
int dliom_inserter_insert_cloud_multi(const dliom_inserter* inserter, int num_targets,
                                      dliom_grid* const* grids, const float* poses7, const int* num_poses,
                                      const float origin[3], const dliom_cloud* cloud,
                                      const float* max_range);

/* ---- device-resident point cloud (sensor::PointCloud staged in HBM) ------- */
int dliom_cloud_create(dliom_ctx* ctx, const float* points_xyz, int64_t n, dliom_cloud** out);
int dliom_cloud_destroy(dliom_cloud* cloud);
int dliom_cloud_size(const dliom_cloud* cloud, int64_t* n);

/* ---- RealTimeCorrelativeScanMatcher3D -------------------------------------
 * proto::RealTimeCorrelativeScanMatcherOptions
 * (mapping/proto/scan_matching/real_time_correlative_scan_matcher_options.proto) */
typedef struct dliom_rtcsm_options {
  double linear_search_window;
  double angular_search_window;
  double translation_delta_cost_weight;
  double rotation_delta_cost_weight;
} dliom_rtcsm_options;

/* float RealTimeCorrelativeScanMatcher3D::Match(initial_pose_estimate,
 * point_cloud, hybrid_grid, pose_estimate)
 * (mapping/internal/3d/scan_matching/real_time_correlative_scan_matcher_3d.h:47-50,
 *  .cc:34-53).  *score receives the returned best score. */
int dliom_rtcsm3d_match(dliom_ctx* ctx, const dliom_rtcsm_options* options,
                        const double initial_pose_estimate[7], const float* points_xyz,
                        int64_t n, const dliom_grid* grid, double pose_estimate[7],
                        float* score);
/* Same with the cloud already in HBM. */
int dliom_rtcsm3d_match_cloud(dliom_ctx* ctx, const dliom_rtcsm_options* options,
                              const double initial_pose_estimate[7], const dliom_cloud* cloud,
                              const dliom_grid* grid, double pose_estimate[7], float* score);

/* The same match with the search window sharded across GPUs (BASELINE config 4): every rank
 * holds the cloud and the grid, owns the rotations [shard*R/num_shards, (shard+1)*R/num_shards)
 * (all translations of each), and the ranks exchange two words:
 *   begin  -> local best score LOWER BOUND (float bits, non-negative: orders like an integer)
 *             ... all-reduce MAX over ranks ...
 *   finish <- that global bound; -> local winner packed as (score_bits << 32) | (0xFFFFFFFF - index)
 *             ... all-reduce MAX over ranks (the lower index wins ties, like the reference's
 *             first strictly greater score in generation order, rtcsm_3d.cc:46-51) ...
 *   decode <- the global word; -> pose_estimate and score, identical on every rank.
 * With num_shards == 1 the three calls equal dliom_rtcsm3d_match_cloud. */
int dliom_rtcsm3d_shard_begin(dliom_ctx* ctx, const dliom_rtcsm_options* options,
                              const double initial_pose_estimate[7], const dliom_cloud* cloud,
                              const dliom_grid* grid, int shard, int num_shards,
                              uint32_t* local_best_lower_bound_bits);
int dliom_rtcsm3d_shard_finish(dliom_ctx* ctx, uint32_t global_best_lower_bound_bits,
                               uint64_t* local_best_packed);
int dliom_rtcsm3d_shard_decode(dliom_ctx* ctx, uint64_t global_best_packed, double pose_estimate[7],
                               float* score);

/* Search-window geometry of the last/next match (rtcsm_3d.cc:58-70). */
typedef struct dliom_rtcsm_window {
  int linear_window_size;
  int angular_window_size;
  float angular_step_size;
  float max_scan_range;
  int64_t num_translations; /* (2L+1)^3 */
  int64_t num_rotations;    /* (2A+1)^3 */
  int64_t num_candidates;
} dliom_rtcsm_window;
int dliom_rtcsm3d_window(const dliom_rtcsm_options* options, float resolution,
                         const float* points_xyz, int64_t n, dliom_rtcsm_window* window);

/* Statistics of the last dliom_rtcsm3d_match* on this context. */
typedef struct dliom_rtcsm_stats {
  dliom_rtcsm_window window;
  int64_t num_points;
  int64_t num_rescored;   /* candidates re-scored with the sequential float sum */
  int64_t best_index;     /* generation order: ((z,y,x) * R + (rz,ry,rx)) */
  int64_t score_kernel;   /* score-volume kernel that ran: 3 LDS-box, 2 dense mirror, 1 rotation per lane over the leaf
                             table, 0 point per lane over the leaf table */
  int64_t box_kernel_status; /* why: DLIOM_BOX_* below (a refusal is not an error -- the other kernels give the same
                                bits 1.9 to 4.4 times slower -- but it should not go unnoticed) */
  int64_t box_kernel_variant; /* score_kernel == 3: which instantiation ran -- 0: 27 translations per pass at four waves per
                                 SIMD (one-pass windows), 1: the same at three waves per SIMD with larger boxes, 2: 49
                                 translations per pass (windows of many passes); -1 otherwise */
} dliom_rtcsm_stats;
enum {
  DLIOM_BOX_RAN = 0,               /* the LDS-box kernel scored the volume */
  DLIOM_BOX_NOT_REQUESTED = 1,     /* DLIOM_TUNE_SCORE_KERNEL chose another kernel */
  DLIOM_BOX_REFUSED_SMALL = 2,     /* fewer than 8 translations or 2^24 candidate-point pairs: launch-bound anyway */
  DLIOM_BOX_REFUSED_NO_MIRROR = 3, /* the grid has no dense mirror (bits >= 5, or no memory for it) */
  DLIOM_BOX_REFUSED_RANGE = 4,     /* the scan's range beyond the fast index's (max ||p|| / resolution > ~860 cells); the
                                      POSITION in the grid is not limited (any cell DynamicGrid addresses) */
  DLIOM_BOX_REFUSED_WINDOW = 5,    /* angular window x range: a typical point's lookups do not fit a box */
  DLIOM_BOX_REFUSED_LDS = 6,       /* the box + lists exceed the LDS budget */
  DLIOM_BOX_REFUSED_FLAGGED = 7    /* the box kernel flagged an inconsistency ("cannot happen"): this match was redone
                                      on the dense kernel (dliom_rtcsm3d_box_error holds the sticky flag) */
};
int dliom_rtcsm3d_last_stats(const dliom_ctx* ctx, dliom_rtcsm_stats* stats);
/* BASELINE config 4 in one call: this rank scores rotations [shard R / num_shards, (shard + 1) R / num_shards), finds its
 * own winner exactly, and ONE max all-reduce of one uint64 (score_bits << 32 | ~candidate_index) yields the reference's
 * winner on every rank (rtcsm_3d.cc:46-51: first strictly greater score in generation order).
 *   _sharded       the collective is the caller's: `exchange` replaces *value by the maximum over all ranks (0 = ok)
 *   _sharded_rccl  `nccl_comm` is an ncclComm_t (RCCL, resolved with dlopen at first use; rank and size come from the
 *                  communicator): ncclAllReduce(ncclMax, ncclUint64, count 1) on the context's stream
 * A rank whose local part fails still takes part in the exchange (it contributes the reserved word
 * 0x7FFFFFFFFFFFFFFF, above every real word under signed or unsigned MAX) and returns its own status afterwards;
 * every other rank returns DLIOM_ERR_PEER_FAILED.  No rank is ever left waiting in the collective. */
typedef int (*dliom_allreduce_max_u64)(uint64_t* value, void* user);
int dliom_rtcsm3d_match_sharded(dliom_ctx* ctx, const dliom_rtcsm_options* options, const double initial_pose_estimate[7],
                                const dliom_cloud* cloud, const dliom_grid* grid, int shard, int num_shards,
                                dliom_allreduce_max_u64 exchange, void* user, double pose_estimate[7], float* score);
int dliom_rtcsm3d_match_sharded_rccl(dliom_ctx* ctx, const dliom_rtcsm_options* options,
                                     const double initial_pose_estimate[7], const dliom_cloud* cloud, const dliom_grid* grid,
                                     void* nccl_comm, double pose_estimate[7], float* score);
/* Diagnostic: sticky consistency flags of the LDS-box score kernel on this context (0 = every exactly
 * resolved lookup fell inside its staged box, as the construction guarantees).  Synchronises. */
int dliom_rtcsm3d_box_error(dliom_ctx* ctx, uint32_t* flags);

/* ---- CeresScanMatcher3D ------------------------------------------------------
 * proto::CeresScanMatcherOptions3D + common.proto.CeresSolverOptions
 * (mapping/proto/scan_matching/ceres_scan_matcher_options_3d.proto,
 *  common/proto/ceres_solver_options.proto). */
#define DLIOM_MAX_CLOUDS 8
typedef struct dliom_csm_options {
  int num_occupied_space_weights;
  double occupied_space_weight[DLIOM_MAX_CLOUDS];
  double translation_weight;
  double rotation_weight;
  int only_optimize_yaw;
  int use_nonmonotonic_steps;
  int max_num_iterations;
  int num_threads; /* accepted, unused: the device evaluates all points at once */
} dliom_csm_options;

/* The fields of ceres::Solver::Summary the path reads
 * (local_trajectory_builder_3d.cc:543) plus evaluation counts. */
typedef struct dliom_csm_summary {
  double initial_cost;
  double final_cost;
  int num_successful_steps;
  int num_unsuccessful_steps;
  int num_iterations;
  int num_residual_evaluations;
  int num_jacobian_evaluations;
  int termination_type; /* 0 CONVERGENCE, 1 NO_CONVERGENCE, 2 FAILURE */
} dliom_csm_summary;

/* void CeresScanMatcher3D::Match(target_translation, initial_pose_estimate,
 * point_clouds_and_hybrid_grids, pose_estimate, summary)
 * (mapping/internal/3d/scan_matching/ceres_scan_matcher_3d.h:51-56, .cc:71-123). */
int dliom_csm3d_match(dliom_ctx* ctx, const dliom_csm_options* options,
                      const double target_translation[3], const double initial_pose_estimate[7],
                      int num_clouds, const float* const* points_xyz, const int64_t* n,
                      const dliom_grid* const* grids, double pose_estimate[7],
                      dliom_csm_summary* summary);
int dliom_csm3d_match_cloud(dliom_ctx* ctx, const dliom_csm_options* options,
                            const double target_translation[3],
                            const double initial_pose_estimate[7], int num_clouds,
                            const dliom_cloud* const* clouds, const dliom_grid* const* grids,
                            double pose_estimate[7], dliom_csm_summary* summary);

/* ---- scan-to-submap front end --------------------------------------------------------------
 * LocalTrajectoryBuilder3D::AddAccumulatedRangeData + InsertIntoSubmap
 * (mapping/internal/3d/local_trajectory_builder_3d.cc:493-622) around ActiveSubmaps3D
 * (mapping/3d/submap_3d.cc:281-326), WITHOUT the GTSAM window: the caller runs WindowOptimize
 * between dliom_front_end_match() and dliom_front_end_insert(), exactly where the reference does
 * (:555-557), and passes the optimised pose to insert. */
typedef struct dliom_front_end dliom_front_end;

typedef struct dliom_adaptive_voxel_filter_options { /* sensor/proto/adaptive_voxel_filter_options.proto */
  float max_length;
  float min_num_points;
  float max_range;
} dliom_adaptive_voxel_filter_options;

typedef struct dliom_front_end_options { /* proto/3d/local_trajectory_builder_options_3d.proto (subset) */
  dliom_adaptive_voxel_filter_options high_resolution_adaptive_voxel_filter;
  dliom_adaptive_voxel_filter_options low_resolution_adaptive_voxel_filter;
  int use_online_correlative_scan_matching;
  dliom_rtcsm_options real_time_correlative_scan_matcher;
  dliom_csm_options ceres_scan_matcher;
  /* proto/motion_filter_options.proto */
  double motion_filter_max_time_seconds;
  double motion_filter_max_distance_meters;
  double motion_filter_max_angle_radians;
  /* proto/3d/submaps_options_3d.proto */
  double high_resolution;
  double high_resolution_max_range;
  double low_resolution;
  int num_range_data;
  double hit_probability;
  double miss_probability;
  int num_free_space_voxels;
} dliom_front_end_options;

typedef struct dliom_match_result {
  int dropped;                        /* 1: the reference returns nullptr (:497-500,510-513,531-534) */
  double pose_estimate[7];            /* local frame: submap.local_pose * observation (:552-553) */
  double pose_observation_in_submap[7];
  double initial_ceres_pose[7];       /* after the optional RTCSM (:504-521) */
  float rtcsm_score;                  /* 0 when the online matcher is off */
  dliom_csm_summary summary;
  double residual_distance;           /* :544-547 */
  double residual_angle;              /* :548-551 */
  int64_t num_high_resolution_points; /* after the adaptive voxel filters (:506-509,526-530) */
  int64_t num_low_resolution_points;
  int matching_submap_index;
} dliom_match_result;

typedef struct dliom_insertion_result {
  int inserted;                 /* 0: MotionFilter::IsSimilar said "similar" (:593-595) */
  int num_insertion_submaps;    /* submaps the range data went into (1 or 2) */
  int insertion_submap_index[2];/* trajectory-wide submap indices */
  int submap_added;             /* a new submap was started after this insertion */
  int submap_finished;          /* the oldest active submap was finished and dropped */
} dliom_insertion_result;

int dliom_front_end_create(dliom_ctx* ctx, const dliom_front_end_options* options, dliom_front_end** out);
int dliom_front_end_destroy(dliom_front_end* fe);
/* filtered_range_data_in_tracking = {origin, returns} (misses are not used by the 3D path).
 * pose_prediction: tracking frame -> local frame. */
int dliom_front_end_match(dliom_front_end* fe, const double pose_prediction[7], const float origin[3],
                          const float* returns_xyz, int64_t num_returns, dliom_match_result* result);
/* The same with the range data already resident on the device (e.g. the output of
 * dliom_cloud_voxel_filter); `returns` must stay alive until the following dliom_front_end_insert. */
int dliom_front_end_match_cloud(dliom_front_end* fe, const double pose_prediction[7], const float origin[3],
                                const dliom_cloud* returns, dliom_match_result* result);
/* InsertIntoSubmap with the range data of the preceding match.  time_ticks: cartographer
 * common::Time ticks (100 ns).  gravity_alignment: quaternion (w,x,y,z) used for new submaps. */
int dliom_front_end_insert(dliom_front_end* fe, int64_t time_ticks, const double pose_estimate[7],
                           const double gravity_alignment[4], dliom_insertion_result* result);
/* ActiveSubmaps3D::submaps() / matching_index(). */
int dliom_front_end_num_active_submaps(const dliom_front_end* fe, int* n);
int dliom_front_end_matching_index(const dliom_front_end* fe, int* index);
/* ActiveSubmaps3D::InsertRangeData(range_data, gravity_alignment) (mapping/3d/submap_3d.h:111-112, .cc:296-314) on the
 * front end's active submaps: range data already in the LOCAL frame, no MotionFilter. */
int dliom_front_end_insert_range_data(dliom_front_end* fe, const float origin_in_local[3],
                                      const dliom_cloud* returns_in_local, const double gravity_alignment[4],
                                      dliom_insertion_result* result);
/* Finished submaps (Submap3D::Finish(), submap_3d.cc:316-326) leave the active pair but stay alive for the back end:
 * the reference hands them on as shared_ptr and drops them when the pose graph is done.  Here the front end holds
 * them (leaf pools shrunk to the leaves in use, dense mirror released) until the caller TAKES them, oldest first;
 * a taken submap's grids belong to the caller (dliom_grid_destroy).  Not taking them keeps every finished submap
 * resident. */
int dliom_front_end_num_finished_submaps(const dliom_front_end* fe, int* n);
int dliom_front_end_take_finished_submap(dliom_front_end* fe, double local_pose[7], int* num_range_data,
                                         dliom_grid** high_resolution_grid, dliom_grid** low_resolution_grid);
/* The adaptively filtered clouds of the last successful match, in the tracking frame -- TrajectoryNode::Data's
 * high_resolution_point_cloud / low_resolution_point_cloud (local_trajectory_builder_3d.cc:506-533,613-619).  Borrowed:
 * valid until the next match on this front end; NULL when the last match dropped its scan. */
int dliom_front_end_matched_clouds(const dliom_front_end* fe, const dliom_cloud** high_resolution,
                                   const dliom_cloud** low_resolution);
int dliom_front_end_active_submap(const dliom_front_end* fe, int i, double local_pose[7], int* num_range_data,
                                  int* finished, dliom_grid** high_resolution_grid,
                                  dliom_grid** low_resolution_grid);
/* sensor::VoxelFilter::Filter / AdaptiveVoxelFilter::Filter (sensor/internal/voxel_filter.cc:81-90,
 * 28-77,147-150) on device-resident clouds: *out is a new cloud (dliom_cloud_destroy) holding the
 * survivors in input order -- the first point of every voxel lround(p / size).  Bit-identical to
 * the host filters below for |p / size| < 2^20 per axis, DLIOM_ERR_INVALID_ARGUMENT beyond. */
int dliom_cloud_voxel_filter(dliom_ctx* ctx, const dliom_cloud* in, float size, dliom_cloud** out);
int dliom_cloud_adaptive_voxel_filter(dliom_ctx* ctx, const dliom_cloud* in,
                                      const dliom_adaptive_voxel_filter_options* options, dliom_cloud** out);
/* The two AdaptiveVoxelFilter::Filter calls of LocalTrajectoryBuilder3D::AddAccumulatedRangeData
 * (local_trajectory_builder_3d.cc:507-512 high resolution, :523-533 low resolution) on the same cloud, searched
 * together: each result equals dliom_cloud_adaptive_voxel_filter() with that option set, for the launch and
 * readback round trips of one call. */
int dliom_cloud_adaptive_voxel_filter_pair(dliom_ctx* ctx, const dliom_cloud* in,
                                           const dliom_adaptive_voxel_filter_options* first,
                                           const dliom_adaptive_voxel_filter_options* second, dliom_cloud** out_first,
                                           dliom_cloud** out_second);
/* The points of a cloud in input order, packed xyz (room for dliom_cloud_size points). */
int dliom_cloud_download(const dliom_cloud* cloud, float* points_xyz);
/* ... moved by a float pose [tx, ty, tz, qw, qx, qy, qz]: sensor::TransformPointCloud(cloud, pose) (sensor/point_cloud.cc:25-33,
 * rotation * p + translation in Eigen's operation order) -- LocalTrajectoryBuilder3D's
 * filtered_range_data_in_local = TransformRangeData(filtered_range_data_in_tracking, opt_pose.cast<float>())
 * (local_trajectory_builder_3d.cc:556-559) without a host loop over the returns. */
int dliom_cloud_download_transformed(const dliom_cloud* cloud, const float pose[7], float* points_xyz);
/* The same filters on host buffers (the reference's own placement): out_xyz has room for n points;
 * *num_out receives the survivors (first point per voxel). */
int dliom_voxel_filter(float size, const float* points_xyz, int64_t n, float* out_xyz, int64_t* num_out);
int dliom_adaptive_voxel_filter(const dliom_adaptive_voxel_filter_options* options, const float* points_xyz,
                                int64_t n, float* out_xyz, int64_t* num_out);

/* ---- per-hit de-skew of LocalTrajectoryBuilder3D::AddRangeData -------------------------------
 * (mapping/internal/3d/local_trajectory_builder_3d.cc:421-472, InterpolatePose :869-877).
 * hits_xyzt: n x (x, y, z, t) in the tracking frame, t <= 0 seconds relative to the scan end (the
 * hits that survived VoxelFilter(0.5 * voxel_filter_size), :393-395).  For every hit: s = (T + t)/T,
 * pose_i = (prev_pose * [s t_rel, slerp(I, q_rel, s)]).cast<float>() with rel = prev^-1 * predicted,
 * hit and origin moved into the local frame, then the range gate: out_kind 0 = dropped
 * (range < min_range), 1 = return (out_xyz = hit_in_local), 2 = miss (out_xyz = ray cropped at
 * max_range).  If |t_0| < 1e-3 every hit takes the predicted pose ("not de-skewing", :429-433).
 * current_pose (float, [t,q]) = the last hit's pose (:477).  Slerp coefficients use the device's
 * double-precision sin/acos, which may differ from glibc's in the last ulp of a DOUBLE; the per-hit pose is cast to
 * float right after (:446).  Every hit whose cast could change under that difference is re-examined on the host with
 * glibc (dliom_deskew_check_stats below): the floats are the reference's by proof, per call. */
int dliom_deskew(dliom_ctx* ctx, const double prev_pose[7], const double predicted_pose[7], double scan_period,
                 const float* hits_xyzt, int64_t n, const float origin[3], float min_range, float max_range,
                 float* out_xyz, uint8_t* out_kind, float current_pose[7]);
/* The de-skew's float casts are PROVEN equal to the reference's, not sampled: the per-hit pose is computed in double with
 * the device's sin / acos, which may differ from glibc's in the last bits; a hit whose cast to float could change under
 * that difference (a bound derived from the libraries' documented errors, preprocess.hip) is recorded, recomputed on the
 * host with glibc -- the reference's own arithmetic -- and, should the floats differ, redone with the host's value.
 * *records_checked: hits re-examined on the host so far on this context; *ring_overflows: launches with more records
 * than ride along in the regular read-back (a second pass collects them); *hits_fixed: hits whose device cast differed
 * from glibc's (none has ever been observed). */
int dliom_deskew_check_stats(const dliom_ctx* ctx, int64_t* records_checked, int64_t* ring_overflows, int64_t* hits_fixed);

/* The whole pre-processing chain of AddRangeData on the device (:393-487), one scan per call
 * (num_accumulated_range_data = 1): ranges_xyzt (host, n x (x,y,z,t)) -> VoxelFilter(0.5 *
 * voxel_filter_size) -> de-skew + range gate as in dliom_deskew -> returns only (the 3D inserter
 * never reads RangeData::misses) -> VoxelFilter(voxel_filter_size) -> TransformRangeData by
 * current_pose.inverse().  *returns_in_tracking is a new device cloud (dliom_cloud_destroy), ready
 * for dliom_front_end_match_cloud; origin_in_tracking / current_pose as the reference computes them. */
int dliom_add_range_data(dliom_ctx* ctx, const double prev_pose[7], const double predicted_pose[7],
                         double scan_period, const float* ranges_xyzt, int64_t n, const float origin[3],
                         float min_range, float max_range, float voxel_filter_size,
                         dliom_cloud** returns_in_tracking, float origin_in_tracking[3], float current_pose[7]);
/* num_accumulated_range_data > 1 (local_trajectory_builder_3d.cc:449-476) and the RangeDataSynchronizer's two-lidar
 * output (internal/3d/range_data_synchronizer.cc:29-130): _add is one AddRangeData call up to the accumulation -- the
 * returns are de-skewed into the local frame and appended; origin_index (one float per range, NULL = all 0) selects
 * the sensor origin of `origins` (num_origins x 3, <= 4) a hit is gated against.  _finish is the tail
 * (VoxelFilter(size) over everything accumulated, TransformRangeData by the LAST current_pose^-1) and resets. */
typedef struct dliom_range_accumulator dliom_range_accumulator;
int dliom_range_accumulator_create(dliom_ctx* ctx, dliom_range_accumulator** out);
int dliom_range_accumulator_destroy(dliom_range_accumulator* accumulator);
int dliom_range_accumulator_add(dliom_range_accumulator* accumulator, const double prev_pose[7],
                                const double predicted_pose[7], double scan_period, const float* ranges_xyzt,
                                const float* origin_index, int64_t n, const float* origins, int num_origins,
                                float min_range, float max_range, float voxel_filter_size, float current_pose[7],
                                int* num_accumulated);
int dliom_range_accumulator_finish(dliom_range_accumulator* accumulator, float voxel_filter_size,
                                   dliom_cloud** returns_in_tracking, float origin_in_tracking[3]);

/* ---- FastCorrelativeScanMatcher3D (loop closure) --------------------------------
 * mapping/internal/3d/scan_matching/fast_correlative_scan_matcher_3d.h:100-132.  The constructor's
 * `nodes` arrive as what HistogramsAtAnglesFromNodes (:114-127) extracts from them: one rotational
 * histogram and one yaw per node.  The uint8 max-pool pyramid (PrecomputationGridStack3D, :57-77) is
 * built on the device at creation from `high_resolution_grid`; both grids must outlive the matcher
 * and stay unchanged (finished submaps).  Results equal the reference's: same candidate scores
 * (integer sums), same traversal order and tie handling (host recursion with the reference's
 * std::sort calls), low-resolution score by the exact sequential float sum. */
typedef struct dliom_fast_csm dliom_fast_csm;
typedef struct dliom_fast_csm_options { /* proto/scan_matching/fast_correlative_scan_matcher_options_3d.proto */
  int branch_and_bound_depth;
  int full_resolution_depth;
  double min_rotational_score;
  double min_low_resolution_score;
  double linear_xy_search_window;
  double linear_z_search_window;
  double angular_search_window;
} dliom_fast_csm_options;
typedef struct dliom_fast_csm_node_data { /* the fields of TrajectoryNode::Data the matcher reads */
  double gravity_alignment[4];                     /* w, x, y, z */
  const float* high_resolution_points;             /* packed xyz */
  int64_t num_high_resolution_points;
  const float* low_resolution_points;
  int64_t num_low_resolution_points;
  const float* rotational_scan_matcher_histogram;  /* histogram_size floats */
} dliom_fast_csm_node_data;
typedef struct dliom_fast_csm_result { /* FastCorrelativeScanMatcher3D::Result; found == 0 <=> nullptr */
  int found;
  float score;
  double pose_estimate[7];
  float rotational_score;
  float low_resolution_score;
  int num_discrete_scans;
  int64_t num_scored_candidates; /* candidates the device scored (incl. prefetched ones) */
  int64_t num_score_launches;
} dliom_fast_csm_result;
int dliom_fast_csm_create(dliom_ctx* ctx, const dliom_grid* high_resolution_grid, const dliom_grid* low_resolution_grid,
                          const float* node_histograms, const float* node_angles, int num_nodes, int histogram_size,
                          const dliom_fast_csm_options* options, dliom_fast_csm** out);
int dliom_fast_csm_destroy(dliom_fast_csm* matcher);
/* Match (:147-165), MatchFullSubmap (:204-232), MatchWith3DofInitial (:168-201). */
int dliom_fast_csm_match(dliom_ctx* ctx, const dliom_fast_csm* matcher, const double global_node_pose[7], const double global_submap_pose[7],
                         const dliom_fast_csm_node_data* constant_data, float min_score, dliom_fast_csm_result* result);
int dliom_fast_csm_match_full_submap(dliom_ctx* ctx, const dliom_fast_csm* matcher, const double global_node_rotation[4],
                                     const double global_submap_rotation[4], const dliom_fast_csm_node_data* constant_data,
                                     float min_score, dliom_fast_csm_result* result);
int dliom_fast_csm_match_with_3dof_initial(dliom_ctx* ctx, const dliom_fast_csm* matcher, const double pose_in_submap_guess[7],
                                           const dliom_fast_csm_node_data* constant_data, float min_score,
                                           dliom_fast_csm_result* result);
/* One level of the pyramid: a dense box of dims[0]*dims[1]*dims[2] uint8 (x fastest) whose element
 * (0,0,0) is cell lo[]; everything outside is 0.  values may be NULL to query the box only. */
int dliom_fast_csm_level(const dliom_fast_csm* matcher, int depth, int32_t lo[3], int32_t dims[3], uint8_t* values,
                         int64_t capacity);

/* RotationalScanMatcher::ComputeHistogram (rotational_scan_matcher.cc:159-170; called per inserted
 * scan, local_trajectory_builder_3d.cc:605-610) on the host: histogram_size floats.  Clouds of 8192 points and more
 * are processed on up to 8 host threads (same bits at any thread count; dliom_rotational_histogram_mt(..., forced_threads = 1, ...)
 * keeps the call on the caller's thread -- the library reads no environment variable). */
int dliom_rotational_histogram(const float* points_xyz, int64_t n, int histogram_size, float* histogram);
/* The same on the device, on a cloud that is already in HBM (the filtered cloud of the front end), fused with the
 * gravity alignment of local_trajectory_builder_3d.cc:605-610: histogram of Rigid3f::Rotation(rotation_wxyz) * point
 * (rotation_wxyz == NULL: the points as they are).  Bit-identical to the host function (additions in the reference's
 * order -- the sequential float sums are replayed in parallel, exactly; atan2f is glibc 2.35's float algorithm; equal
 * angles of one slice come out in the order libstdc++'s std::sort leaves them in).  histogram_size <= 255.  Slices of
 * any size: up to 4096 points of one 0.2 m slice are processed in LDS, larger ones (the floor of every real scan: 15 000
 * returns of a filtered 64-beam scan) in HBM.  Whether a cloud has such slices is only known on the device; the context
 * enqueues their kernels when the previous cloud had any, and runs a cloud again that needed them without having them.
 * DLIOM_ERR_CAPACITY: |z| >= 409.6 m, non-finite coordinates, more than 63 slices above 4096 points, or std::sort's
 * heap-sort fallback on a segment of more than 8192 elements of such a slice -- use the host function. */
int dliom_cloud_rotational_histogram(dliom_ctx* ctx, const dliom_cloud* cloud, const float rotation_wxyz[4],
                                     int histogram_size, float* histogram);
/* The same in two halves: _begin enqueues the kernels on an auxiliary stream of the context, behind everything the
 * context has been given so far, and returns; the caller may then put other work on the context (the adapter inserts
 * the scan into the submaps, local_trajectory_builder_3d.cc:590-604 -- neither step writes the filtered cloud);
 * _finish waits for the histogram only.  One histogram may be pending per context; the cloud must stay alive until
 * _finish.  Status of the limits (DLIOM_ERR_CAPACITY, see above) comes from _finish. */
int dliom_cloud_rotational_histogram_begin(dliom_ctx* ctx, const dliom_cloud* cloud, const float rotation_wxyz[4],
                                           int histogram_size);
int dliom_cloud_rotational_histogram_finish(dliom_ctx* ctx, float* histogram);
/* Diagnostic: indices of n float keys in the order the device's SortSlice leaves them -- libstdc++'s std::sort
 * on (key, index) pairs compared by key only, EQUAL keys included (introsort's partitions restated as data-parallel
 * rounds, its heap sort at the depth limit, and a stable sort for the final insertion sort).  n <= 4096: the LDS path of
 * the small slices; above: the path of the floor slices (radix sort + the partitions of the segments that hold ties --
 * in LDS on 32-bit (rank, position) items up to ~16 000 keys, on arrays in HBM beyond). */
int dliom_diag_std_sort_order(dliom_ctx* ctx, const float* keys, int n, int32_t* order);
/* Diagnostic: the exact parallel replay of sequential float sums the histogram uses for ComputeCentroid and
 * histogram(bucket) += value (rotational_scan_matcher.cc:49,52-59): k arrays of n floats (values: k x n, row major),
 * sums[i] = (((acc0[i] + v[0]) + v[1]) + ...) in exactly that order, computed by one workgroup per array. */
int dliom_diag_sequential_sums(dliom_ctx* ctx, const float* values, int k, int n, const float* acc0, float* sums);
/* Diagnostic: every `histogram(bucket) += value` of dliom_cloud_rotational_histogram as (bucket, value) in the order of
 * the additions (AddValueToHistogram, rotational_scan_matcher.cc:35-50: slices in key order, points in SortSlice's
 * order).  Stronger than comparing histograms: a bucket whose sum is in the hundreds hides a contribution of 1e-5 that
 * went elsewhere.  Blocking; stores the first `capacity` pairs, *count = how many there are. */
int dliom_diag_histogram_contributions(dliom_ctx* ctx, const dliom_cloud* cloud, const float rotation_wxyz[4],
                                       int histogram_size, int32_t* buckets, float* values, int64_t capacity, int64_t* count);
/* The same with an explicit number of host threads (0 = as many as pay, at most 8; the bits do not depend on it). */
int dliom_rotational_histogram_mt(const float* points_xyz, int64_t n, int histogram_size, int num_threads, float* histogram);
/* RotationalScanMatcher(histograms_at_angles).Match(histogram, initial_angle, angles)
 * (rotational_scan_matcher.cc:174-194), host: one score per angle.  The loop-closure matcher calls
 * the same code to pick the yaw candidates worth discretising. */
int dliom_rotational_scan_match(const float* node_histograms, const float* node_angles, int num_nodes,
                                int histogram_size, const float* scan_histogram, float initial_angle,
                                const float* angles, int num_angles, float* scores);

/* ---- IMU preintegration between scans (host; SURVEY 8f rank 4, PARITY UNPINNED) ------------------
 * Mid-point preintegration with bias Jacobians and covariance as the reference's in-tree
 * IntegrationBase (mapping/internal/3d/initialization/integration_base.h:106-278), plus the VINS-Mono
 * residual it keeps commented out (:280-316) and the matching state prediction -- a self-contained
 * stand-in for the gtsam::PreintegratedImuMeasurements calls of the steady-state path
 * (local_trajectory_builder_3d.cc:179-199).  Blocks of jacobian / covariance (row-major 15 x 15):
 * P 0, R 3, V 6, BA 9, BG 12.  States are [P(3), Q(w,x,y,z), V(3), Ba(3), Bg(3)]; gravity is the
 * world-frame vector G of the residual (e.g. {0, 0, 9.80511}). */
typedef struct dliom_imu_integrator dliom_imu_integrator;
typedef struct dliom_imu_noise { double acc_n, gyr_n, acc_w, gyr_w; } dliom_imu_noise;
typedef struct dliom_imu_preintegration {
  double sum_dt;
  double delta_p[3];
  double delta_q[4];
  double delta_v[3];
  double linearized_ba[3];
  double linearized_bg[3];
  double jacobian[225];
  double covariance[225];
} dliom_imu_preintegration;
int dliom_imu_integrator_create(const double ba[3], const double bg[3], const dliom_imu_noise* noise,
                                dliom_imu_integrator** out);
int dliom_imu_integrator_destroy(dliom_imu_integrator* integrator);
int dliom_imu_integrator_reset(dliom_imu_integrator* integrator, const double ba[3], const double bg[3],
                               const dliom_imu_noise* noise);
int dliom_imu_integrator_push_back(dliom_imu_integrator* integrator, double dt, const double acc[3],
                                   const double gyr[3]);
int dliom_imu_integrator_repropagate(dliom_imu_integrator* integrator, const double ba[3], const double bg[3]);
int dliom_imu_integrator_get(const dliom_imu_integrator* integrator, dliom_imu_preintegration* out);
int dliom_imu_integrator_evaluate(const dliom_imu_integrator* integrator, const double state_i[16],
                                  const double state_j[16], const double gravity[3], double residuals[15]);
int dliom_imu_integrator_predict(const dliom_imu_integrator* integrator, const double state_i[16],
                                 const double gravity[3], double state_j[16]);

/* ---- LocalTrajectoryBuilder3D::WindowOptimize: IMU-preintegration cost in the optimisation ----------
 * (host; PARITY UNPINNED: GTSAM 4.0.2 is not in the reference tree and the reference has no
 * test at this boundary).  Replaces mapping/internal/3d/local_trajectory_builder_3d.cc:693-863 and the
 * gtsam::PreintegratedImuMeasurements calls of AddImuData (:179-199): IMU preintegration (tangent form like the
 * reference's GTSAM build, or the manifold form: options.tangent_preintegration), ImuFactor +
 * bias BetweenFactor + PriorFactor<Pose3> on the matched pose + Pose3GravityFactor
 * (gravity_factor/gravity_factor.cc:10-33), solved as a fixed-lag Gauss-Newton smoother with marginalisation
 * in place of ISAM2.  Poses are [x, y, z, qw, qx, qy, qz], biases [ax, ay, az, gx, gy, gz].
 * Call order per scan, as the reference's AddImuData / AddRangeData / WindowOptimize:
 *   _add_imu (every IMU sample) ... _predict (initial pose for the matchers) ... _add_pose (matched pose)
 * DLIOM_ERR_DIVERGED mirrors FailureDetection (:856-859,896-913): |v| > 30 m/s or a bias norm > 1 after a scan.  The
 * outputs hold that scan's estimate and, like ResetParams(), the window only forgets that its graph was started: the next
 * _window_optimize starts a new graph at that estimate with the initial priors (the reference does exactly this and goes
 * on); a caller may re-initialise instead. */
#define DLIOM_ERR_DIVERGED (-11)
/* add_pose returns DLIOM_ERR_SOLVER when the normal equations cannot be factorised: the window is then exactly as before
 * the call (the new key, its factors and the gravity estimator's entry are taken back, the running preintegration kept). */
typedef struct dliom_imu_window dliom_imu_window;
typedef struct dliom_imu_window_options {
  double acc_noise, gyr_noise, acc_bias_noise, gyr_bias_noise; /* trajectory_builder_3d.lua:88-91 */
  double gravity;                                               /* :92, n_gravity = (0, 0, -gravity) */
  double integration_sigma;                                     /* 1e-4, local_trajectory_builder_3d.cc:81-82 */
  double prior_pose_noise;                                      /* lua :93 */
  double prior_velocity_sigma, prior_bias_sigma;                /* 1e4 / 1e-2, .cc:88-89 */
  double ceres_pose_noise_t, ceres_pose_noise_r;                /* lua :96-97 */
  double ceres_pose_noise_t_drift, ceres_pose_noise_r_drift;    /* lua :98-99 (is_drift) */
  double prior_gravity_noise;                                   /* lua :100 */
  int window_size;  /* 0: the reference's rule -- EVERY key since the last graph reset stays in the problem (needs
                       graph_reset_every >= 2), linearisation points move by relinearize_threshold, nothing is
                       marginalised (.cc:693-863 with ISAM2 as it is parameterised at :676-679);
                       2..4096: fixed-lag Gauss-Newton smoother, older states marginalised (Schur complement) */
  int iterations;   /* Gauss-Newton iterations per scan (the reference calls ISAM2::update twice) */
  int enable_gravity_factor;               /* lua :31 (false; dlio/config/basic_config_3d.lua:80 true): EstimateGravity per
                                              scan and, when it succeeds, a Pose3GravityFactor (.cc:819-831) */
  int frames_for_online_gravity_estimate;  /* lua :29 (7): estimator window, and the factor's key distance; needs
                                              window_size >= this + 1 */
  double lidar_in_imu_translation[3];      /* transform_lb_.translation() (.cc:1140), zero if the poses are the IMU's */
  int graph_reset_every;                   /* submaps.num_range_data (.cc:749-792): when key_ reaches it the reference
                                              replaces its graph by the newest state with the marginal covariances of
                                              pose, velocity and bias taken separately; 0 = never (plain fixed-lag
                                              smoothing, which keeps the cross-covariances that reset drops); >= 2 */
  int tangent_preintegration;              /* 1 (default): gtsam::TangentPreintegration, what PreintegratedImuMeasurements
                                              IS in the reference's build (README.MD:13-15 builds GTSAM 4.0.2 without
                                              -DGTSAM_TANGENT_PREINTEGRATION=OFF; 4.0.x defaults to ON): [theta, p, v]
                                              integrated in the tangent space of the first state, error = local
                                              coordinates of the predicted state at state j.  0: the manifold form of
                                              Forster et al. (GTSAM with the flag OFF).  Still PARITY UNPINNED either way */
  double relinearize_threshold;            /* window_size == 0 only: ISAM2Params::relinearizeThreshold (.cc:677, 0.1): a key's
                                              linearisation point -- X(i), V(i) and B(i) each by itself -- moves to its estimate
                                              when a component of ITS increment exceeds it (checked at each of the `iterations`
                                              updates, relinearizeSkip = 1); the
                                              estimate is linearisation point (+) increment, the increments solving the
                                              linearised problem exactly.  0 = every update relinearises every key = batch
                                              Gauss-Newton over the whole graph (oracle/imu_window_ref.py's
                                              ReferenceRuleSmoother idealisation) */
} dliom_imu_window_options;
int dliom_imu_window_default_options(dliom_imu_window_options* options);
int dliom_imu_window_create(const dliom_imu_window_options* options, dliom_imu_window** out);
int dliom_imu_window_destroy(dliom_imu_window* window);
/* InitializeIMU (:331-356) + the priors of the gtsam_initialized_ == false branch (:712-745): prev_state_ / prev_bias_,
 * X(0), V(0), B(0) with their priors */
int dliom_imu_window_initialize(dliom_imu_window* window, const double pose7[7], const double velocity[3],
                                const double bias6[6]);
/* imu_integrator_opt_->integrateMeasurement(acc, gyr, dt) (:188-196) */
int dliom_imu_window_add_imu(dliom_imu_window* window, const double acc[3], const double gyr[3], double dt);
/* the same for n samples (acc, gyr: n x 3; dt: n) in one call: for callers that buffer the IMU between two scans */
int dliom_imu_window_add_imu_batch(dliom_imu_window* window, int n, const double* acc, const double* gyr, const double* dt);
/* imu_integrator_opt_->predict(prev_state_, prev_bias_) (:198-199) */
int dliom_imu_window_predict(const dliom_imu_window* window, double pose7[7], double velocity[3]);
/* Pose3GravityFactor on the state `states_back` keys before the newest (:819-831); direction = estimated gravity */
int dliom_imu_window_add_gravity(dliom_imu_window* window, int states_back, const double direction[3]);
/* WindowOptimize(matched_pose, is_drift): new key, factors, optimisation; outputs prev_pose_ / prev_vel_ / prev_bias_ */
int dliom_imu_window_add_pose(dliom_imu_window* window, const double matched_pose7[7], int is_drift, double pose7[7],
                              double velocity[3], double bias6[6]);
/* Work done by the solver so far: linearisation points moved, chain blocks (states) eliminated -- what a scan costs */
int dliom_imu_window_solver_stats(const dliom_imu_window* window, int64_t* relinearizations, int64_t* blocks_eliminated);
/* WindowOptimize as the reference calls it, once per scan, the first call after _initialize included: that call only
 * starts the graph (gtsam_initialized_ == false, .cc:712-745: priors at the initial state, the preintegration since
 * InitializeIMU dropped) and returns the initial state -- the scan's matched pose is not used, as in the reference; every
 * later call is dliom_imu_window_add_pose.  (_add_pose alone adds a key on every call: the primitive the solver tests use.) */
int dliom_imu_window_window_optimize(dliom_imu_window* window, const double matched_pose7[7], int is_drift, double pose7[7],
                                     double velocity[3], double bias6[6]);
int dliom_imu_window_state(const dliom_imu_window* window, int states_back, double pose7[7], double velocity[3],
                           double bias6[6]);
int dliom_imu_window_size(const dliom_imu_window* window);
/* Diagnostic: Jacobian (15 x 30, row major) of the IMU factor + bias random walk between the window's two newest
 * states -- the closed form the solver uses and central differences of its residual.  Needs >= 2 states. */
int dliom_diag_imu_factor_jacobians(dliom_imu_window* window, double* analytic, double* numeric);
/* g_vec_est_G_ of the last EstimateGravity() (local_trajectory_builder_3d.cc:1106-1154), whether that call passed the
 * reference's gates, and how many gravity factors add_pose has added so far */
int dliom_imu_window_gravity_estimate(const dliom_imu_window* window, double gravity_in_global[3], int* valid,
                                      int64_t* factors_added);
/* GravityEstimator::Estimate (gravity_factor/gravity_estimator.cc:172-188) on explicit frames: pose [t, q(wxyz)] relative to
 * the first frame, the preintegration stored WITH each frame (deltaTij, deltaPij, deltaVij) and its body-frame velocity.
 * gravity_out is in the first frame (the reference's sign: it points up); *accepted = the function's return value. */
int dliom_gravity_estimate(int num_frames, const double* poses7, const double* delta_t, const double* delta_p,
                           const double* delta_v, const double* velocities, const double lidar_in_imu_translation[3],
                           double gravity_norm, double gravity_out[3], int* accepted);

/* ---- RealTimeCorrelativeScanMatcher2D (BASELINE config 1: host only, by contract) -------------
 * double Match(initial_pose_estimate, point_cloud, probability_grid, pose_estimate)
 * (mapping/internal/2d/scan_matching/real_time_correlative_scan_matcher_2d.h:66-69, .cc:74-108).
 * Poses are [x, y, theta].  The ProbabilityGrid is passed as what it stores: num_x_cells *
 * num_y_cells uint16 correspondence-cost cells, row-major num_x_cells * y + x
 * (mapping/2d/grid_2d.cc:168-171), with MapLimits{resolution, max} (mapping/2d/map_limits.h). */
int dliom_rtcsm2d_match(const dliom_rtcsm_options* options, const double initial_pose_estimate[3],
                        const float* points_xyz, int64_t n, const uint16_t* correspondence_cost_cells,
                        int num_x_cells, int num_y_cells, double resolution, double max_x, double max_y,
                        double pose_estimate[3], double* score);

/* ---- diagnostics used by the parity tests and bench.py ------------------------
 * These expose intermediate results of the same device code the matchers run. */
/* Cell index of R(pose)*p + t per point, computed by the score kernel's own
 * device function (bit-exactness probe for hybrid_grid.h:430-435). */
int dliom_probe_transform_cell_indices(dliom_ctx* ctx, const float pose[7],
                                       const float* points_xyz, int64_t n, float resolution,
                                       int32_t* cell_xyz);
/* Per candidate (generation order) the exact integer sum over points of
 * max(value & 0x7fff, 1): the order-independent score volume. */
int dliom_rtcsm3d_score_volume(dliom_ctx* ctx, const dliom_rtcsm_options* options,
                               const double initial_pose_estimate[7], const float* points_xyz,
                               int64_t n, const dliom_grid* grid, uint64_t* sums,
                               int64_t capacity, int64_t* num_candidates);
/* For k given candidates: the reference's SEQUENTIAL float sum of probabilities in point order
 * (rtcsm_3d.cc:101-104, before the division by N), method 0 = one lane replays the loop,
 * method 1 = the binade-wise exact parallel scan over elements, method 2 = the same scan over
 * precomputed 64-point chunk functions (both n <= 65536).  All three must be bit-identical. */
int dliom_rtcsm3d_sequential_sums(dliom_ctx* ctx, const dliom_rtcsm_options* options,
                                  const double initial_pose_estimate[7], const float* points_xyz, int64_t n,
                                  const dliom_grid* grid, const int64_t* candidate_indices, int64_t k,
                                  int method, float* sums);
/* One evaluation of the stacked problem at `pose`: cost = 1/2 sum r^2,
 * gradient[6] = J^T r and jtj[36] = J^T J (row-major) in the 6-dof tangent
 * space of (translation, QuaternionParameterization). */
int dliom_csm3d_evaluate(dliom_ctx* ctx, const dliom_csm_options* options,
                         const double target_translation[3],
                         const double initial_pose_estimate[7], const double pose[7],
                         int num_clouds, const float* const* points_xyz, const int64_t* n,
                         const dliom_grid* const* grids, double* cost, double gradient[6],
                         double jtj[36]);

/* Per-context choices a caller may make; none of them changes a result (every kernel variant is parity-tested).  The
 * library never reads the environment (tuning experiments live in `make experiments` builds only). */
enum {
  DLIOM_TUNE_SCORE_KERNEL = 0,        /* RTCSM3D score volume: 3 LDS-box kernel over the dense mirror when the search suits
                                         it (default), 2 rotation-per-lane over the dense mirror, 1 rotation-per-lane over
                                         the leaf table, 0 point-per-lane over the leaf table */
  DLIOM_TUNE_CSM_ONE_LAUNCH_MAX = 1,  /* CeresScanMatcher3D: clouds up to this many points (sum over grids) run the whole
                                         trust-region loop in one launch (default 4096, 0 = never) */
  DLIOM_TUNE_RESERVED_TEST_HOOK = 2,  /* reserved: dliom_ctx_set_tuning refuses it (DLIOM_ERR_INVALID_ARGUMENT).  Only a
                                         test build of the library (-DDLIOM_TEST_HOOKS, `make hooks` ->
                                         libdliom_hooks.so, loaded by one GPU test) accepts it: the next match then
                                         treats the box kernel's consistency word as set and takes the "redo on the
                                         dense kernel" path once; 2 / 3 force the de-skew check's overflow and fix
                                         paths.  The shipped library contains no fault injection. */
  DLIOM_TUNE_CSM_GRID_SYNC = 3,       /* CeresScanMatcher3D on large clouds: 1 = the whole loop in one launch with grid
                                         barriers, 0 = one launch per evaluation (default: measured 0.27 ms against
                                         0.35 ms per 131 072-point match -- the barrier, the final reduction and the
                                         LM step repeated by 256 workgroups cost more than the round trips they save;
                                         both give the same bits) */
  DLIOM_TUNE_COUNT = 4
};
int dliom_ctx_set_tuning(dliom_ctx* ctx, int knob, int value);
int dliom_ctx_get_tuning(const dliom_ctx* ctx, int knob, int* value);
/* Read-backs end in a completion word in coherent pinned host memory that the host polls (150 us), falling back to
 * hipStreamSynchronize: *count = how often that fallback was taken on this context.  A long kernel in front of a
 * read-back takes it legitimately; a count that grows with every call means the polling is not seeing the device's
 * writes (10x the latency, same results). */
int dliom_ctx_poll_fallbacks(const dliom_ctx* ctx, int64_t* count);
/* *count = polled read-backs on this context so far: the host round trips a caller's chain costs (a W-ref scan of the
 * front end: 8, profiles/r5_wref_full.json `read_backs_per_scan`). */
int dliom_ctx_read_backs(const dliom_ctx* ctx, int64_t* count);
/* The device voxel filter (sensor/internal/voxel_filter.cc:81-131) keeps "voxel -> first point" in a hash table whose
 * entries pack the voxel index (13 bits per axis) and the point index (24 bits) into one word, so that a voxel costs
 * one atomic.  A cloud with a point farther than 4095 voxel edges from the origin (61 m at 1.5 cm) does not fit: the
 * launch is repeated with 21-bit keys (same result).  *count = how often that happened on this context. */
int dliom_ctx_voxel_filter_reruns(const dliom_ctx* ctx, int64_t* count);
/* Page-locks a host buffer the caller keeps (a LiDAR driver's ring of scan buffers, the vector a
 * sensor::TimedPointCloudData is filled into): uploads from it -- dliom_add_range_data's scan, dliom_cloud_create's
 * points -- are then one asynchronous DMA instead of the runtime's staged copy of pageable memory (a 64 x 1024 scan of
 * 1 MB: ~38 us of a 0.55 ms W-ref scan, tools/wref_cpp.py --pinned-scans).  Registering costs hundreds of microseconds:
 * for buffers that live across scans, not per call.  Results do not depend on it.  Every entry point that takes such a
 * buffer has consumed it when it returns, pinned or not.  The reference has no counterpart (its data never leaves the host). */
int dliom_host_register(dliom_ctx* ctx, void* buffer, size_t bytes);
int dliom_host_unregister(dliom_ctx* ctx, void* buffer);

/* Kernel timing (HIP events on the context's stream). */
enum {
  DLIOM_KERNEL_RTCSM_SCORE = 0, /* score-volume kernel (dominant) */
  DLIOM_KERNEL_RTCSM_SELECT = 1,
  DLIOM_KERNEL_RTCSM_RESCORE = 2,
  DLIOM_KERNEL_CSM_EVAL = 3,
  DLIOM_KERNEL_INSERT = 4,
  DLIOM_KERNEL_ALLREDUCE = 5,   /* dliom_rtcsm3d_match_sharded_rccl: copy in, ncclAllReduce(max, u64), copy out -- the
                                   collective's own time on this rank's stream, the wait for the slowest peer included */
  DLIOM_KERNEL_COUNT = 6
};
/* enabled: 0 off, 1 every kernel id, otherwise a mask with bit (id + 1) per timed kernel id
 * (2 = the score kernel only: what bench.py's timed region uses; 2 | 64 = score kernel and the sharded match's
 * all-reduce). */
int dliom_ctx_set_profiling(dliom_ctx* ctx, int enabled);
int dliom_ctx_reset_profiling(dliom_ctx* ctx);
int dliom_ctx_kernel_time(dliom_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* DLIOM_H_ */
