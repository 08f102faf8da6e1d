"""Python mirror of the reference's operator surface over libdliom.so (C ABI in
include/dliom.h).  Class and method names follow cartographer::mapping so the
parity tests read like the reference's own tests:

    HybridGrid                          mapping/3d/hybrid_grid.h:470-547
    RangeDataInserter3D                 mapping/3d/range_data_inserter_3d.h:35-47
    RealTimeCorrelativeScanMatcher3D    .../real_time_correlative_scan_matcher_3d.h:34-66
    CeresScanMatcher3D                  .../ceres_scan_matcher_3d.h:37-63

Everything here calls the HIP library; there is NO CPU fallback.  Importing the
package without a built libdliom.so, or creating a Context without a GPU, fails
loudly (DliomError).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libdliom.so")


class DliomError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = _lib.dliom_status_string(status).decode() if _lib is not None else "library not loaded"
        detail = _lib.dliom_last_error().decode() if _lib is not None and status == -2 else ""
        super().__init__("%s: %s (%d) %s" % (where, msg, status, detail))


OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_HIP = -2
ERR_NO_DEVICE = -3
ERR_SCORE_NOT_POSITIVE = -4
ERR_WEIGHTS = -5
ERR_GRID_EXTENT = -6
ERR_RAY_TOO_LONG = -7
ERR_EMPTY_CLOUD = -8
ERR_CAPACITY = -9
ERR_SOLVER = -10
ERR_PEER_FAILED = -12
TUNE_SCORE_KERNEL, TUNE_CSM_ONE_LAUNCH_MAX, TUNE_RESERVED_TEST_HOOK, TUNE_CSM_GRID_SYNC = 0, 1, 2, 3
HOOKS_LIB_PATH = os.path.join(os.path.dirname(_HERE), "libdliom_hooks.so")  # -DDLIOM_TEST_HOOKS build (tests only)

KERNEL_RTCSM_SCORE, KERNEL_RTCSM_SELECT, KERNEL_RTCSM_RESCORE, KERNEL_CSM_EVAL, KERNEL_INSERT, KERNEL_ALLREDUCE = range(6)

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u16p = C.POINTER(C.c_uint16)
_u64p = C.POINTER(C.c_uint64)
_vp = C.c_void_p


class RtcsmOptions(C.Structure):
    _fields_ = [("linear_search_window", C.c_double), ("angular_search_window", C.c_double),
                ("translation_delta_cost_weight", C.c_double), ("rotation_delta_cost_weight", C.c_double)]


class RtcsmWindow(C.Structure):
    _fields_ = [("linear_window_size", C.c_int), ("angular_window_size", C.c_int),
                ("angular_step_size", C.c_float), ("max_scan_range", C.c_float),
                ("num_translations", C.c_int64), ("num_rotations", C.c_int64),
                ("num_candidates", C.c_int64)]


class RtcsmStats(C.Structure):
    _fields_ = [("window", RtcsmWindow), ("num_points", C.c_int64), ("num_rescored", C.c_int64),
                ("best_index", C.c_int64), ("score_kernel", C.c_int64), ("box_kernel_status", C.c_int64),
                ("box_kernel_variant", C.c_int64)]


BOX_RAN, BOX_NOT_REQUESTED, BOX_REFUSED_SMALL, BOX_REFUSED_NO_MIRROR = 0, 1, 2, 3
BOX_REFUSED_RANGE, BOX_REFUSED_WINDOW, BOX_REFUSED_LDS, BOX_REFUSED_FLAGGED = 4, 5, 6, 7


MAX_CLOUDS = 8


class CsmOptions(C.Structure):
    _fields_ = [("num_occupied_space_weights", C.c_int), ("occupied_space_weight", C.c_double * MAX_CLOUDS),
                ("translation_weight", C.c_double), ("rotation_weight", C.c_double),
                ("only_optimize_yaw", C.c_int), ("use_nonmonotonic_steps", C.c_int),
                ("max_num_iterations", C.c_int), ("num_threads", C.c_int)]


class CsmSummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("num_successful_steps", C.c_int), ("num_unsuccessful_steps", C.c_int),
                ("num_iterations", C.c_int), ("num_residual_evaluations", C.c_int),
                ("num_jacobian_evaluations", C.c_int), ("termination_type", C.c_int)]


class AdaptiveVoxelFilterOptions(C.Structure):
    _fields_ = [("max_length", C.c_float), ("min_num_points", C.c_float), ("max_range", C.c_float)]


class FrontEndOptions(C.Structure):
    _fields_ = [("high_resolution_adaptive_voxel_filter", AdaptiveVoxelFilterOptions),
                ("low_resolution_adaptive_voxel_filter", AdaptiveVoxelFilterOptions),
                ("use_online_correlative_scan_matching", C.c_int),
                ("real_time_correlative_scan_matcher", RtcsmOptions),
                ("ceres_scan_matcher", CsmOptions),
                ("motion_filter_max_time_seconds", C.c_double),
                ("motion_filter_max_distance_meters", C.c_double),
                ("motion_filter_max_angle_radians", C.c_double),
                ("high_resolution", C.c_double), ("high_resolution_max_range", C.c_double),
                ("low_resolution", C.c_double), ("num_range_data", C.c_int),
                ("hit_probability", C.c_double), ("miss_probability", C.c_double),
                ("num_free_space_voxels", C.c_int)]


class FastCsmOptions(C.Structure):
    _fields_ = [("branch_and_bound_depth", C.c_int), ("full_resolution_depth", C.c_int),
                ("min_rotational_score", C.c_double), ("min_low_resolution_score", C.c_double),
                ("linear_xy_search_window", C.c_double), ("linear_z_search_window", C.c_double),
                ("angular_search_window", C.c_double)]


class FastCsmNodeData(C.Structure):
    _fields_ = [("gravity_alignment", C.c_double * 4), ("high_resolution_points", C.POINTER(C.c_float)),
                ("num_high_resolution_points", C.c_int64), ("low_resolution_points", C.POINTER(C.c_float)),
                ("num_low_resolution_points", C.c_int64),
                ("rotational_scan_matcher_histogram", C.POINTER(C.c_float))]


class FastCsmResult(C.Structure):
    _fields_ = [("found", C.c_int), ("score", C.c_float), ("pose_estimate", C.c_double * 7),
                ("rotational_score", C.c_float), ("low_resolution_score", C.c_float), ("num_discrete_scans", C.c_int),
                ("num_scored_candidates", C.c_int64), ("num_score_launches", C.c_int64)]


class ImuNoise(C.Structure):
    _fields_ = [("acc_n", C.c_double), ("gyr_n", C.c_double), ("acc_w", C.c_double), ("gyr_w", C.c_double)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_uint64), C.c_void_p)


class ImuWindowOptions(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "acc_noise", "gyr_noise", "acc_bias_noise", "gyr_bias_noise", "gravity", "integration_sigma", "prior_pose_noise",
        "prior_velocity_sigma", "prior_bias_sigma", "ceres_pose_noise_t", "ceres_pose_noise_r", "ceres_pose_noise_t_drift",
        "ceres_pose_noise_r_drift", "prior_gravity_noise")] + [
            ("window_size", C.c_int), ("iterations", C.c_int), ("enable_gravity_factor", C.c_int),
            ("frames_for_online_gravity_estimate", C.c_int), ("lidar_in_imu_translation", C.c_double * 3),
            ("graph_reset_every", C.c_int), ("tangent_preintegration", C.c_int), ("relinearize_threshold", C.c_double)]


class ImuPreintegration(C.Structure):
    _fields_ = [("sum_dt", C.c_double), ("delta_p", C.c_double * 3), ("delta_q", C.c_double * 4),
                ("delta_v", C.c_double * 3), ("linearized_ba", C.c_double * 3), ("linearized_bg", C.c_double * 3),
                ("jacobian", C.c_double * 225), ("covariance", C.c_double * 225)]


class MatchResult(C.Structure):
    _fields_ = [("dropped", C.c_int), ("pose_estimate", C.c_double * 7),
                ("pose_observation_in_submap", C.c_double * 7), ("initial_ceres_pose", C.c_double * 7),
                ("rtcsm_score", C.c_float), ("summary", CsmSummary), ("residual_distance", C.c_double),
                ("residual_angle", C.c_double), ("num_high_resolution_points", C.c_int64),
                ("num_low_resolution_points", C.c_int64), ("matching_submap_index", C.c_int)]


class MemoryStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("grids", "leaf_table_bytes", "leaf_pool_bytes", "mirror_bytes", "mirror_budget_bytes",
                                         "mirrors_refused", "scratch_bytes", "leaf_capacity", "leaf_slots_upper_bound")] + [
                                             ("mirror_windowed", C.c_int)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class InsertionResult(C.Structure):
    _fields_ = [("inserted", C.c_int), ("num_insertion_submaps", C.c_int), ("insertion_submap_index", C.c_int * 2),
                ("submap_added", C.c_int), ("submap_finished", C.c_int)]


# Every symbol include/dliom.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("dliom_status_string", C.c_char_p, [C.c_int]),
    ("dliom_last_error", C.c_char_p, []),
    ("dliom_device_count", C.c_int, []),
    ("dliom_ctx_create", C.c_int, [C.c_int, C.POINTER(_vp)]),
    ("dliom_ctx_create_on_stream", C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    ("dliom_ctx_destroy", C.c_int, [_vp]),
    ("dliom_ctx_synchronize", C.c_int, [_vp]),
    ("dliom_compute_lookup_table_to_apply_odds", C.c_int, [C.c_float, _u16p]),
    ("dliom_odds", C.c_float, [C.c_float]),
    ("dliom_probability_to_value", C.c_uint16, [C.c_float]),
    ("dliom_grid_set_values", C.c_int, [_vp, _i32p, _u16p, C.c_int64]),
    ("dliom_value_to_probability_table", C.c_int, [_f32p]),
    ("dliom_grid_create", C.c_int, [_vp, C.c_float, C.POINTER(_vp)]),
    ("dliom_grid_destroy", C.c_int, [_vp]),
    ("dliom_grid_resolution", C.c_int, [_vp, _f32p]),
    ("dliom_grid_bits", C.c_int, [_vp, C.POINTER(C.c_int)]),
    ("dliom_grid_mirror_stats", C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    ("dliom_ctx_memory_stats", C.c_int, [_vp, C.POINTER(MemoryStats)]),
    ("dliom_ctx_set_mirror_budget", C.c_int, [_vp, C.c_int64]),
    ("dliom_grid_memory_stats", C.c_int, [_vp, C.POINTER(MemoryStats)]),
    ("dliom_imu_window_solver_stats", C.c_int, [_vp, _i64p, _i64p]),
    ("dliom_imu_window_window_optimize", C.c_int, [_vp, _f64p, C.c_int, _f64p, _f64p, _f64p]),
    ("dliom_grid_upload_blocks", C.c_int, [_vp, _i32p, _u16p, C.c_int64]),
    ("dliom_grid_num_blocks", C.c_int, [_vp, _i64p]),
    ("dliom_grid_download_blocks", C.c_int, [_vp, _i32p, _u16p, C.c_int64, _i64p]),
    ("dliom_grid_get_values", C.c_int, [_vp, _i32p, C.c_int64, _u16p]),
    ("dliom_grid_to_proto", C.c_int, [_vp, C.POINTER(C.c_uint8), C.c_int64, _i64p]),
    ("dliom_grid_from_proto", C.c_int, [_vp, C.POINTER(C.c_uint8), C.c_int64, C.POINTER(_vp)]),
    ("dliom_submap3d_to_proto", C.c_int, [_f64p, C.c_int32, C.c_int, C.POINTER(C.c_uint8), C.c_int64, C.POINTER(C.c_uint8),
                                          C.c_int64, C.c_int, C.POINTER(C.c_uint8), C.c_int64, _i64p]),
    ("dliom_submap3d_from_proto", C.c_int, [C.POINTER(C.c_uint8), C.c_int64, C.c_int, _f64p, C.POINTER(C.c_int32),
                                            C.POINTER(C.c_int), _i64p, _i64p, _i64p, _i64p]),
    ("dliom_grid_insert", C.c_int, [_vp, _f32p, _f32p, C.c_int64, _u16p, _u16p, C.c_int]),
    ("dliom_inserter_create", C.c_int, [_vp, C.c_double, C.c_double, C.c_int, C.POINTER(_vp)]),
    ("dliom_inserter_destroy", C.c_int, [_vp]),
    ("dliom_inserter_tables", C.c_int, [_vp, _u16p, _u16p]),
    ("dliom_inserter_insert", C.c_int, [_vp, _vp, _f32p, _f32p, C.c_int64]),
    ("dliom_inserter_insert_cloud", C.c_int, [_vp, _vp, _f32p, C.c_int, _f32p, _vp, C.c_float]),
    ("dliom_inserter_insert_cloud_multi", C.c_int, [_vp, C.c_int, C.POINTER(_vp), _f32p, C.POINTER(C.c_int), _f32p, _vp, _f32p]),
    ("dliom_cloud_create", C.c_int, [_vp, _f32p, C.c_int64, C.POINTER(_vp)]),
    ("dliom_cloud_destroy", C.c_int, [_vp]),
    ("dliom_cloud_size", C.c_int, [_vp, _i64p]),
    ("dliom_cloud_voxel_filter", C.c_int, [_vp, _vp, C.c_float, C.POINTER(_vp)]),
    ("dliom_cloud_adaptive_voxel_filter", C.c_int, [_vp, _vp, C.POINTER(AdaptiveVoxelFilterOptions), C.POINTER(_vp)]),
    ("dliom_cloud_adaptive_voxel_filter_pair", C.c_int, [_vp, _vp, C.POINTER(AdaptiveVoxelFilterOptions),
                                                         C.POINTER(AdaptiveVoxelFilterOptions), C.POINTER(_vp), C.POINTER(_vp)]),
    ("dliom_cloud_download", C.c_int, [_vp, _f32p]),
    ("dliom_cloud_download_transformed", C.c_int, [_vp, _f32p, _f32p]),
    ("dliom_rtcsm3d_match", C.c_int, [_vp, C.POINTER(RtcsmOptions), _f64p, _f32p, C.c_int64, _vp, _f64p, _f32p]),
    ("dliom_rtcsm3d_match_cloud", C.c_int, [_vp, C.POINTER(RtcsmOptions), _f64p, _vp, _vp, _f64p, _f32p]),
    ("dliom_rtcsm3d_shard_begin", C.c_int, [_vp, C.POINTER(RtcsmOptions), _f64p, _vp, _vp, C.c_int, C.c_int,
                                            C.POINTER(C.c_uint32)]),
    ("dliom_rtcsm3d_shard_finish", C.c_int, [_vp, C.c_uint32, C.POINTER(C.c_uint64)]),
    ("dliom_rtcsm3d_shard_decode", C.c_int, [_vp, C.c_uint64, _f64p, _f32p]),
    ("dliom_rtcsm3d_window", C.c_int, [C.POINTER(RtcsmOptions), C.c_float, _f32p, C.c_int64, C.POINTER(RtcsmWindow)]),
    ("dliom_rtcsm3d_last_stats", C.c_int, [_vp, C.POINTER(RtcsmStats)]),
    ("dliom_rtcsm3d_box_error", C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    ("dliom_rtcsm3d_match_sharded", C.c_int, [_vp, C.POINTER(RtcsmOptions), _f64p, _vp, _vp, C.c_int, C.c_int, EXCHANGE_FN,
                                              _vp, _f64p, C.POINTER(C.c_float)]),
    ("dliom_rtcsm3d_match_sharded_rccl", C.c_int, [_vp, C.POINTER(RtcsmOptions), _f64p, _vp, _vp, _vp, _f64p,
                                                   C.POINTER(C.c_float)]),
    ("dliom_csm3d_match", C.c_int, [_vp, C.POINTER(CsmOptions), _f64p, _f64p, C.c_int, C.POINTER(_f32p), _i64p,
                                    C.POINTER(_vp), _f64p, C.POINTER(CsmSummary)]),
    ("dliom_csm3d_match_cloud", C.c_int, [_vp, C.POINTER(CsmOptions), _f64p, _f64p, C.c_int, C.POINTER(_vp),
                                          C.POINTER(_vp), _f64p, C.POINTER(CsmSummary)]),
    ("dliom_front_end_create", C.c_int, [_vp, C.POINTER(FrontEndOptions), C.POINTER(_vp)]),
    ("dliom_front_end_destroy", C.c_int, [_vp]),
    ("dliom_front_end_match", C.c_int, [_vp, _f64p, _f32p, _f32p, C.c_int64, C.POINTER(MatchResult)]),
    ("dliom_front_end_match_cloud", C.c_int, [_vp, _f64p, _f32p, _vp, C.POINTER(MatchResult)]),
    ("dliom_front_end_insert", C.c_int, [_vp, C.c_int64, _f64p, _f64p, C.POINTER(InsertionResult)]),
    ("dliom_front_end_num_active_submaps", C.c_int, [_vp, C.POINTER(C.c_int)]),
    ("dliom_front_end_insert_range_data", C.c_int, [_vp, _f32p, _vp, _f64p, C.POINTER(InsertionResult)]),
    ("dliom_front_end_num_finished_submaps", C.c_int, [_vp, C.POINTER(C.c_int)]),
    ("dliom_front_end_take_finished_submap", C.c_int, [_vp, _f64p, C.POINTER(C.c_int), C.POINTER(_vp), C.POINTER(_vp)]),
    ("dliom_front_end_matching_index", C.c_int, [_vp, C.POINTER(C.c_int)]),
    ("dliom_front_end_matched_clouds", C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    ("dliom_front_end_active_submap", C.c_int, [_vp, C.c_int, _f64p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                                C.POINTER(_vp), C.POINTER(_vp)]),
    ("dliom_voxel_filter", C.c_int, [C.c_float, _f32p, C.c_int64, _f32p, _i64p]),
    ("dliom_adaptive_voxel_filter", C.c_int, [C.POINTER(AdaptiveVoxelFilterOptions), _f32p, C.c_int64, _f32p, _i64p]),
    ("dliom_deskew", C.c_int, [_vp, _f64p, _f64p, C.c_double, _f32p, C.c_int64, _f32p, C.c_float, C.c_float, _f32p,
                               C.POINTER(C.c_uint8), _f32p]),
    ("dliom_add_range_data", C.c_int, [_vp, _f64p, _f64p, C.c_double, _f32p, C.c_int64, _f32p, C.c_float, C.c_float,
                                       C.c_float, C.POINTER(_vp), _f32p, _f32p]),
    ("dliom_fast_csm_create", C.c_int, [_vp, _vp, _vp, _f32p, _f32p, C.c_int, C.c_int, C.POINTER(FastCsmOptions),
                                        C.POINTER(_vp)]),
    ("dliom_fast_csm_destroy", C.c_int, [_vp]),
    ("dliom_ctx_device", C.c_int, [_vp]),
    ("dliom_fast_csm_match", C.c_int, [_vp, _vp, _f64p, _f64p, C.POINTER(FastCsmNodeData), C.c_float,
                                       C.POINTER(FastCsmResult)]),
    ("dliom_fast_csm_match_full_submap", C.c_int, [_vp, _vp, _f64p, _f64p, C.POINTER(FastCsmNodeData), C.c_float,
                                                   C.POINTER(FastCsmResult)]),
    ("dliom_fast_csm_match_with_3dof_initial", C.c_int, [_vp, _vp, _f64p, C.POINTER(FastCsmNodeData), C.c_float,
                                                         C.POINTER(FastCsmResult)]),
    ("dliom_fast_csm_level", C.c_int, [_vp, C.c_int, _i32p, _i32p, C.POINTER(C.c_uint8), C.c_int64]),
    ("dliom_range_accumulator_create", C.c_int, [_vp, C.POINTER(_vp)]),
    ("dliom_range_accumulator_destroy", C.c_int, [_vp]),
    ("dliom_range_accumulator_add", C.c_int, [_vp, _f64p, _f64p, C.c_double, _f32p, _f32p, C.c_int64, _f32p, C.c_int,
                                              C.c_float, C.c_float, C.c_float, _f32p, C.POINTER(C.c_int)]),
    ("dliom_range_accumulator_finish", C.c_int, [_vp, C.c_float, C.POINTER(_vp), _f32p]),
    ("dliom_imu_window_default_options", C.c_int, [C.POINTER(ImuWindowOptions)]),
    ("dliom_imu_window_create", C.c_int, [C.POINTER(ImuWindowOptions), C.POINTER(_vp)]),
    ("dliom_imu_window_destroy", C.c_int, [_vp]),
    ("dliom_imu_window_initialize", C.c_int, [_vp, _f64p, _f64p, _f64p]),
    ("dliom_imu_window_add_imu", C.c_int, [_vp, _f64p, _f64p, C.c_double]),
    ("dliom_imu_window_predict", C.c_int, [_vp, _f64p, _f64p]),
    ("dliom_imu_window_add_gravity", C.c_int, [_vp, C.c_int, _f64p]),
    ("dliom_imu_window_add_pose", C.c_int, [_vp, _f64p, C.c_int, _f64p, _f64p, _f64p]),
    ("dliom_imu_window_state", C.c_int, [_vp, C.c_int, _f64p, _f64p, _f64p]),
    ("dliom_imu_window_size", C.c_int, [_vp]),
    ("dliom_diag_imu_factor_jacobians", C.c_int, [_vp, _f64p, _f64p]),
    ("dliom_imu_window_add_imu_batch", C.c_int, [_vp, C.c_int, _f64p, _f64p, _f64p]),
    ("dliom_imu_window_gravity_estimate", C.c_int, [_vp, _f64p, C.POINTER(C.c_int), _i64p]),
    ("dliom_gravity_estimate", C.c_int, [C.c_int, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, C.c_double, _f64p, C.POINTER(C.c_int)]),
    ("dliom_imu_integrator_create", C.c_int, [_f64p, _f64p, C.POINTER(ImuNoise), C.POINTER(_vp)]),
    ("dliom_imu_integrator_destroy", C.c_int, [_vp]),
    ("dliom_imu_integrator_reset", C.c_int, [_vp, _f64p, _f64p, C.POINTER(ImuNoise)]),
    ("dliom_imu_integrator_push_back", C.c_int, [_vp, C.c_double, _f64p, _f64p]),
    ("dliom_imu_integrator_repropagate", C.c_int, [_vp, _f64p, _f64p]),
    ("dliom_imu_integrator_get", C.c_int, [_vp, C.POINTER(ImuPreintegration)]),
    ("dliom_imu_integrator_evaluate", C.c_int, [_vp, _f64p, _f64p, _f64p, _f64p]),
    ("dliom_imu_integrator_predict", C.c_int, [_vp, _f64p, _f64p, _f64p]),
    ("dliom_rotational_histogram", C.c_int, [_f32p, C.c_int64, C.c_int, _f32p]),
    ("dliom_rotational_histogram_mt", C.c_int, [_f32p, C.c_int64, C.c_int, C.c_int, _f32p]),
    ("dliom_cloud_rotational_histogram", C.c_int, [_vp, _vp, _f32p, C.c_int, _f32p]),
    ("dliom_cloud_rotational_histogram_begin", C.c_int, [_vp, _vp, _f32p, C.c_int]),
    ("dliom_cloud_rotational_histogram_finish", C.c_int, [_vp, _f32p]),
    ("dliom_diag_std_sort_order", C.c_int, [_vp, _f32p, C.c_int, C.POINTER(C.c_int32)]),
    ("dliom_diag_sequential_sums", C.c_int, [_vp, _f32p, C.c_int, C.c_int, _f32p, _f32p]),
    ("dliom_diag_histogram_contributions", C.c_int, [_vp, _vp, _f32p, C.c_int, C.POINTER(C.c_int32), _f32p, C.c_int64,
                                                      C.POINTER(C.c_int64)]),
    ("dliom_rotational_scan_match", C.c_int, [_f32p, _f32p, C.c_int, C.c_int, _f32p, C.c_float, _f32p, C.c_int, _f32p]),
    ("dliom_rtcsm2d_match", C.c_int, [C.POINTER(RtcsmOptions), _f64p, _f32p, C.c_int64, _u16p, C.c_int, C.c_int,
                                      C.c_double, C.c_double, C.c_double, _f64p, _f64p]),
    ("dliom_probe_transform_cell_indices", C.c_int, [_vp, _f32p, _f32p, C.c_int64, C.c_float, _i32p]),
    ("dliom_rtcsm3d_score_volume", C.c_int, [_vp, C.POINTER(RtcsmOptions), _f64p, _f32p, C.c_int64, _vp, _u64p,
                                             C.c_int64, _i64p]),
    ("dliom_rtcsm3d_sequential_sums", C.c_int, [_vp, C.POINTER(RtcsmOptions), _f64p, _f32p, C.c_int64, _vp, _i64p,
                                                C.c_int64, C.c_int, _f32p]),
    ("dliom_csm3d_evaluate", C.c_int, [_vp, C.POINTER(CsmOptions), _f64p, _f64p, _f64p, C.c_int, C.POINTER(_f32p),
                                       _i64p, C.POINTER(_vp), _f64p, _f64p, _f64p]),
    ("dliom_ctx_set_tuning", C.c_int, [_vp, C.c_int, C.c_int]),
    ("dliom_ctx_poll_fallbacks", C.c_int, [_vp, C.POINTER(C.c_int64)]),
    ("dliom_ctx_read_backs", C.c_int, [_vp, C.POINTER(C.c_int64)]),
    ("dliom_ctx_voxel_filter_reruns", C.c_int, [_vp, C.POINTER(C.c_int64)]),
    ("dliom_host_register", C.c_int, [_vp, _vp, C.c_size_t]),
    ("dliom_host_unregister", C.c_int, [_vp, _vp]),
    ("dliom_ctx_get_tuning", C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    ("dliom_deskew_check_stats", C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("dliom_ctx_set_profiling", C.c_int, [_vp, C.c_int]),
    ("dliom_ctx_reset_profiling", C.c_int, [_vp]),
    ("dliom_ctx_kernel_time", C.c_int, [_vp, C.c_int, _f64p, _i64p]),
]

_lib = None


def load_library(path=None):
    """Loads libdliom.so and binds every symbol of include/dliom.h (AttributeError if one is
    missing).  Loading needs the HIP runtime but no GPU."""
    global _lib
    if _lib is None:
        p = path or os.environ.get("DLIOM_LIB") or LIB_PATH  # DLIOM_LIB: A/B builds of the library (tools/)
        if not os.path.exists(p):
            raise DliomError(ERR_NO_DEVICE, "libdliom.so not built at %s (run __graft_entry__.build())" % p)
        lib = C.CDLL(p)
        for name, res, args in SYMBOLS:
            f = getattr(lib, name)
            f.restype = res
            f.argtypes = args
        _lib = lib
    return _lib


def _check(status, where):
    if status != OK:
        raise DliomError(status, where)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t):
    return a.ctypes.data_as(t)


def device_count():
    return load_library().dliom_device_count()


def compute_lookup_table_to_apply_odds(odds):
    """mapping/probability_values.cc:73-83"""
    out = np.zeros(32768, dtype=np.uint16)
    _check(load_library().dliom_compute_lookup_table_to_apply_odds(C.c_float(odds), _p(out, _u16p)), "lookup table")
    return out


def odds(p):
    return load_library().dliom_odds(C.c_float(p))


def value_to_probability_table():
    out = np.zeros(65536, dtype=np.float32)
    _check(load_library().dliom_value_to_probability_table(_p(out, _f32p)), "value table")
    return out


class Context:
    """One HIP stream + scratch memory (dliom_ctx)."""

    def __init__(self, device_id=0, stream=None):
        L = load_library()
        self._L = L
        h = _vp()
        if stream is None:
            _check(L.dliom_ctx_create(device_id, C.byref(h)), "dliom_ctx_create")
        else:
            _check(L.dliom_ctx_create_on_stream(device_id, _vp(stream), C.byref(h)), "dliom_ctx_create_on_stream")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self._L.dliom_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(self._L.dliom_ctx_synchronize(self.h), "synchronize")

    def voxel_filter_reruns(self):
        """dliom_ctx_voxel_filter_reruns: voxel filter launches repeated with 21-bit keys (a point beyond 4095 edges)."""
        n = C.c_int64(0)
        _check(self._L.dliom_ctx_voxel_filter_reruns(self.h, C.byref(n)), "voxel_filter_reruns")
        return int(n.value)

    def memory_stats(self):
        """dliom_ctx_memory_stats: HBM held by this context's grids and scratch buffers; no sync."""
        m = MemoryStats()
        _check(self._L.dliom_ctx_memory_stats(self.h, C.byref(m)), "ctx_memory_stats")
        return m.as_dict()

    def set_mirror_budget(self, num_bytes):
        """dliom_ctx_set_mirror_budget: cap on the sum of this context's dense mirrors (0 = none)."""
        _check(self._L.dliom_ctx_set_mirror_budget(self.h, int(num_bytes)), "ctx_set_mirror_budget")

    def host_register(self, array):
        """dliom_host_register: page-locks a numpy array the caller keeps (uploads from it become one asynchronous DMA)."""
        _check(self._L.dliom_host_register(self.h, C.c_void_p(array.ctypes.data), C.c_size_t(array.nbytes)), "host_register")

    def host_unregister(self, array):
        _check(self._L.dliom_host_unregister(self.h, C.c_void_p(array.ctypes.data)), "host_unregister")

    def read_backs(self):
        """dliom_ctx_read_backs: polled host round trips on this context so far."""
        n = C.c_int64(0)
        _check(self._L.dliom_ctx_read_backs(self.h, C.byref(n)), "read_backs")
        return int(n.value)

    def poll_fallbacks(self):
        """dliom_ctx_poll_fallbacks: read-backs that ran out of polling time and synchronised the stream instead."""
        n = C.c_int64(0)
        _check(self._L.dliom_ctx_poll_fallbacks(self.h, C.byref(n)), "poll_fallbacks")
        return int(n.value)

    def set_tuning(self, knob, value):
        """dliom_ctx_set_tuning: TUNE_SCORE_KERNEL, TUNE_CSM_ONE_LAUNCH_MAX, TUNE_CSM_GRID_SYNC (knob 2 is reserved: refused by
        the shipped library, the fault injection of libdliom_hooks.so)."""
        _check(self._L.dliom_ctx_set_tuning(self.h, int(knob), int(value)), "set_tuning")

    def deskew_check_stats(self):
        """(records checked against glibc on the host, ring overflows, hits whose device cast differed and were redone)."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _check(self._L.dliom_deskew_check_stats(self.h, C.byref(a), C.byref(b), C.byref(c)), "deskew_check_stats")
        return int(a.value), int(b.value), int(c.value)

    def get_tuning(self, knob):
        v = C.c_int(0)
        _check(self._L.dliom_ctx_get_tuning(self.h, int(knob), C.byref(v)), "get_tuning")
        return v.value

    def set_profiling(self, on):
        _check(self._L.dliom_ctx_set_profiling(self.h, int(on)), "set_profiling")

    def reset_profiling(self):
        _check(self._L.dliom_ctx_reset_profiling(self.h), "reset_profiling")

    def kernel_time(self, kernel_id):
        ms, n = C.c_double(), C.c_int64()
        _check(self._L.dliom_ctx_kernel_time(self.h, kernel_id, C.byref(ms), C.byref(n)), "kernel_time")
        return ms.value, n.value

    def probe_transform_cell_indices(self, pose7_f32, points, resolution):
        pose = _f32(pose7_f32)
        pts = _f32(points).reshape(-1, 3)
        out = np.zeros((len(pts), 3), dtype=np.int32)
        _check(self._L.dliom_probe_transform_cell_indices(self.h, _p(pose, _f32p), _p(pts, _f32p), len(pts),
                                                          C.c_float(resolution), _p(out, _i32p)), "probe cells")
        return out


class PointCloud:
    """sensor::PointCloud staged in HBM (dliom_cloud)."""

    def __init__(self, ctx, points=None, _handle=None):
        self._L = ctx._L
        self.ctx = ctx
        if _handle is not None:  # a cloud built on the device (filters)
            self.h = _handle
            n = C.c_int64()
            _check(self._L.dliom_cloud_size(self.h, C.byref(n)), "dliom_cloud_size")
            self.n = n.value
            return
        pts = _f32(points).reshape(-1, 3)
        self.n = len(pts)
        h = _vp()
        _check(self._L.dliom_cloud_create(ctx.h, _p(pts, _f32p), len(pts), C.byref(h)), "dliom_cloud_create")
        self.h = h

    def __len__(self):
        return self.n

    def voxel_filter(self, size):
        """sensor::VoxelFilter(size).Filter(cloud) on the device -> new PointCloud."""
        h = _vp()
        _check(self._L.dliom_cloud_voxel_filter(self.ctx.h, self.h, C.c_float(size), C.byref(h)), "dliom_cloud_voxel_filter")
        return PointCloud(self.ctx, _handle=h)

    def adaptive_voxel_filter(self, max_length, min_num_points, max_range):
        """sensor::AdaptiveVoxelFilter(options).Filter(cloud) on the device -> new PointCloud."""
        o = AdaptiveVoxelFilterOptions(max_length, min_num_points, max_range)
        h = _vp()
        _check(self._L.dliom_cloud_adaptive_voxel_filter(self.ctx.h, self.h, C.byref(o), C.byref(h)),
               "dliom_cloud_adaptive_voxel_filter")
        return PointCloud(self.ctx, _handle=h)

    def adaptive_voxel_filter_pair(self, first, second):
        """Two AdaptiveVoxelFilter option triples (max_length, min_num_points, max_range) searched together
        -> (PointCloud, PointCloud), each equal to adaptive_voxel_filter(*triple)."""
        a, b = AdaptiveVoxelFilterOptions(*first), AdaptiveVoxelFilterOptions(*second)
        ha, hb = _vp(), _vp()
        _check(self._L.dliom_cloud_adaptive_voxel_filter_pair(self.ctx.h, self.h, C.byref(a), C.byref(b), C.byref(ha),
                                                              C.byref(hb)), "dliom_cloud_adaptive_voxel_filter_pair")
        return PointCloud(self.ctx, _handle=ha), PointCloud(self.ctx, _handle=hb)

    def download(self, pose7=None):
        """The points, packed xyz; pose7 (float [t, q]): sensor::TransformPointCloud(cloud, pose) applied on the device."""
        out = np.zeros((self.n, 3), dtype=np.float32)
        if pose7 is None:
            _check(self._L.dliom_cloud_download(self.h, _p(out, _f32p)), "dliom_cloud_download")
        else:
            _check(self._L.dliom_cloud_download_transformed(self.h, _p(_f32(pose7), _f32p), _p(out, _f32p)),
                   "dliom_cloud_download_transformed")
        return out

    def close(self):
        if getattr(self, "h", None):
            self._L.dliom_cloud_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HybridGrid:
    """Device-resident mapping::HybridGrid."""

    def __init__(self, ctx, resolution):
        self._L = ctx._L
        self.ctx = ctx
        self.resolution = float(np.float32(resolution))
        h = _vp()
        _check(self._L.dliom_grid_create(ctx.h, C.c_float(resolution), C.byref(h)), "dliom_grid_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self._L.dliom_grid_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def bits(self):
        b = C.c_int()
        _check(self._L.dliom_grid_bits(self.h, C.byref(b)), "grid_bits")
        return b.value

    def memory_stats(self):
        """dliom_grid_memory_stats: this grid's HBM (leaf table, pool, mirror), capacity and slot upper bound; no sync."""
        m = MemoryStats()
        _check(self._L.dliom_grid_memory_stats(self.h, C.byref(m)), "grid_memory_stats")
        return m.as_dict()

    def mirror_stats(self):
        """(rebuilds, bytes, windowed) of the correlative matcher's dense mirror of this grid."""
        r, b, w = C.c_int64(), C.c_int64(), C.c_int()
        _check(self._L.dliom_grid_mirror_stats(self.h, C.byref(r), C.byref(b), C.byref(w)), "grid_mirror_stats")
        return int(r.value), int(b.value), bool(w.value)

    def upload_blocks(self, origins, values512):
        origins = np.ascontiguousarray(origins, dtype=np.int32).reshape(-1, 3)
        values512 = np.ascontiguousarray(values512, dtype=np.uint16).reshape(-1, 512)
        assert len(origins) == len(values512)
        _check(self._L.dliom_grid_upload_blocks(self.h, _p(origins, _i32p), _p(values512, _u16p), len(origins)),
               "grid_upload_blocks")

    def num_blocks(self):
        n = C.c_int64()
        _check(self._L.dliom_grid_num_blocks(self.h, C.byref(n)), "grid_num_blocks")
        return n.value

    def download_blocks(self):
        n = self.num_blocks()
        origins = np.zeros((n, 3), dtype=np.int32)
        values = np.zeros((n, 512), dtype=np.uint16)
        got = C.c_int64()
        _check(self._L.dliom_grid_download_blocks(self.h, _p(origins, _i32p), _p(values, _u16p), n, C.byref(got)),
               "grid_download_blocks")
        return origins[:got.value], values[:got.value]

    def cells(self):
        """Non-zero cells as a dict {(x,y,z): value} (what HybridGrid's iterator yields)."""
        origins, values = self.download_blocks()
        out = {}
        for o, v in zip(origins, values):
            nz = np.nonzero(v)[0]
            for c in nz:
                out[(int(o[0]) + (c & 7), int(o[1]) + ((c >> 3) & 7), int(o[2]) + (c >> 6))] = int(v[c])
        return out

    def to_proto(self):
        """Serialized mapping::proto::HybridGrid (bytes)."""
        n = C.c_int64()
        _check(self._L.dliom_grid_to_proto(self.h, None, 0, C.byref(n)), "dliom_grid_to_proto")
        buf = (C.c_uint8 * max(n.value, 1))()
        _check(self._L.dliom_grid_to_proto(self.h, buf, n.value, C.byref(n)), "dliom_grid_to_proto")
        return bytes(buf[:n.value])

    @classmethod
    def from_proto(cls, ctx, data):
        buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data if len(data) else b"\0")
        h = _vp()
        _check(ctx._L.dliom_grid_from_proto(ctx.h, buf, len(data), C.byref(h)), "dliom_grid_from_proto")
        g = cls.__new__(cls)
        g._L = ctx._L
        g.ctx = ctx
        g.h = h
        res = C.c_float()
        _check(ctx._L.dliom_grid_resolution(h, C.byref(res)), "dliom_grid_resolution")
        g.resolution = res.value
        return g

    def set_values(self, cells_xyz, values):
        cells = np.ascontiguousarray(cells_xyz, dtype=np.int32).reshape(-1, 3)
        vals = np.ascontiguousarray(values, dtype=np.uint16)
        _check(self._L.dliom_grid_set_values(self.h, _p(cells, _i32p), _p(vals, _u16p), len(vals)), "grid_set_values")

    def SetProbability(self, index, probability):
        """hybrid_grid.h:489-491"""
        self.set_values([index], [self._L.dliom_probability_to_value(C.c_float(probability))])

    def values(self, cells_xyz):
        cells = np.ascontiguousarray(cells_xyz, dtype=np.int32).reshape(-1, 3)
        out = np.zeros(len(cells), dtype=np.uint16)
        _check(self._L.dliom_grid_get_values(self.h, _p(cells, _i32p), len(cells), _p(out, _u16p)), "grid_get_values")
        return out


class RangeDataInserter3D:
    """mapping::RangeDataInserter3D: options -> two odds tables; Insert(range_data, grid).

    With a Context the tables live in HBM (dliom_inserter); without one the host tables are
    passed on every Insert (dliom_grid_insert)."""

    def __init__(self, hit_probability, miss_probability, num_free_space_voxels, ctx=None):
        if not hit_probability > 0.5 or not miss_probability < 0.5:
            raise ValueError("CHECK_GT(hit, 0.5) / CHECK_LT(miss, 0.5) (range_data_inserter_3d.cc:64-65)")
        self.num_free_space_voxels = int(num_free_space_voxels)
        self.h = None
        if ctx is not None:
            self._L = ctx._L
            h = _vp()
            _check(self._L.dliom_inserter_create(ctx.h, hit_probability, miss_probability,
                                                 self.num_free_space_voxels, C.byref(h)), "dliom_inserter_create")
            self.h = h
            self.hit_table = np.zeros(32768, dtype=np.uint16)
            self.miss_table = np.zeros(32768, dtype=np.uint16)
            _check(self._L.dliom_inserter_tables(h, _p(self.hit_table, _u16p), _p(self.miss_table, _u16p)),
                   "dliom_inserter_tables")
        else:
            # Odds(float(options.hit_probability()))
            self.hit_table = compute_lookup_table_to_apply_odds(odds(np.float32(hit_probability)))
            self.miss_table = compute_lookup_table_to_apply_odds(odds(np.float32(miss_probability)))

    def close(self):
        if getattr(self, "h", None):
            self._L.dliom_inserter_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def Insert(self, origin, returns, grid):
        origin = _f32(origin)
        returns = _f32(returns).reshape(-1, 3)
        if self.h is not None:
            _check(grid._L.dliom_inserter_insert(self.h, grid.h, _p(origin, _f32p), _p(returns, _f32p), len(returns)),
                   "dliom_inserter_insert")
        else:
            _check(grid._L.dliom_grid_insert(grid.h, _p(origin, _f32p), _p(returns, _f32p), len(returns),
                                             _p(self.hit_table, _u16p), _p(self.miss_table, _u16p),
                                             self.num_free_space_voxels), "dliom_grid_insert")

    def InsertCloud(self, grid, cloud, poses=(), origin=(0.0, 0.0, 0.0), max_range=0.0):
        """Submap3D::InsertRangeData's data path on the device: transform the HBM-resident cloud
        through `poses` (0..2 float poses, applied in order), range-filter, insert."""
        assert self.h is not None, "InsertCloud needs an inserter created with a Context"
        poses = _f32(np.asarray(poses, dtype=np.float32).reshape(-1, 7)) if len(poses) else np.zeros((0, 7), np.float32)
        origin = _f32(origin)
        _check(grid._L.dliom_inserter_insert_cloud(self.h, grid.h, _p(poses, _f32p), len(poses), _p(origin, _f32p),
                                                   cloud.h, C.c_float(max_range)), "dliom_inserter_insert_cloud")


def insert_cloud_multi(inserter, cloud, targets, origin=(0.0, 0.0, 0.0)):
    """targets: list of (grid, poses (0..2 float poses), max_range).  One set of launches, one sync."""
    k = len(targets)
    grids = (_vp * k)(*[g.h for g, _, _ in targets])
    poses = np.zeros((k, 14), dtype=np.float32)
    nposes = (C.c_int * k)()
    ranges = np.zeros(k, dtype=np.float32)
    for i, (_, ps, mr) in enumerate(targets):
        ps = np.asarray(ps, dtype=np.float32).reshape(-1, 7) if len(ps) else np.zeros((0, 7), np.float32)
        nposes[i] = len(ps)
        poses[i, :7 * len(ps)] = ps.reshape(-1)
        ranges[i] = mr
    _check(inserter._L.dliom_inserter_insert_cloud_multi(inserter.h, k, grids, _p(poses, _f32p), nposes,
                                                         _p(_f32(origin), _f32p), cloud.h, _p(ranges, _f32p)),
           "dliom_inserter_insert_cloud_multi")


def _rtcsm_opts(o):
    return RtcsmOptions(o["linear_search_window"], o["angular_search_window"],
                        o["translation_delta_cost_weight"], o["rotation_delta_cost_weight"])


class RealTimeCorrelativeScanMatcher3D:
    def __init__(self, ctx, options):
        self.ctx = ctx
        self._L = ctx._L
        self.options = _rtcsm_opts(options)

    def Match(self, initial_pose_estimate, point_cloud, hybrid_grid):
        """Returns (score, pose_estimate[7]).  point_cloud: ndarray or PointCloud."""
        init = _f64(initial_pose_estimate)
        out = np.zeros(7)
        score = C.c_float()
        if isinstance(point_cloud, PointCloud):
            s = self._L.dliom_rtcsm3d_match_cloud(self.ctx.h, C.byref(self.options), _p(init, _f64p), point_cloud.h,
                                                  hybrid_grid.h, _p(out, _f64p), C.byref(score))
        else:
            pts = _f32(point_cloud).reshape(-1, 3)
            s = self._L.dliom_rtcsm3d_match(self.ctx.h, C.byref(self.options), _p(init, _f64p), _p(pts, _f32p),
                                            len(pts), hybrid_grid.h, _p(out, _f64p), C.byref(score))
        _check(s, "dliom_rtcsm3d_match")
        return score.value, out

    def window(self, resolution, point_cloud):
        pts = _f32(point_cloud).reshape(-1, 3)
        w = RtcsmWindow()
        _check(self._L.dliom_rtcsm3d_window(C.byref(self.options), C.c_float(resolution), _p(pts, _f32p), len(pts),
                                            C.byref(w)), "dliom_rtcsm3d_window")
        return w

    def box_error(self):
        f = C.c_uint32(0)
        _check(self._L.dliom_rtcsm3d_box_error(self.ctx.h, C.byref(f)), "box_error")
        return f.value

    def last_stats(self):
        st = RtcsmStats()
        _check(self._L.dliom_rtcsm3d_last_stats(self.ctx.h, C.byref(st)), "last_stats")
        return st

    def sequential_sums(self, initial_pose_estimate, point_cloud, hybrid_grid, candidate_indices, method):
        pts = _f32(point_cloud).reshape(-1, 3)
        idx = np.ascontiguousarray(candidate_indices, dtype=np.int64)
        out = np.zeros(len(idx), dtype=np.float32)
        _check(self._L.dliom_rtcsm3d_sequential_sums(self.ctx.h, C.byref(self.options),
                                                     _p(_f64(initial_pose_estimate), _f64p), _p(pts, _f32p), len(pts),
                                                     hybrid_grid.h, _p(idx, _i64p), len(idx), int(method),
                                                     _p(out, _f32p)), "dliom_rtcsm3d_sequential_sums")
        return out

    def score_volume(self, initial_pose_estimate, point_cloud, hybrid_grid):
        init = _f64(initial_pose_estimate)
        pts = _f32(point_cloud).reshape(-1, 3)
        n = C.c_int64()
        _check(self._L.dliom_rtcsm3d_score_volume(self.ctx.h, C.byref(self.options), _p(init, _f64p), _p(pts, _f32p),
                                                  len(pts), hybrid_grid.h, None, 0, C.byref(n)), "score_volume(size)")
        sums = np.zeros(n.value, dtype=np.uint64)
        _check(self._L.dliom_rtcsm3d_score_volume(self.ctx.h, C.byref(self.options), _p(init, _f64p), _p(pts, _f32p),
                                                  len(pts), hybrid_grid.h, _p(sums, _u64p), n.value, C.byref(n)),
               "score_volume")
        return sums


class RtcsmShard:
    """One rank's share of a sharded RealTimeCorrelativeScanMatcher3D::Match (BASELINE config 4): match() is the
    one-collective protocol of dliom_rtcsm3d_match_sharded (the collective is the caller's function), begin / finish /
    decode the three local phases of the two-collective variant."""

    def match(self, initial_pose_estimate, cloud, hybrid_grid, all_reduce_max):
        """all_reduce_max(int) -> int: MAX over the ranks of one unsigned 64-bit value."""
        def cb(value_ptr, _user):
            try:
                value_ptr[0] = int(all_reduce_max(int(value_ptr[0])))
                return 0
            except Exception:
                return 1
        fn = EXCHANGE_FN(cb)
        out, score = np.zeros(7), C.c_float()
        _check(self._L.dliom_rtcsm3d_match_sharded(self.ctx.h, C.byref(self.options), _p(_f64(initial_pose_estimate), _f64p),
                                                   cloud.h, hybrid_grid.h, self.shard, self.num_shards, fn, None,
                                                   _p(out, _f64p), C.byref(score)), "dliom_rtcsm3d_match_sharded")
        return score.value, out

    def match_rccl(self, initial_pose_estimate, cloud, hybrid_grid, nccl_comm):
        """nccl_comm: an ncclComm_t as an integer / c_void_p; rank and size are the communicator's."""
        out, score = np.zeros(7), C.c_float()
        _check(self._L.dliom_rtcsm3d_match_sharded_rccl(self.ctx.h, C.byref(self.options),
                                                        _p(_f64(initial_pose_estimate), _f64p), cloud.h, hybrid_grid.h,
                                                        C.c_void_p(nccl_comm), _p(out, _f64p), C.byref(score)),
               "dliom_rtcsm3d_match_sharded_rccl")
        return score.value, out

    def __init__(self, ctx, options, shard, num_shards):
        self.ctx = ctx
        self._L = ctx._L
        self.options = _rtcsm_opts(options)
        self.shard, self.num_shards = int(shard), int(num_shards)

    def begin(self, initial_pose_estimate, cloud, hybrid_grid):
        bits = C.c_uint32()
        _check(self._L.dliom_rtcsm3d_shard_begin(self.ctx.h, C.byref(self.options), _p(_f64(initial_pose_estimate), _f64p),
                                                 cloud.h, hybrid_grid.h, self.shard, self.num_shards, C.byref(bits)),
               "dliom_rtcsm3d_shard_begin")
        return bits.value

    def finish(self, global_best_lower_bound_bits):
        packed = C.c_uint64()
        _check(self._L.dliom_rtcsm3d_shard_finish(self.ctx.h, int(global_best_lower_bound_bits), C.byref(packed)),
               "dliom_rtcsm3d_shard_finish")
        return packed.value

    def decode(self, global_best_packed):
        out = np.zeros(7)
        score = C.c_float()
        _check(self._L.dliom_rtcsm3d_shard_decode(self.ctx.h, int(global_best_packed), _p(out, _f64p), C.byref(score)),
               "dliom_rtcsm3d_shard_decode")
        return score.value, out


def _csm_opts(o):
    c = CsmOptions()
    w = list(o["occupied_space_weight"])
    c.num_occupied_space_weights = len(w)
    for i, v in enumerate(w[:MAX_CLOUDS]):
        c.occupied_space_weight[i] = v
    c.translation_weight = o["translation_weight"]
    c.rotation_weight = o["rotation_weight"]
    c.only_optimize_yaw = int(o.get("only_optimize_yaw", False))
    c.use_nonmonotonic_steps = int(o.get("use_nonmonotonic_steps", False))
    c.max_num_iterations = int(o["max_num_iterations"])
    c.num_threads = int(o.get("num_threads", 1))
    return c


class CeresScanMatcher3D:
    def __init__(self, ctx, options):
        self.ctx = ctx
        self._L = ctx._L
        self.options = _csm_opts(options)

    def Match(self, target_translation, initial_pose_estimate, point_clouds_and_hybrid_grids):
        """Returns (pose_estimate[7], summary dict)."""
        k = len(point_clouds_and_hybrid_grids)
        tgt = _f64(target_translation)
        init = _f64(initial_pose_estimate)
        out = np.zeros(7)
        summ = CsmSummary()
        grids = (_vp * k)(*[g.h for _, g in point_clouds_and_hybrid_grids])
        if all(isinstance(c, PointCloud) for c, _ in point_clouds_and_hybrid_grids):
            clouds = (_vp * k)(*[c.h for c, _ in point_clouds_and_hybrid_grids])
            s = self._L.dliom_csm3d_match_cloud(self.ctx.h, C.byref(self.options), _p(tgt, _f64p), _p(init, _f64p), k,
                                                clouds, grids, _p(out, _f64p), C.byref(summ))
        else:
            arrs = [_f32(c).reshape(-1, 3) for c, _ in point_clouds_and_hybrid_grids]
            ptrs = (_f32p * k)(*[_p(a, _f32p) for a in arrs])
            ns = np.array([len(a) for a in arrs], dtype=np.int64)
            s = self._L.dliom_csm3d_match(self.ctx.h, C.byref(self.options), _p(tgt, _f64p), _p(init, _f64p), k, ptrs,
                                          _p(ns, _i64p), grids, _p(out, _f64p), C.byref(summ))
        _check(s, "dliom_csm3d_match")
        return out, {f: getattr(summ, f) for f, _ in CsmSummary._fields_}

    def evaluate(self, target_translation, initial_pose_estimate, pose, point_clouds_and_hybrid_grids):
        k = len(point_clouds_and_hybrid_grids)
        arrs = [_f32(c).reshape(-1, 3) for c, _ in point_clouds_and_hybrid_grids]
        ptrs = (_f32p * k)(*[_p(a, _f32p) for a in arrs])
        ns = np.array([len(a) for a in arrs], dtype=np.int64)
        grids = (_vp * k)(*[g.h for _, g in point_clouds_and_hybrid_grids])
        cost = C.c_double()
        grad = np.zeros(6)
        jtj = np.zeros((6, 6))
        _check(self._L.dliom_csm3d_evaluate(self.ctx.h, C.byref(self.options), _p(_f64(target_translation), _f64p),
                                            _p(_f64(initial_pose_estimate), _f64p), _p(_f64(pose), _f64p), k, ptrs,
                                            _p(ns, _i64p), grids, C.byref(cost), _p(grad, _f64p), _p(jtj, _f64p)),
               "dliom_csm3d_evaluate")
        return cost.value, grad, jtj


def submap3d_to_proto(local_pose7, num_range_data, finished, high_grid_proto=None, low_grid_proto=None, wrap=False):
    """Serialized mapping::proto::Submap3D (or proto::Submap around it) -- Submap3D::ToProto, host only."""
    L = load_library()

    def buf(b):
        return (None, 0) if b is None else ((C.c_uint8 * max(len(b), 1)).from_buffer_copy(b if len(b) else b"\0"), len(b))
    (hp, hn), (lp, ln) = buf(high_grid_proto), buf(low_grid_proto)
    if high_grid_proto is not None and hn == 0:
        hp = (C.c_uint8 * 1)()
    if low_grid_proto is not None and ln == 0:
        lp = (C.c_uint8 * 1)()
    pose = _f64(local_pose7)
    n = C.c_int64()
    args = (_p(pose, _f64p), int(num_range_data), int(bool(finished)), hp, hn, lp, ln, int(bool(wrap)))
    _check(L.dliom_submap3d_to_proto(*args, None, 0, C.byref(n)), "dliom_submap3d_to_proto")
    out = (C.c_uint8 * max(n.value, 1))()
    _check(L.dliom_submap3d_to_proto(*args, out, n.value, C.byref(n)), "dliom_submap3d_to_proto")
    return bytes(out[:n.value])


def submap3d_from_proto(data, wrapped=False):
    """(local_pose7, num_range_data, finished, high grid bytes or None, low grid bytes or None)."""
    L = load_library()
    b = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data if len(data) else b"\0")
    pose = np.zeros(7)
    n, fin = C.c_int32(), C.c_int()
    ho, hs, lo, ls = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    _check(L.dliom_submap3d_from_proto(b, len(data), int(bool(wrapped)), _p(pose, _f64p), C.byref(n), C.byref(fin),
                                       C.byref(ho), C.byref(hs), C.byref(lo), C.byref(ls)), "dliom_submap3d_from_proto")
    hi = None if hs.value < 0 else bytes(data[ho.value:ho.value + hs.value])
    low = None if ls.value < 0 else bytes(data[lo.value:lo.value + ls.value])
    return pose, n.value, bool(fin.value), hi, low


def voxel_filter(size, points):
    """sensor::VoxelFilter(size).Filter(points) (host)."""
    pts = _f32(points).reshape(-1, 3)
    out = np.zeros_like(pts)
    n = C.c_int64()
    _check(load_library().dliom_voxel_filter(C.c_float(size), _p(pts, _f32p), len(pts), _p(out, _f32p), C.byref(n)),
           "dliom_voxel_filter")
    return out[:n.value].copy()


def adaptive_voxel_filter(max_length, min_num_points, max_range, points):
    """sensor::AdaptiveVoxelFilter(options).Filter(points) (host)."""
    pts = _f32(points).reshape(-1, 3)
    out = np.zeros_like(pts)
    n = C.c_int64()
    o = AdaptiveVoxelFilterOptions(max_length, min_num_points, max_range)
    _check(load_library().dliom_adaptive_voxel_filter(C.byref(o), _p(pts, _f32p), len(pts), _p(out, _f32p), C.byref(n)),
           "dliom_adaptive_voxel_filter")
    return out[:n.value].copy()


class _BorrowedGrid(HybridGrid):
    """A grid owned by a front end (never destroyed from Python)."""

    def __init__(self, ctx, handle, resolution):
        self._L = ctx._L
        self.ctx = ctx
        self.resolution = float(np.float32(resolution))
        self.h = handle

    def close(self):
        self.h = None


def front_end_options_struct(options):
    """dliom_front_end_options from the nested dict the tests and tools use."""
    o = FrontEndOptions()
    hi, lo = options["high_resolution_adaptive_voxel_filter"], options["low_resolution_adaptive_voxel_filter"]
    o.high_resolution_adaptive_voxel_filter = AdaptiveVoxelFilterOptions(hi["max_length"], hi["min_num_points"], hi["max_range"])
    o.low_resolution_adaptive_voxel_filter = AdaptiveVoxelFilterOptions(lo["max_length"], lo["min_num_points"], lo["max_range"])
    o.use_online_correlative_scan_matching = int(options["use_online_correlative_scan_matching"])
    o.real_time_correlative_scan_matcher = _rtcsm_opts(options["real_time_correlative_scan_matcher"])
    o.ceres_scan_matcher = _csm_opts(options["ceres_scan_matcher"])
    m, s = options["motion_filter"], options["submaps"]
    o.motion_filter_max_time_seconds = m["max_time_seconds"]
    o.motion_filter_max_distance_meters = m["max_distance_meters"]
    o.motion_filter_max_angle_radians = m["max_angle_radians"]
    o.high_resolution = s["high_resolution"]
    o.high_resolution_max_range = s["high_resolution_max_range"]
    o.low_resolution = s["low_resolution"]
    o.num_range_data = s["num_range_data"]
    o.hit_probability = s["hit_probability"]
    o.miss_probability = s["miss_probability"]
    o.num_free_space_voxels = s["num_free_space_voxels"]
    return o


class LocalTrajectoryBuilder3D:
    """AddAccumulatedRangeData + InsertIntoSubmap of the reference's LocalTrajectoryBuilder3D
    (local_trajectory_builder_3d.cc:493-622), over ActiveSubmaps3D; WindowOptimize sits between match and insert
    (dliom.ImuWindow)."""

    def __init__(self, ctx, options):
        self.ctx = ctx
        self._L = ctx._L
        o = FrontEndOptions()
        hi, lo = options["high_resolution_adaptive_voxel_filter"], options["low_resolution_adaptive_voxel_filter"]
        o.high_resolution_adaptive_voxel_filter = AdaptiveVoxelFilterOptions(hi["max_length"], hi["min_num_points"], hi["max_range"])
        o.low_resolution_adaptive_voxel_filter = AdaptiveVoxelFilterOptions(lo["max_length"], lo["min_num_points"], lo["max_range"])
        o.use_online_correlative_scan_matching = int(options["use_online_correlative_scan_matching"])
        o.real_time_correlative_scan_matcher = _rtcsm_opts(options["real_time_correlative_scan_matcher"])
        o.ceres_scan_matcher = _csm_opts(options["ceres_scan_matcher"])
        m, s = options["motion_filter"], options["submaps"]
        o.motion_filter_max_time_seconds = m["max_time_seconds"]
        o.motion_filter_max_distance_meters = m["max_distance_meters"]
        o.motion_filter_max_angle_radians = m["max_angle_radians"]
        o.high_resolution = s["high_resolution"]
        o.high_resolution_max_range = s["high_resolution_max_range"]
        o.low_resolution = s["low_resolution"]
        o.num_range_data = s["num_range_data"]
        o.hit_probability = s["hit_probability"]
        o.miss_probability = s["miss_probability"]
        o.num_free_space_voxels = s["num_free_space_voxels"]
        self.resolutions = (s["high_resolution"], s["low_resolution"])
        h = _vp()
        _check(self._L.dliom_front_end_create(ctx.h, C.byref(o), C.byref(h)), "dliom_front_end_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self._L.dliom_front_end_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def match(self, pose_prediction, origin, returns):
        returns = _f32(returns).reshape(-1, 3)
        r = MatchResult()
        _check(self._L.dliom_front_end_match(self.h, _p(_f64(pose_prediction), _f64p), _p(_f32(origin), _f32p),
                                             _p(returns, _f32p), len(returns), C.byref(r)), "dliom_front_end_match")
        return self._result(r)

    @staticmethod
    def _result(r):
        return dict(dropped=bool(r.dropped), pose_estimate=np.array(r.pose_estimate),
                    pose_observation_in_submap=np.array(r.pose_observation_in_submap),
                    initial_ceres_pose=np.array(r.initial_ceres_pose), rtcsm_score=r.rtcsm_score,
                    final_cost=r.summary.final_cost, num_iterations=r.summary.num_iterations,
                    num_high=r.num_high_resolution_points, num_low=r.num_low_resolution_points,
                    residual_distance=r.residual_distance, residual_angle=r.residual_angle,
                    matching_submap_index=r.matching_submap_index)

    def match_cloud(self, pose_prediction, origin, cloud):
        """match() on range data that already lives on the device (keep `cloud` alive until insert())."""
        r = MatchResult()
        _check(self._L.dliom_front_end_match_cloud(self.h, _p(_f64(pose_prediction), _f64p), _p(_f32(origin), _f32p),
                                                   cloud.h, C.byref(r)), "dliom_front_end_match_cloud")
        return self._result(r)

    def insert(self, time_ticks, pose_estimate, gravity_alignment):
        r = InsertionResult()
        _check(self._L.dliom_front_end_insert(self.h, int(time_ticks), _p(_f64(pose_estimate), _f64p),
                                              _p(_f64(gravity_alignment), _f64p), C.byref(r)), "dliom_front_end_insert")
        return dict(inserted=bool(r.inserted), num_insertion_submaps=r.num_insertion_submaps,
                    insertion_submap_index=list(r.insertion_submap_index)[:r.num_insertion_submaps],
                    submap_added=bool(r.submap_added), submap_finished=bool(r.submap_finished))

    def num_active_submaps(self):
        n = C.c_int()
        _check(self._L.dliom_front_end_num_active_submaps(self.h, C.byref(n)), "num_active_submaps")
        return n.value

    def matching_index(self):
        n = C.c_int()
        _check(self._L.dliom_front_end_matching_index(self.h, C.byref(n)), "matching_index")
        return n.value

    def num_finished_submaps(self):
        n = C.c_int()
        _check(self._L.dliom_front_end_num_finished_submaps(self.h, C.byref(n)), "num_finished_submaps")
        return n.value

    def take_finished_submap(self):
        """Oldest finished submap; its grids are OWNED by the returned HybridGrid objects (close() frees them)."""
        pose = np.zeros(7)
        n = C.c_int()
        hi, lo = _vp(), _vp()
        _check(self._L.dliom_front_end_take_finished_submap(self.h, _p(pose, _f64p), C.byref(n), C.byref(hi), C.byref(lo)),
               "take_finished_submap")
        grids = []
        for h, res in ((hi, self.resolutions[0]), (lo, self.resolutions[1])):
            g = HybridGrid.__new__(HybridGrid)
            g._L, g.ctx, g.resolution, g.h = self._L, self.ctx, float(np.float32(res)), h
            grids.append(g)
        return dict(local_pose=pose, num_range_data=n.value, hi=grids[0], lo=grids[1])

    def active_submap(self, i):
        pose = np.zeros(7)
        n, fin = C.c_int(), C.c_int()
        hi, lo = _vp(), _vp()
        _check(self._L.dliom_front_end_active_submap(self.h, i, _p(pose, _f64p), C.byref(n), C.byref(fin),
                                                     C.byref(hi), C.byref(lo)), "active_submap")
        return dict(local_pose=pose, num_range_data=n.value, finished=bool(fin.value),
                    hi=_BorrowedGrid(self.ctx, hi, self.resolutions[0]), lo=_BorrowedGrid(self.ctx, lo, self.resolutions[1]))


class RealTimeCorrelativeScanMatcher2D:
    """BASELINE config 1: the 2D matcher over a dense ProbabilityGrid, host only by contract."""

    def __init__(self, options):
        self._L = load_library()
        self.options = _rtcsm_opts(options)

    def Match(self, initial_pose_estimate, point_cloud, cells, resolution, max_xy):
        """cells: uint16 [num_y_cells, num_x_cells] correspondence-cost values.  Returns (score, pose[3])."""
        pts = _f32(point_cloud).reshape(-1, 3)
        cells = np.ascontiguousarray(cells, dtype=np.uint16)
        out = np.zeros(3)
        score = C.c_double()
        _check(self._L.dliom_rtcsm2d_match(C.byref(self.options), _p(_f64(initial_pose_estimate), _f64p),
                                           _p(pts, _f32p), len(pts), _p(cells, _u16p), cells.shape[1], cells.shape[0],
                                           resolution, max_xy[0], max_xy[1], _p(out, _f64p), C.byref(score)),
               "dliom_rtcsm2d_match")
        return score.value, out


def deskew(ctx, prev_pose, predicted_pose, scan_period, hits_xyzt, origin, min_range, max_range):
    """local_trajectory_builder_3d.cc:421-472 on the device.  Returns (xyz[n,3], kind[n], current_pose[7] float32)."""
    h = _f32(hits_xyzt).reshape(-1, 4)
    n = len(h)
    out = np.zeros((n, 3), dtype=np.float32)
    kind = np.zeros(n, dtype=np.uint8)
    cur = np.zeros(7, dtype=np.float32)
    _check(ctx._L.dliom_deskew(ctx.h, _p(_f64(prev_pose), _f64p), _p(_f64(predicted_pose), _f64p), scan_period,
                               _p(h, _f32p), n, _p(_f32(origin), _f32p), C.c_float(min_range), C.c_float(max_range),
                               _p(out, _f32p), kind.ctypes.data_as(C.POINTER(C.c_uint8)), _p(cur, _f32p)), "dliom_deskew")
    return out, kind, cur


def add_range_data_preprocess(ctx, prev_pose, predicted_pose, scan_period, ranges_xyzt, origin, min_range, max_range,
                              voxel_filter_size):
    """AddRangeData's pre-processing chain (:393-487): VoxelFilter(0.5 vfs) [host] -> de-skew + range gate
    [device] -> VoxelFilter(vfs) [host] -> back to the tracking frame by current_pose.inverse() [host,
    float].  Returns (returns_in_tracking, origin_in_tracking, current_pose)."""
    r = _f32(ranges_xyzt).reshape(-1, 4)
    keep = voxel_filter_indices(0.5 * np.float32(voxel_filter_size), r[:, :3])
    hits = r[keep]
    xyz, kind, cur = deskew(ctx, prev_pose, predicted_pose, scan_period, hits, origin, min_range, max_range)
    returns = voxel_filter(voxel_filter_size, xyz[kind == 1])
    inv = _pose_inverse_f32(cur)
    return _transform_f32(inv, returns), _transform_f32(inv, cur[:3].reshape(1, 3))[0], cur


def add_range_data(ctx, prev_pose, predicted_pose, scan_period, ranges_xyzt, origin, min_range, max_range,
                   voxel_filter_size):
    """The same chain entirely on the device (dliom_add_range_data).
    Returns (PointCloud returns_in_tracking, origin_in_tracking, current_pose)."""
    r = _f32(ranges_xyzt).reshape(-1, 4)
    h = _vp()
    o = np.zeros(3, dtype=np.float32)
    cur = np.zeros(7, dtype=np.float32)
    _check(ctx._L.dliom_add_range_data(ctx.h, _p(_f64(prev_pose), _f64p), _p(_f64(predicted_pose), _f64p), scan_period,
                                       _p(r, _f32p), len(r), _p(_f32(origin), _f32p), C.c_float(min_range),
                                       C.c_float(max_range), C.c_float(voxel_filter_size), C.byref(h), _p(o, _f32p),
                                       _p(cur, _f32p)), "dliom_add_range_data")
    return PointCloud(ctx, _handle=h), o, cur


def voxel_filter_indices(size, points):
    """Indices kept by sensor::VoxelFilter (first point per voxel), via the host filter."""
    pts = _f32(points).reshape(-1, 3)
    kept = voxel_filter(size, pts)
    # the filter preserves order and keeps exact copies: recover the indices by a single sweep
    idx = np.zeros(len(kept), dtype=np.int64)
    j = 0
    for i in range(len(pts)):
        if j < len(kept) and pts[i, 0] == kept[j, 0] and pts[i, 1] == kept[j, 1] and pts[i, 2] == kept[j, 2]:
            idx[j] = i
            j += 1
    assert j == len(kept)
    return idx


def _pose_inverse_f32(p):
    """Rigid3f::inverse() in float32 with Eigen's operation order (transform/rigid_transform.h:167-171)."""
    f = np.float32
    w, x, y, z = f(p[3]), f(-p[4]), f(-p[5]), f(-p[6])
    t = _transform_f32(np.array([0, 0, 0, w, x, y, z], dtype=np.float32), np.asarray(p[:3], dtype=np.float32).reshape(1, 3))[0]
    return np.array([-t[0], -t[1], -t[2], w, x, y, z], dtype=np.float32)


def _transform_f32(pose, pts):
    """rigid * point in float32: uv = 2 (u x v); (v + w uv) + u x uv, then + t."""
    f = np.float32
    pts = np.asarray(pts, dtype=np.float32).reshape(-1, 3)
    w, ux, uy, uz = f(pose[3]), f(pose[4]), f(pose[5]), f(pose[6])
    vx, vy, vz = pts[:, 0], pts[:, 1], pts[:, 2]
    uvx = uy * vz - uz * vy
    uvy = uz * vx - ux * vz
    uvz = ux * vy - uy * vx
    uvx = uvx + uvx
    uvy = uvy + uvy
    uvz = uvz + uvz
    cx = uy * uvz - uz * uvy
    cy = uz * uvx - ux * uvz
    cz = ux * uvy - uy * uvx
    out = np.stack([((vx + w * uvx) + cx) + f(pose[0]), ((vy + w * uvy) + cy) + f(pose[1]),
                    ((vz + w * uvz) + cz) + f(pose[2])], axis=1)
    return out.astype(np.float32)


class FastCorrelativeScanMatcher3D:
    """mapping::scan_matching::FastCorrelativeScanMatcher3D on the device (dliom_fast_csm_*); `nodes` are given
    as the (histogram, yaw) pairs HistogramsAtAnglesFromNodes extracts."""

    def __init__(self, ctx, hybrid_grid, low_resolution_hybrid_grid, node_histograms, node_angles, options):
        self.ctx = ctx
        self._L = ctx._L
        h = _f32(node_histograms).reshape(len(node_angles), -1)
        self.hist_size = h.shape[1]
        self.grids = (hybrid_grid, low_resolution_hybrid_grid)
        o = FastCsmOptions(options["branch_and_bound_depth"], options["full_resolution_depth"],
                           options["min_rotational_score"], options["min_low_resolution_score"],
                           options["linear_xy_search_window"], options["linear_z_search_window"],
                           options["angular_search_window"])
        hnd = _vp()
        _check(self._L.dliom_fast_csm_create(ctx.h, hybrid_grid.h, low_resolution_hybrid_grid.h, _p(h, _f32p),
                                             _p(_f32(node_angles), _f32p), h.shape[0], h.shape[1], C.byref(o),
                                             C.byref(hnd)), "dliom_fast_csm_create")
        self.h = hnd
        self.depth = options["branch_and_bound_depth"]

    def close(self):
        if getattr(self, "h", None):
            self._L.dliom_fast_csm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def level(self, depth):
        """(lo[3], values[nz, ny, nx] uint8) of one pyramid level."""
        lo = np.zeros(3, dtype=np.int32)
        dims = np.zeros(3, dtype=np.int32)
        _check(self._L.dliom_fast_csm_level(self.h, depth, _p(lo, _i32p), _p(dims, _i32p), None, 0), "level")
        v = np.zeros((int(dims[2]), int(dims[1]), int(dims[0])), dtype=np.uint8)
        if v.size:
            _check(self._L.dliom_fast_csm_level(self.h, depth, _p(lo, _i32p), _p(dims, _i32p),
                                                v.ctypes.data_as(C.POINTER(C.c_uint8)), v.size), "level")
        return lo, v

    def _data(self, data):
        self._hi = _f32(data["high_resolution_point_cloud"]).reshape(-1, 3)
        self._lo = _f32(data["low_resolution_point_cloud"]).reshape(-1, 3)
        self._hist = _f32(data["rotational_scan_matcher_histogram"])
        assert len(self._hist) == self.hist_size
        d = FastCsmNodeData()
        for i in range(4):
            d.gravity_alignment[i] = float(data["gravity_alignment"][i])
        d.high_resolution_points = _p(self._hi, _f32p)
        d.num_high_resolution_points = len(self._hi)
        d.low_resolution_points = _p(self._lo, _f32p)
        d.num_low_resolution_points = len(self._lo)
        d.rotational_scan_matcher_histogram = _p(self._hist, _f32p)
        return d

    @staticmethod
    def _result(r):
        return dict(found=bool(r.found), score=np.float32(r.score), rotational_score=np.float32(r.rotational_score),
                    low_resolution_score=np.float32(r.low_resolution_score), num_discrete_scans=r.num_discrete_scans,
                    num_scored_candidates=r.num_scored_candidates, num_score_launches=r.num_score_launches,
                    pose=np.array(r.pose_estimate) if r.found else None)

    def Match(self, global_node_pose, global_submap_pose, data, min_score, ctx=None):
        """ctx: the calling thread's Context (default: the one the matcher was created on)."""
        d, r = self._data(data), FastCsmResult()
        _check(self._L.dliom_fast_csm_match((ctx or self.ctx).h, self.h, _p(_f64(global_node_pose), _f64p), _p(_f64(global_submap_pose), _f64p),
                                            C.byref(d), C.c_float(min_score), C.byref(r)), "dliom_fast_csm_match")
        return self._result(r)

    def MatchFullSubmap(self, global_node_rotation, global_submap_rotation, data, min_score, ctx=None):
        d, r = self._data(data), FastCsmResult()
        _check(self._L.dliom_fast_csm_match_full_submap((ctx or self.ctx).h, self.h, _p(_f64(global_node_rotation), _f64p),
                                                        _p(_f64(global_submap_rotation), _f64p), C.byref(d),
                                                        C.c_float(min_score), C.byref(r)), "dliom_fast_csm_match_full_submap")
        return self._result(r)

    def MatchWith3DofInitial(self, pose_in_submap_guess, data, min_score, ctx=None):
        d, r = self._data(data), FastCsmResult()
        _check(self._L.dliom_fast_csm_match_with_3dof_initial((ctx or self.ctx).h, self.h, _p(_f64(pose_in_submap_guess), _f64p), C.byref(d),
                                                              C.c_float(min_score), C.byref(r)),
               "dliom_fast_csm_match_with_3dof_initial")
        return self._result(r)


def cloud_rotational_histogram(ctx, cloud, histogram_size, rotation_wxyz=None):
    """RotationalScanMatcher::ComputeHistogram on the device (dliom_cloud_rotational_histogram) of a device cloud,
    optionally rotated by a float quaternion first (the gravity alignment)."""
    out = np.zeros(histogram_size, dtype=np.float32)
    rot = None if rotation_wxyz is None else _p(_f32(rotation_wxyz), _f32p)
    _check(load_library().dliom_cloud_rotational_histogram(ctx.h, cloud.h, rot, int(histogram_size), _p(out, _f32p)),
           "dliom_cloud_rotational_histogram")
    return out


def cloud_rotational_histogram_begin(ctx, cloud, histogram_size, rotation_wxyz=None):
    """First half of cloud_rotational_histogram: the kernels go onto the context's auxiliary stream and the call returns."""
    rot = None if rotation_wxyz is None else _p(_f32(rotation_wxyz), _f32p)
    _check(load_library().dliom_cloud_rotational_histogram_begin(ctx.h, cloud.h, rot, int(histogram_size)),
           "dliom_cloud_rotational_histogram_begin")


def cloud_rotational_histogram_finish(ctx, histogram_size):
    """Second half: waits for the pending histogram of the context and returns it."""
    out = np.zeros(histogram_size, dtype=np.float32)
    _check(load_library().dliom_cloud_rotational_histogram_finish(ctx.h, _p(out, _f32p)), "dliom_cloud_rotational_histogram_finish")
    return out


def diag_sequential_sums(ctx, values, acc0=None):
    """dliom_diag_sequential_sums: the device's exact parallel replay of sequential float sums; values (k, n)."""
    v = np.ascontiguousarray(values, dtype=np.float32)
    if v.ndim == 1:
        v = v.reshape(1, -1)
    k, n = v.shape
    a = np.zeros(k, dtype=np.float32) if acc0 is None else np.ascontiguousarray(acc0, dtype=np.float32).reshape(k)
    out = np.zeros(k, dtype=np.float32)
    _check(load_library().dliom_diag_sequential_sums(ctx.h, _p(v, _f32p), k, n, _p(a, _f32p), _p(out, _f32p)),
           "dliom_diag_sequential_sums")
    return out


def diag_histogram_contributions(ctx, cloud, histogram_size, rotation_wxyz=None):
    """dliom_diag_histogram_contributions: (buckets, values) of every addition the device histogram is made of, in order."""
    cap = int(cloud.n) + 1
    buckets = np.zeros(cap, dtype=np.int32)
    values = np.zeros(cap, dtype=np.float32)
    count = C.c_int64(0)
    rot = None if rotation_wxyz is None else _p(_f32(rotation_wxyz), _f32p)
    _check(load_library().dliom_diag_histogram_contributions(ctx.h, cloud.h, rot, int(histogram_size),
                                                             buckets.ctypes.data_as(C.POINTER(C.c_int32)), _p(values, _f32p), cap,
                                                             C.byref(count)), "dliom_diag_histogram_contributions")
    return buckets[:count.value].copy(), values[:count.value].copy()


def diag_std_sort_order(ctx, keys):
    """dliom_diag_std_sort_order: the device's restatement of std::sort's order (ties included); <= 4096 keys through the
    LDS path of the small slices, more through the HBM path."""
    keys = _f32(keys).reshape(-1)
    out = np.zeros(len(keys), dtype=np.int32)
    _check(load_library().dliom_diag_std_sort_order(ctx.h, _p(keys, _f32p), len(keys), out.ctypes.data_as(C.POINTER(C.c_int32))),
           "dliom_diag_std_sort_order")
    return out


def rotational_histogram(points, histogram_size, threads=None):
    """RotationalScanMatcher::ComputeHistogram (host); threads: explicit host thread count (same bits at any)."""
    pts = _f32(points).reshape(-1, 3)
    out = np.zeros(histogram_size, dtype=np.float32)
    if threads is None:
        _check(load_library().dliom_rotational_histogram(_p(pts, _f32p), len(pts), histogram_size, _p(out, _f32p)),
               "dliom_rotational_histogram")
    else:
        _check(load_library().dliom_rotational_histogram_mt(_p(pts, _f32p), len(pts), histogram_size, int(threads), _p(out, _f32p)),
               "dliom_rotational_histogram_mt")
    return out


ERR_DIVERGED = -11


class RangeDataAccumulator:
    """dliom_range_accumulator_*: AddRangeData calls feeding one AddAccumulatedRangeData
    (num_accumulated_range_data > 1), with the RangeDataSynchronizer's origin table."""

    def __init__(self, ctx):
        self.ctx, self._L = ctx, ctx._L
        h = _vp()
        _check(self._L.dliom_range_accumulator_create(ctx.h, C.byref(h)), "dliom_range_accumulator_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self._L.dliom_range_accumulator_destroy(self.h)
            self.h = None

    __del__ = close

    def add(self, prev_pose, predicted_pose, scan_period, ranges_xyzt, min_range, max_range, voxel_filter_size,
            origins=((0.0, 0.0, 0.0),), origin_index=None):
        r = _f32(ranges_xyzt).reshape(-1, 4)
        og = _f32(origins).reshape(-1, 3)
        oi = None if origin_index is None else _f32(origin_index)
        cur = np.zeros(7, dtype=np.float32)
        k = C.c_int()
        _check(self._L.dliom_range_accumulator_add(self.h, _p(_f64(prev_pose), _f64p), _p(_f64(predicted_pose), _f64p),
                                                   float(scan_period), _p(r, _f32p), None if oi is None else _p(oi, _f32p),
                                                   len(r), _p(og, _f32p), len(og), C.c_float(min_range), C.c_float(max_range),
                                                   C.c_float(voxel_filter_size), _p(cur, _f32p), C.byref(k)),
               "dliom_range_accumulator_add")
        return cur, k.value

    def finish(self, voxel_filter_size):
        h = _vp()
        org = np.zeros(3, dtype=np.float32)
        _check(self._L.dliom_range_accumulator_finish(self.h, C.c_float(voxel_filter_size), C.byref(h), _p(org, _f32p)),
               "dliom_range_accumulator_finish")
        return PointCloud(self.ctx, _handle=h), org


class ImuWindow:
    """LocalTrajectoryBuilder3D::WindowOptimize without GTSAM (dliom_imu_window_*; host): IMU-preintegration factor,
    bias random walk, matched-pose prior and gravity factor in a fixed-lag Gauss-Newton smoother."""

    def __init__(self, **overrides):
        self._L = load_library()
        self.options = ImuWindowOptions()
        _check(self._L.dliom_imu_window_default_options(C.byref(self.options)), "dliom_imu_window_default_options")
        for k, v in overrides.items():
            setattr(self.options, k, v)
        h = _vp()
        _check(self._L.dliom_imu_window_create(C.byref(self.options), C.byref(h)), "dliom_imu_window_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self._L.dliom_imu_window_destroy(self.h)
            self.h = None

    __del__ = close

    def initialize(self, pose7, velocity, bias6):
        _check(self._L.dliom_imu_window_initialize(self.h, _p(_f64(pose7), _f64p), _p(_f64(velocity), _f64p),
                                                   _p(_f64(bias6), _f64p)), "dliom_imu_window_initialize")

    def add_imu(self, acc, gyr, dt):
        _check(self._L.dliom_imu_window_add_imu(self.h, _p(_f64(acc), _f64p), _p(_f64(gyr), _f64p), float(dt)),
               "dliom_imu_window_add_imu")

    def add_imu_batch(self, acc, gyr, dt):
        """n samples in one call (acc, gyr: n x 3; dt: scalar or n)."""
        acc, gyr = _f64(acc).reshape(-1, 3), _f64(gyr).reshape(-1, 3)
        dts = np.ascontiguousarray(np.broadcast_to(np.asarray(dt, dtype=np.float64), (len(acc),)))
        _check(self._L.dliom_imu_window_add_imu_batch(self.h, len(acc), _p(acc, _f64p), _p(gyr, _f64p), _p(dts, _f64p)),
               "dliom_imu_window_add_imu_batch")

    def predict(self):
        pose, vel = np.zeros(7), np.zeros(3)
        _check(self._L.dliom_imu_window_predict(self.h, _p(pose, _f64p), _p(vel, _f64p)), "dliom_imu_window_predict")
        return pose, vel

    def add_gravity(self, states_back, direction):
        _check(self._L.dliom_imu_window_add_gravity(self.h, int(states_back), _p(_f64(direction), _f64p)),
               "dliom_imu_window_add_gravity")

    def add_pose(self, matched_pose7, is_drift=False):
        """Returns (pose7, velocity, bias6, status); status is 0 or ERR_DIVERGED (FailureDetection)."""
        pose, vel, bias = np.zeros(7), np.zeros(3), np.zeros(6)
        s = self._L.dliom_imu_window_add_pose(self.h, _p(_f64(matched_pose7), _f64p), int(bool(is_drift)), _p(pose, _f64p),
                                              _p(vel, _f64p), _p(bias, _f64p))
        if s not in (ERR_DIVERGED, ERR_SOLVER):
            _check(s, "dliom_imu_window_add_pose")
        return pose, vel, bias, s

    def window_optimize(self, matched_pose7, is_drift=False):
        """LocalTrajectoryBuilder3D::WindowOptimize as the reference calls it: the first call after initialize() only starts
        the graph and returns the initial state; every later one is add_pose.  Returns (pose7, velocity, bias6, status)."""
        pose, vel, bias = np.zeros(7), np.zeros(3), np.zeros(6)
        s = self._L.dliom_imu_window_window_optimize(self.h, _p(_f64(matched_pose7), _f64p), int(bool(is_drift)), _p(pose, _f64p),
                                                     _p(vel, _f64p), _p(bias, _f64p))
        if s not in (ERR_DIVERGED, ERR_SOLVER):
            _check(s, "dliom_imu_window_window_optimize")
        return pose, vel, bias, s

    def state(self, states_back=0):
        pose, vel, bias = np.zeros(7), np.zeros(3), np.zeros(6)
        _check(self._L.dliom_imu_window_state(self.h, int(states_back), _p(pose, _f64p), _p(vel, _f64p), _p(bias, _f64p)),
               "dliom_imu_window_state")
        return pose, vel, bias

    def __len__(self):
        return int(self._L.dliom_imu_window_size(self.h))

    def solver_stats(self):
        """(linearisation points moved, chain blocks eliminated) so far."""
        a, b = C.c_int64(0), C.c_int64(0)
        _check(self._L.dliom_imu_window_solver_stats(self.h, C.byref(a), C.byref(b)), "dliom_imu_window_solver_stats")
        return int(a.value), int(b.value)

    def gravity_estimate(self):
        """(g_vec_est_G_, passed the reference's gates?, gravity factors added so far) -- EstimateGravity, .cc:1106-1154."""
        g, ok, n = np.zeros(3), C.c_int(0), C.c_int64(0)
        _check(self._L.dliom_imu_window_gravity_estimate(self.h, _p(g, _f64p), C.byref(ok), C.byref(n)),
               "dliom_imu_window_gravity_estimate")
        return g, bool(ok.value), int(n.value)

    def diag_imu_factor_jacobians(self):
        """(analytic, numeric) 15 x 30 Jacobians of the IMU factor between the two newest states."""
        a, b = np.zeros((15, 30)), np.zeros((15, 30))
        _check(self._L.dliom_diag_imu_factor_jacobians(self.h, _p(a, _f64p), _p(b, _f64p)), "dliom_diag_imu_factor_jacobians")
        return a, b


def gravity_estimate(poses7, delta_t, delta_p, delta_v, velocities, gravity_norm, lidar_in_imu_translation=(0.0, 0.0, 0.0)):
    """GravityEstimator::Estimate (gravity_estimator.cc:172-188) -> (gravity in the first frame, accepted)."""
    poses7 = _f64(poses7).reshape(-1, 7)
    g, ok = np.zeros(3), C.c_int(0)
    _check(load_library().dliom_gravity_estimate(len(poses7), _p(poses7, _f64p), _p(_f64(delta_t), _f64p),
                                                 _p(_f64(delta_p).reshape(-1, 3), _f64p), _p(_f64(delta_v).reshape(-1, 3), _f64p),
                                                 _p(_f64(velocities).reshape(-1, 3), _f64p),
                                                 _p(_f64(lidar_in_imu_translation), _f64p), float(gravity_norm), _p(g, _f64p),
                                                 C.byref(ok)), "dliom_gravity_estimate")
    return g, bool(ok.value)


class ImuIntegrator:
    """IMU preintegration between two scans (dliom_imu_integrator_*; host)."""

    def __init__(self, ba, bg, noise):
        self._L = load_library()
        self._noise = ImuNoise(*[float(x) for x in noise])
        h = _vp()
        _check(self._L.dliom_imu_integrator_create(_p(_f64(ba), _f64p), _p(_f64(bg), _f64p), C.byref(self._noise),
                                                   C.byref(h)), "dliom_imu_integrator_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self._L.dliom_imu_integrator_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, ba, bg):
        _check(self._L.dliom_imu_integrator_reset(self.h, _p(_f64(ba), _f64p), _p(_f64(bg), _f64p), C.byref(self._noise)),
               "dliom_imu_integrator_reset")

    def push_back(self, dt, acc, gyr):
        _check(self._L.dliom_imu_integrator_push_back(self.h, float(dt), _p(_f64(acc), _f64p), _p(_f64(gyr), _f64p)),
               "dliom_imu_integrator_push_back")

    def repropagate(self, ba, bg):
        _check(self._L.dliom_imu_integrator_repropagate(self.h, _p(_f64(ba), _f64p), _p(_f64(bg), _f64p)),
               "dliom_imu_integrator_repropagate")

    def get(self):
        o = ImuPreintegration()
        _check(self._L.dliom_imu_integrator_get(self.h, C.byref(o)), "dliom_imu_integrator_get")
        return dict(sum_dt=o.sum_dt, delta_p=np.array(o.delta_p), delta_q=np.array(o.delta_q),
                    delta_v=np.array(o.delta_v), jacobian=np.array(o.jacobian).reshape(15, 15),
                    covariance=np.array(o.covariance).reshape(15, 15))

    def evaluate(self, state_i, state_j, gravity):
        r = np.zeros(15)
        _check(self._L.dliom_imu_integrator_evaluate(self.h, _p(_f64(state_i), _f64p), _p(_f64(state_j), _f64p),
                                                     _p(_f64(gravity), _f64p), _p(r, _f64p)), "dliom_imu_integrator_evaluate")
        return r

    def predict(self, state_i, gravity):
        sj = np.zeros(16)
        _check(self._L.dliom_imu_integrator_predict(self.h, _p(_f64(state_i), _f64p), _p(_f64(gravity), _f64p),
                                                    _p(sj, _f64p)), "dliom_imu_integrator_predict")
        return sj


def rotational_scan_match(node_histograms, node_angles, scan_histogram, initial_angle, angles):
    """RotationalScanMatcher(nodes).Match(histogram, initial_angle, angles) (host)."""
    h = _f32(node_histograms).reshape(len(node_angles), -1)
    a = _f32(angles)
    out = np.zeros(len(a), dtype=np.float32)
    _check(load_library().dliom_rotational_scan_match(_p(h, _f32p), _p(_f32(node_angles), _f32p), h.shape[0], h.shape[1],
                                                      _p(_f32(scan_histogram), _f32p), C.c_float(initial_angle),
                                                      _p(a, _f32p), len(a), _p(out, _f32p)), "dliom_rotational_scan_match")
    return out
