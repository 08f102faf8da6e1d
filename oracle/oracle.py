"""ctypes front end of the CPU oracle (oracle/liboracle.so).

ORACLE = test infrastructure.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product package
(d-liom_amd/) never does.

Conventions: poses are float64[7] = [tx,ty,tz,qw,qx,qy,qz]; clouds are
float32[n,3] C-contiguous.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    """Compile liboracle.so with the committed Makefile (g++ only)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int)
_u16p = C.POINTER(C.c_uint16)
_u64p = C.POINTER(C.c_uint64)


def _declare(L):
    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    vp = C.c_void_p
    sig("orc_value_to_probability_table", None, _f32p)
    sig("orc_value_to_correspondence_cost_table", None, _f32p)
    sig("orc_probability_to_value", C.c_uint16, C.c_float)
    sig("orc_correspondence_cost_to_value", C.c_uint16, C.c_float)
    sig("orc_odds", C.c_float, C.c_float)
    sig("orc_probability_from_odds", C.c_float, C.c_float)
    sig("orc_probability_value_to_correspondence_cost_value", C.c_uint16, C.c_uint16)
    sig("orc_correspondence_cost_value_to_probability_value", C.c_uint16, C.c_uint16)
    sig("orc_lookup_table_to_apply_odds", None, C.c_float, _u16p)
    sig("orc_lookup_table_to_apply_correspondence_cost_odds", None, C.c_float, _u16p)
    sig("orc_grid_new", vp, C.c_float)
    sig("orc_grid_free", None, vp)
    sig("orc_grid_resolution", C.c_float, vp)
    sig("orc_grid_bits", C.c_int, vp)
    sig("orc_cell_indices", None, C.c_float, _f32p, C.c_int, _i32p)
    sig("orc_grid_center_of_cell", None, vp, C.c_int, C.c_int, C.c_int, _f32p)
    sig("orc_grid_set_probability", None, vp, C.c_int, C.c_int, C.c_int, C.c_float)
    sig("orc_grid_set_value", None, vp, C.c_int, C.c_int, C.c_int, C.c_uint16)
    sig("orc_grid_set_values", None, vp, _i32p, _u16p, C.c_int)
    sig("orc_grid_value", C.c_uint16, vp, C.c_int, C.c_int, C.c_int)
    sig("orc_grid_values", None, vp, _i32p, C.c_int, _u16p)
    sig("orc_grid_probability", C.c_float, vp, C.c_int, C.c_int, C.c_int)
    sig("orc_grid_is_known", C.c_int, vp, C.c_int, C.c_int, C.c_int)
    sig("orc_grid_apply_lookup_table", C.c_int, vp, C.c_int, C.c_int, C.c_int, _u16p)
    sig("orc_grid_finish_update", None, vp)
    sig("orc_grid_num_cells", C.c_int64, vp)
    sig("orc_grid_export_cells", None, vp, _i32p, _u16p)
    sig("orc_grid_num_leaves", C.c_int64, vp)
    sig("orc_grid_export_leaves", None, vp, _i32p, _u16p)
    sig("orc_insert_range_data", None, vp, _f32p, _f32p, C.c_int, C.c_double, C.c_double, C.c_int)
    sig("orc_insert_range_data_tables", None, vp, _f32p, _f32p, C.c_int, _u16p, _u16p, C.c_int)
    sig("orc_voxel_filter", C.c_int, C.c_float, _f32p, C.c_int, _i32p)
    sig("orc_adaptive_voxel_filter", C.c_int, C.c_float, C.c_float, C.c_float, _f32p, C.c_int, _f32p)
    sig("orc_transform_points", None, _f32p, _f32p, C.c_int, _f32p)
    sig("orc_rigid3d_multiply", None, _f64p, _f64p, _f64p)
    sig("orc_rigid3d_inverse", None, _f64p, _f64p)
    sig("orc_rtcsm3d_window", None, _f64p, C.c_float, _f32p, C.c_int, _i32p, _i32p, _f32p, _f32p)
    sig("orc_rtcsm3d_candidates", C.c_int64, _f64p, C.c_float, _f32p, C.c_int, _f64p, _f32p, _f32p)
    sig("orc_rtcsm3d_match", C.c_float, _f64p, _f64p, _f32p, C.c_int, vp, _f64p, _f32p, _i32p)
    sig("orc_rtcsm3d_match_range", C.c_float, _f64p, _f64p, _f32p, C.c_int, vp, C.c_int64, C.c_int64,
        C.POINTER(C.c_int64))
    sig("orc_rtcsm3d_float_sums", None, _f64p, _f64p, _f32p, C.c_int, vp, C.POINTER(C.c_int64), C.c_int64, _f32p)
    sig("orc_rtcsm3d_at", None, _f64p, _f64p, _f32p, C.c_int, vp, C.POINTER(C.c_int64), C.c_int64, _u64p, _f32p)
    sig("orc_rtcsm3d_value_sums", None, _f64p, _f64p, _f32p, C.c_int, vp, C.c_int64, C.c_int64, _u64p)
    sig("orc_transform_cell_indices", None, _f32p, _f32p, C.c_int, C.c_float, _i32p)
    sig("orc_interpolated_probability", C.c_double, vp, C.c_double, C.c_double, C.c_double)
    sig("orc_occupied_space_evaluate", None, vp, _f32p, C.c_int, C.c_double, _f64p, _f64p, _f64p, _f64p, _f64p)
    sig("orc_rotation_delta_squared_cost", C.c_double, _f64p, C.c_double, _f64p)
    sig("orc_csm3d_match", None, _f64p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
        _f64p, _f64p, C.POINTER(_f32p), _i32p, C.POINTER(vp), _f64p, _f64p)
    sig("orc_front_end_new", vp, _f64p)
    sig("orc_front_end_free", None, vp)
    sig("orc_front_end_match", None, vp, _f64p, _f32p, _f32p, C.c_int, _f64p)
    sig("orc_front_end_insert", C.c_int, vp, C.c_int64, _f64p, _f64p)
    sig("orc_front_end_num_active_submaps", C.c_int, vp)
    sig("orc_front_end_matching_index", C.c_int, vp)
    sig("orc_front_end_active_submap", None, vp, C.c_int, _f64p, _i32p, C.POINTER(vp), C.POINTER(vp))
    sig("orc_deskew_and_filter", None, _f64p, _f64p, _f64p, _f32p, C.c_int, _f32p, _f32p, _i32p, _i32p, _f32p,
        _f32p, _f32p, _f32p)
    sig("orc_pg_new", vp, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int)
    sig("orc_pg_free", None, vp)
    sig("orc_pg_set_probability", None, vp, C.c_int, C.c_int, C.c_float)
    sig("orc_pg_probability", C.c_float, vp, C.c_int, C.c_int)
    sig("orc_pg_cell_index", None, vp, C.c_float, C.c_float, _i32p)
    sig("orc_pg_cells", None, vp, _u16p)
    sig("orc_pg_limits", None, vp, _f64p)
    sig("orc_pg_insert", None, vp, _f32p, _f32p, C.c_int, C.c_double, C.c_double, C.c_int)
    sig("orc_rtcsm2d_match", C.c_double, _f64p, _f64p, _f32p, C.c_int, vp, _f64p)
    sig("orc_rtcsm2d_score_single", C.c_float, _f64p, _f32p, C.c_int, vp, C.c_int, C.c_int)
    u8p = C.POINTER(C.c_uint8)
    sig("orc_fast_csm_new", vp, vp, vp, _f32p, _f32p, C.c_int, C.c_int, _f64p)
    sig("orc_fast_csm_free", None, vp)
    sig("orc_fast_csm_max_depth", C.c_int, vp)
    sig("orc_fast_csm_stack_num_cells", C.c_int64, vp, C.c_int)
    sig("orc_fast_csm_stack_cells", None, vp, C.c_int, _i32p, u8p)
    sig("orc_fast_csm_match", None, vp, _f64p, _f64p, _f64p, _f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int,
        C.c_float, _f64p, _f64p)
    sig("orc_fast_csm_match_full_submap", None, vp, _f64p, _f64p, _f64p, _f32p, C.c_int, _f32p, C.c_int, _f32p,
        C.c_int, C.c_float, _f64p, _f64p)
    sig("orc_fast_csm_match_3dof", None, vp, _f64p, _f64p, _f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int, C.c_float,
        _f64p, _f64p)
    sig("orc_compute_histogram", None, _f32p, C.c_int, C.c_int, _f32p)
    sig("orc_histogram_contributions", C.c_int, _f32p, C.c_int, C.c_int, C.POINTER(C.c_int), _f32p, C.c_int)
    sig("orc_std_sort_order", None, _f32p, C.c_int, C.POINTER(C.c_int))
    sig("orc_accumulator_new", C.c_void_p)
    sig("orc_accumulator_free", None, C.c_void_p)
    sig("orc_accumulator_add", None, C.c_void_p, _f64p, _f64p, _f64p, _f32p, C.c_int, _i32p, _f32p, C.c_int, _f32p)
    sig("orc_accumulator_finish", C.c_int, C.c_void_p, _f64p, _f32p, C.c_int, _f32p)
    sig("orc_motion_filter_create", C.c_void_p, C.c_double, C.c_double, C.c_double)
    sig("orc_motion_filter_destroy", None, C.c_void_p)
    sig("orc_motion_filter_is_similar", C.c_int, C.c_void_p, C.c_int64, _f64p)
    sig("orc_rotational_match", None, _f32p, _f32p, C.c_int, C.c_int, _f32p, C.c_float, _f32p, C.c_int, _f32p)
    sig("orc_kat_precomputation_grid", C.c_double)
    sig("orc_kat_transform_get_angle", C.c_double)
    sig("orc_kat_rigid_transform", C.c_double, C.c_int)
    sig("orc_kat_fast_csm", C.c_int, C.c_int, _f64p)
    sig("orc_imu_new", vp, _f64p, _f64p, _f64p)
    sig("orc_imu_free", None, vp)
    sig("orc_imu_push_back", None, vp, C.c_double, _f64p, _f64p)
    sig("orc_imu_repropagate", None, vp, _f64p, _f64p)
    sig("orc_imu_get", None, vp, _f64p)
    sig("orc_imu_evaluate", None, vp, _f64p, _f64p, _f64p, _f64p)
    sig("orc_now_seconds", C.c_double)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(t)


def pose(t=(0, 0, 0), q=(1, 0, 0, 0)):
    return np.array(list(t) + list(q), dtype=np.float64)


def angle_axis_quat(angle, axis, normalize_axis=False):
    """Eigen::AngleAxisd -> Quaterniond: w = cos(a/2), vec = sin(a/2) * axis.
    Like Eigen, the axis is NOT normalised unless asked (the reference's
    RotationAroundYZ test passes the raw axis (0,1,1))."""
    axis = np.asarray(axis, dtype=np.float64)
    if normalize_axis:
        axis = axis / np.linalg.norm(axis)
    s = np.sin(0.5 * angle)
    return np.array([np.cos(0.5 * angle), s * axis[0], s * axis[1], s * axis[2]])


def rigid_multiply(a, b):
    out = np.zeros(7)
    lib().orc_rigid3d_multiply(_p(_f64(a), _f64p), _p(_f64(b), _f64p), _p(out, _f64p))
    return out


def rigid_inverse(a):
    out = np.zeros(7)
    lib().orc_rigid3d_inverse(_p(_f64(a), _f64p), _p(out, _f64p))
    return out


# ------------------------------------------------------------------ probability values
def value_to_probability_table():
    out = np.zeros(65536, dtype=np.float32)
    lib().orc_value_to_probability_table(_p(out, _f32p))
    return out


def value_to_correspondence_cost_table():
    out = np.zeros(65536, dtype=np.float32)
    lib().orc_value_to_correspondence_cost_table(_p(out, _f32p))
    return out


def lookup_table_to_apply_odds(odds):
    out = np.zeros(32768, dtype=np.uint16)
    lib().orc_lookup_table_to_apply_odds(C.c_float(odds), _p(out, _u16p))
    return out


def lookup_table_to_apply_correspondence_cost_odds(odds):
    out = np.zeros(32768, dtype=np.uint16)
    lib().orc_lookup_table_to_apply_correspondence_cost_odds(C.c_float(odds), _p(out, _u16p))
    return out


def odds(p):
    return lib().orc_odds(C.c_float(p))


def probability_from_odds(o):
    return lib().orc_probability_from_odds(C.c_float(o))


def probability_to_value(p):
    return lib().orc_probability_to_value(C.c_float(p))


def correspondence_cost_to_value(c):
    return lib().orc_correspondence_cost_to_value(C.c_float(c))


def cell_indices(resolution, pts):
    pts = _f32(pts).reshape(-1, 3)
    out = np.zeros((len(pts), 3), dtype=np.int32)
    lib().orc_cell_indices(C.c_float(resolution), _p(pts, _f32p), len(pts), _p(out, _i32p))
    return out


# ------------------------------------------------------------------ HybridGrid
class HybridGrid:
    def __init__(self, resolution, borrowed_handle=None):
        self._L = lib()
        self._owned = borrowed_handle is None
        self.h = C.c_void_p(self._L.orc_grid_new(C.c_float(resolution))) if self._owned else borrowed_handle

    def __del__(self):
        try:
            if self._owned:
                self._L.orc_grid_free(self.h)
        except Exception:
            pass

    @property
    def resolution(self):
        return self._L.orc_grid_resolution(self.h)

    @property
    def bits(self):
        return self._L.orc_grid_bits(self.h)

    def get_cell_index(self, p):
        return cell_indices(self.resolution, np.asarray(p, dtype=np.float32).reshape(1, 3))[0]

    def get_center_of_cell(self, idx):
        out = np.zeros(3, dtype=np.float32)
        self._L.orc_grid_center_of_cell(self.h, int(idx[0]), int(idx[1]), int(idx[2]), _p(out, _f32p))
        return out

    def set_probability(self, idx, p):
        self._L.orc_grid_set_probability(self.h, int(idx[0]), int(idx[1]), int(idx[2]), C.c_float(p))

    def set_value(self, idx, v):
        self._L.orc_grid_set_value(self.h, int(idx[0]), int(idx[1]), int(idx[2]), int(v))

    def set_values(self, xyz, v):
        xyz = _i32(xyz).reshape(-1, 3)
        v = np.ascontiguousarray(v, dtype=np.uint16)
        self._L.orc_grid_set_values(self.h, _p(xyz, _i32p), _p(v, _u16p), len(v))

    def value(self, idx):
        return self._L.orc_grid_value(self.h, int(idx[0]), int(idx[1]), int(idx[2]))

    def values(self, xyz):
        xyz = _i32(xyz).reshape(-1, 3)
        out = np.zeros(len(xyz), dtype=np.uint16)
        self._L.orc_grid_values(self.h, _p(xyz, _i32p), len(xyz), _p(out, _u16p))
        return out

    def get_probability(self, idx):
        return self._L.orc_grid_probability(self.h, int(idx[0]), int(idx[1]), int(idx[2]))

    def is_known(self, idx):
        return bool(self._L.orc_grid_is_known(self.h, int(idx[0]), int(idx[1]), int(idx[2])))

    def apply_lookup_table(self, idx, table):
        table = np.ascontiguousarray(table, dtype=np.uint16)
        return bool(self._L.orc_grid_apply_lookup_table(
            self.h, int(idx[0]), int(idx[1]), int(idx[2]), _p(table, _u16p)))

    def finish_update(self):
        self._L.orc_grid_finish_update(self.h)

    def export_cells(self):
        n = self._L.orc_grid_num_cells(self.h)
        xyz = np.zeros((n, 3), dtype=np.int32)
        v = np.zeros(n, dtype=np.uint16)
        if n:
            self._L.orc_grid_export_cells(self.h, _p(xyz, _i32p), _p(v, _u16p))
        return xyz, v

    def export_leaves(self):
        n = self._L.orc_grid_num_leaves(self.h)
        origin = np.zeros((n, 3), dtype=np.int32)
        v = np.zeros((n, 512), dtype=np.uint16)
        if n:
            self._L.orc_grid_export_leaves(self.h, _p(origin, _i32p), _p(v, _u16p))
        return origin, v

    def insert(self, origin, returns, hit_probability, miss_probability, num_free_space_voxels):
        origin = _f32(origin)
        returns = _f32(returns).reshape(-1, 3)
        self._L.orc_insert_range_data(self.h, _p(origin, _f32p), _p(returns, _f32p), len(returns),
                                      hit_probability, miss_probability, num_free_space_voxels)

    def insert_tables(self, origin, returns, hit_table, miss_table, num_free_space_voxels):
        origin = _f32(origin)
        returns = _f32(returns).reshape(-1, 3)
        hit_table = np.ascontiguousarray(hit_table, dtype=np.uint16)
        miss_table = np.ascontiguousarray(miss_table, dtype=np.uint16)
        self._L.orc_insert_range_data_tables(self.h, _p(origin, _f32p), _p(returns, _f32p), len(returns),
                                             _p(hit_table, _u16p), _p(miss_table, _u16p),
                                             num_free_space_voxels)

    def interpolated_probability(self, x, y, z):
        return self._L.orc_interpolated_probability(self.h, x, y, z)


# ------------------------------------------------------------------ filters / transforms
def voxel_filter(size, pts):
    pts = _f32(pts).reshape(-1, 3)
    keep = np.zeros(len(pts), dtype=np.int32)
    n = lib().orc_voxel_filter(C.c_float(size), _p(pts, _f32p), len(pts), _p(keep, _i32p))
    return keep[:n].copy()


def adaptive_voxel_filter(max_length, min_num_points, max_range, pts):
    pts = _f32(pts).reshape(-1, 3)
    out = np.zeros_like(pts)
    n = lib().orc_adaptive_voxel_filter(C.c_float(max_length), C.c_float(min_num_points),
                                        C.c_float(max_range), _p(pts, _f32p), len(pts), _p(out, _f32p))
    return out[:n].copy()


def transform_points(pose7_f32, pts):
    pose7 = _f32(pose7_f32)
    pts = _f32(pts).reshape(-1, 3)
    out = np.zeros_like(pts)
    lib().orc_transform_points(_p(pose7, _f32p), _p(pts, _f32p), len(pts), _p(out, _f32p))
    return out


def transform_cell_indices(pose7_f32, pts, resolution):
    pose7 = _f32(pose7_f32)
    pts = _f32(pts).reshape(-1, 3)
    out = np.zeros((len(pts), 3), dtype=np.int32)
    lib().orc_transform_cell_indices(_p(pose7, _f32p), _p(pts, _f32p), len(pts),
                                     C.c_float(resolution), _p(out, _i32p))
    return out


# ------------------------------------------------------------------ RTCSM3D
def _opts4(o):
    return _f64([o["linear_search_window"], o["angular_search_window"],
                 o["translation_delta_cost_weight"], o["rotation_delta_cost_weight"]])


def rtcsm3d_window(opts, resolution, pts):
    pts = _f32(pts).reshape(-1, 3)
    lw, aw = C.c_int(), C.c_int()
    step, rng = C.c_float(), C.c_float()
    lib().orc_rtcsm3d_window(_p(_opts4(opts), _f64p), C.c_float(resolution), _p(pts, _f32p), len(pts),
                             C.byref(lw), C.byref(aw), C.byref(step), C.byref(rng))
    return dict(linear_window=lw.value, angular_window=aw.value, angular_step=step.value,
                max_scan_range=rng.value)


def rtcsm3d_candidates(opts, resolution, pts, init7):
    pts = _f32(pts).reshape(-1, 3)
    o = _opts4(opts)
    init7 = _f64(init7)
    n = lib().orc_rtcsm3d_candidates(_p(o, _f64p), C.c_float(resolution), _p(pts, _f32p), len(pts),
                                     _p(init7, _f64p), None, None)
    tr = np.zeros((n, 7), dtype=np.float32)
    ca = np.zeros((n, 7), dtype=np.float32)
    lib().orc_rtcsm3d_candidates(_p(o, _f64p), C.c_float(resolution), _p(pts, _f32p), len(pts),
                                 _p(init7, _f64p), _p(tr, _f32p), _p(ca, _f32p))
    return tr, ca


def rtcsm3d_match(opts, init7, pts, grid, want_scores=False):
    pts = _f32(pts).reshape(-1, 3)
    o = _opts4(opts)
    init7 = _f64(init7)
    out = np.zeros(7)
    best = C.c_int(-1)
    scores = None
    sp = None
    if want_scores:
        n = lib().orc_rtcsm3d_candidates(_p(o, _f64p), C.c_float(grid.resolution), _p(pts, _f32p),
                                         len(pts), _p(init7, _f64p), None, None)
        scores = np.zeros(n, dtype=np.float32)
        sp = _p(scores, _f32p)
    s = lib().orc_rtcsm3d_match(_p(o, _f64p), _p(init7, _f64p), _p(pts, _f32p), len(pts), grid.h,
                                _p(out, _f64p), sp, C.byref(best))
    return dict(score=s, pose=out, best_index=best.value, scores=scores)


def rtcsm3d_match_range(opts, init7, pts, grid, first, count):
    """The reference's candidate loop over [first, first+count) only (cpu_baseline sampling)."""
    pts = _f32(pts).reshape(-1, 3)
    best = C.c_int64(-1)
    s = lib().orc_rtcsm3d_match_range(_p(_opts4(opts), _f64p), _p(_f64(init7), _f64p), _p(pts, _f32p), len(pts),
                                      grid.h, first, count, C.byref(best))
    return s, best.value


class FlatGridIndex:
    """orc_flat_grid_new: the flat leaf table of BASELINE.md section 2's "fair-CPU" variant, built once per grid state."""

    def __init__(self, grid):
        self._L = lib()
        self._L.orc_flat_grid_new.restype = C.c_void_p
        self.grid = grid
        self.h = C.c_void_p(self._L.orc_flat_grid_new(grid.h))

    def __del__(self):
        try:
            self._L.orc_flat_grid_free(self.h)
        except Exception:
            pass


def rtcsm3d_match_range_fair(opts, init7, pts, flat, first, count):
    """The reference's Match loop on candidates [first, first + count) in the fair-CPU variant (flat leaf table, no
    per-candidate allocation; same arithmetic, hence the same scores).  Returns (best score of the range, its index)."""
    pts = _f32(pts).reshape(-1, 3)
    best = C.c_int64(-1)
    L = lib()
    L.orc_rtcsm3d_match_range_fair.restype = C.c_float
    s = L.orc_rtcsm3d_match_range_fair(_p(_opts4(opts), _f64p), _p(_f64(init7), _f64p), _p(pts, _f32p), len(pts), flat.grid.h,
                                       flat.h, C.c_int64(first), C.c_int64(count), C.byref(best))
    return s, best.value


def rtcsm3d_volume_fair(opts, init7, pts, flat, threads=8, progress=None):
    """Every candidate's integer value sum and reference score (the whole Match loop, fair-CPU layout) on `threads` host
    threads: (uint64[C], float32[C]).  The reference's winner is the first maximum of the scores (strict `>`,
    rtcsm_3d.cc:46-51) = np.argmax."""
    from concurrent.futures import ThreadPoolExecutor
    pts = _f32(pts).reshape(-1, 3)
    o, i7 = _opts4(opts), _f64(init7)
    L = lib()
    total = L.orc_rtcsm3d_candidates(_p(o, _f64p), C.c_float(flat.grid.resolution), _p(pts, _f32p), len(pts), _p(i7, _f64p), None, None)
    sums = np.zeros(total, dtype=np.uint64)
    scores = np.zeros(total, dtype=np.float32)
    L.orc_rtcsm3d_range_fair_volume.restype = None
    done = [0]

    def run(fc):
        f, c = fc
        L.orc_rtcsm3d_range_fair_volume(_p(o, _f64p), _p(i7, _f64p), _p(pts, _f32p), len(pts), flat.grid.h, flat.h,
                                        C.c_int64(f), C.c_int64(c), _p(sums[f:f + c], _u64p), _p(scores[f:f + c], _f32p))
        done[0] += c
        if progress is not None:
            progress(done[0], total)
    parts = _ranges(total, max(1, threads) * 16)
    with ThreadPoolExecutor(max(1, threads)) as pool:
        list(pool.map(run, parts))
    return sums, scores


def _ranges(total, parts):
    step = (total + parts - 1) // parts
    return [(f, min(step, total - f)) for f in range(0, total, step)]


def rtcsm3d_match_parallel(opts, init7, pts, grid, threads=8):
    """RealTimeCorrelativeScanMatcher3D::Match with the candidate loop cut into contiguous ranges that run
    on `threads` host threads (ctypes releases the GIL); the ranges are combined in generation order with
    the reference's strict `>`, so the result is the serial loop's (rtcsm_3d.cc:40-51)."""
    from concurrent.futures import ThreadPoolExecutor
    pts = _f32(pts).reshape(-1, 3)
    o = _opts4(opts)
    total = lib().orc_rtcsm3d_candidates(_p(o, _f64p), C.c_float(grid.resolution), _p(pts, _f32p), len(pts),
                                         _p(_f64(init7), _f64p), None, None)
    parts = _ranges(total, max(1, threads) * 4)
    with ThreadPoolExecutor(max(1, threads)) as pool:
        res = list(pool.map(lambda fc: rtcsm3d_match_range(opts, init7, pts, grid, fc[0], fc[1]), parts))
    best, best_c = np.float32(-1.0), -1
    for s, c in res:
        if np.float32(s) > best:
            best, best_c = np.float32(s), c
    _, ca = rtcsm3d_candidates(opts, grid.resolution, pts, init7)
    return dict(score=float(best), best_index=int(best_c), pose=ca[best_c].astype(np.float64), num_candidates=total)


def rtcsm3d_value_sums_parallel(opts, init7, pts, grid, threads=8):
    """The whole integer score volume (sum_i max(v_i & 0x7fff, 1) per candidate) on `threads` host threads."""
    from concurrent.futures import ThreadPoolExecutor
    pts = _f32(pts).reshape(-1, 3)
    o = _opts4(opts)
    total = lib().orc_rtcsm3d_candidates(_p(o, _f64p), C.c_float(grid.resolution), _p(pts, _f32p), len(pts),
                                         _p(_f64(init7), _f64p), None, None)
    parts = _ranges(total, max(1, threads) * 4)
    with ThreadPoolExecutor(max(1, threads)) as pool:
        res = list(pool.map(lambda fc: rtcsm3d_value_sums(opts, init7, pts, grid, fc[0], fc[1]), parts))
    return np.concatenate(res)


def rtcsm3d_float_sums(opts, init7, pts, grid, indices):
    """Sequential float sums (before the division by N) of the given candidates."""
    pts = _f32(pts).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, dtype=np.int64)
    out = np.zeros(len(idx), dtype=np.float32)
    lib().orc_rtcsm3d_float_sums(_p(_opts4(opts), _f64p), _p(_f64(init7), _f64p), _p(pts, _f32p), len(pts), grid.h,
                                 idx.ctypes.data_as(C.POINTER(C.c_int64)), len(idx), _p(out, _f32p))
    return out


def rtcsm3d_at(opts, init7, pts, grid, indices, threads=8):
    """Integer value sums and reference scores (ScoreCandidate) of the candidates `indices`, on host threads."""
    from concurrent.futures import ThreadPoolExecutor
    pts = _f32(pts).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, dtype=np.int64)
    sums = np.zeros(len(idx), dtype=np.uint64)
    scores = np.zeros(len(idx), dtype=np.float32)
    o, i7 = _opts4(opts), _f64(init7)

    def run(fc):
        f, c = fc
        if c > 0:
            lib().orc_rtcsm3d_at(_p(o, _f64p), _p(i7, _f64p), _p(pts, _f32p), len(pts), grid.h,
                                 idx[f:f + c].ctypes.data_as(C.POINTER(C.c_int64)), c, _p(sums[f:f + c], _u64p),
                                 _p(scores[f:f + c], _f32p))
    parts = _ranges(len(idx), max(1, threads)) if len(idx) else []
    with ThreadPoolExecutor(max(1, threads)) as pool:
        list(pool.map(run, parts))
    return sums, scores


def rtcsm3d_value_sums(opts, init7, pts, grid, first=0, count=-1):
    pts = _f32(pts).reshape(-1, 3)
    o = _opts4(opts)
    init7 = _f64(init7)
    if count < 0:
        total = lib().orc_rtcsm3d_candidates(_p(o, _f64p), C.c_float(grid.resolution), _p(pts, _f32p),
                                             len(pts), _p(init7, _f64p), None, None)
        n = total - first
    else:
        n = count
    sums = np.zeros(n, dtype=np.uint64)
    lib().orc_rtcsm3d_value_sums(_p(o, _f64p), _p(init7, _f64p), _p(pts, _f32p), len(pts), grid.h,
                                 first, n, _p(sums, _u64p))
    return sums


# ------------------------------------------------------------------ CSM3D
def occupied_space_evaluate(grid, pts, scaling, t3, q4, jacobians=True):
    pts = _f32(pts).reshape(-1, 3)
    n = len(pts)
    r = np.zeros(n)
    jt = np.zeros((n, 3)) if jacobians else None
    jq = np.zeros((n, 4)) if jacobians else None
    lib().orc_occupied_space_evaluate(grid.h, _p(pts, _f32p), n, scaling, _p(_f64(t3), _f64p),
                                      _p(_f64(q4), _f64p), _p(r, _f64p),
                                      _p(jt, _f64p) if jacobians else None,
                                      _p(jq, _f64p) if jacobians else None)
    return r, jt, jq


def rotation_delta_squared_cost(q4, scaling, target4):
    return lib().orc_rotation_delta_squared_cost(_p(_f64(q4), _f64p), scaling, _p(_f64(target4), _f64p))


def csm3d_match(opts, target_translation, init7, clouds_and_grids):
    """opts: dict(occupied_space_weight=[..], translation_weight, rotation_weight,
    only_optimize_yaw, use_nonmonotonic_steps, max_num_iterations)."""
    k = len(clouds_and_grids)
    w = _f64(opts["occupied_space_weight"])
    clouds = [_f32(c).reshape(-1, 3) for c, _ in clouds_and_grids]
    ptrs = (_f32p * k)(*[_p(c, _f32p) for c in clouds])
    ns = _i32([len(c) for c in clouds])
    grids = (C.c_void_p * k)(*[g.h for _, g in clouds_and_grids])
    out = np.zeros(7)
    summ = np.zeros(10)
    lib().orc_csm3d_match(_p(w, _f64p), k, opts["translation_weight"], opts["rotation_weight"],
                          int(opts.get("only_optimize_yaw", False)),
                          int(opts.get("use_nonmonotonic_steps", False)),
                          int(opts["max_num_iterations"]), _p(_f64(target_translation), _f64p),
                          _p(_f64(init7), _f64p), ptrs, _p(ns, _i32p), grids, _p(out, _f64p),
                          _p(summ, _f64p))
    return dict(pose=out, initial_cost=summ[0], final_cost=summ[1], num_successful_steps=int(summ[2]),
                num_unsuccessful_steps=int(summ[3]), num_iterations=int(summ[4]),
                num_residual_evaluations=int(summ[5]), num_jacobian_evaluations=int(summ[6]),
                termination_type=int(summ[7]))


# ------------------------------------------------------------------ 2D
class ProbabilityGrid:
    def __init__(self, resolution, max_xy, num_x_cells, num_y_cells):
        self._L = lib()
        self.num_x_cells = num_x_cells
        self.num_y_cells = num_y_cells
        self.h = C.c_void_p(self._L.orc_pg_new(resolution, max_xy[0], max_xy[1], num_x_cells, num_y_cells))

    def __del__(self):
        try:
            self._L.orc_pg_free(self.h)
        except Exception:
            pass

    def set_probability(self, x, y, p):
        self._L.orc_pg_set_probability(self.h, x, y, C.c_float(p))

    def get_probability(self, x, y):
        return self._L.orc_pg_probability(self.h, x, y)

    def cell_index(self, px, py):
        out = np.zeros(2, dtype=np.int32)
        self._L.orc_pg_cell_index(self.h, C.c_float(px), C.c_float(py), _p(out, _i32p))
        return out

    def cells(self):
        """uint16 correspondence-cost cells [num_y_cells, num_x_cells] (the grid may have grown)."""
        lim = np.zeros(5)
        self._L.orc_pg_limits(self.h, _p(lim, _f64p))
        self.resolution, self.max_xy = lim[0], (lim[1], lim[2])
        self.num_x_cells, self.num_y_cells = int(lim[3]), int(lim[4])
        out = np.zeros((self.num_y_cells, self.num_x_cells), dtype=np.uint16)
        self._L.orc_pg_cells(self.h, _p(out, _u16p))
        return out

    def insert(self, origin, returns, hit_probability, miss_probability, insert_free_space=True):
        origin = _f32(origin)
        returns = _f32(returns).reshape(-1, 3)
        self._L.orc_pg_insert(self.h, _p(origin, _f32p), _p(returns, _f32p), len(returns),
                              hit_probability, miss_probability, int(insert_free_space))


def rtcsm2d_match(opts, init3, pts, pg):
    pts = _f32(pts).reshape(-1, 3)
    out = np.zeros(3)
    s = lib().orc_rtcsm2d_match(_p(_opts4(opts), _f64p), _p(_f64(init3), _f64p), _p(pts, _f32p),
                                len(pts), pg.h, _p(out, _f64p))
    return dict(score=s, pose=out)


def rtcsm2d_score_single(opts, pts, pg, x_index_offset, y_index_offset):
    pts = _f32(pts).reshape(-1, 3)
    return lib().orc_rtcsm2d_score_single(_p(_opts4(opts), _f64p), _p(pts, _f32p), len(pts), pg.h,
                                          x_index_offset, y_index_offset)


# ------------------------------------------------------------------ front end
def front_end_options_vector(o):
    """Flattens the dict used by both the oracle and the product front end (see tests)."""
    hi, lo = o["high_resolution_adaptive_voxel_filter"], o["low_resolution_adaptive_voxel_filter"]
    r, c, m, s = o["real_time_correlative_scan_matcher"], o["ceres_scan_matcher"], o["motion_filter"], o["submaps"]
    return _f64([hi["max_length"], hi["min_num_points"], hi["max_range"], lo["max_length"], lo["min_num_points"],
                 lo["max_range"], float(o["use_online_correlative_scan_matching"]), r["linear_search_window"],
                 r["angular_search_window"], r["translation_delta_cost_weight"], r["rotation_delta_cost_weight"],
                 c["occupied_space_weight"][0], c["occupied_space_weight"][1], c["translation_weight"],
                 c["rotation_weight"], float(c.get("only_optimize_yaw", False)),
                 float(c.get("use_nonmonotonic_steps", False)), c["max_num_iterations"], m["max_time_seconds"],
                 m["max_distance_meters"], m["max_angle_radians"], s["high_resolution"],
                 s["high_resolution_max_range"], s["low_resolution"], s["num_range_data"], s["hit_probability"],
                 s["miss_probability"], s["num_free_space_voxels"]])


class FrontEnd:
    def __init__(self, options):
        self._L = lib()
        v = front_end_options_vector(options)
        self.h = C.c_void_p(self._L.orc_front_end_new(_p(v, _f64p)))

    def __del__(self):
        try:
            self._L.orc_front_end_free(self.h)
        except Exception:
            pass

    def set_threads(self, threads):
        """The RTCSM3D candidate loop on `threads` host threads (a CPU baseline variant: the reference's loop is serial)."""
        self._L.orc_front_end_set_threads(self.h, int(threads))

    def match(self, pose_prediction, origin, returns):
        returns = _f32(returns).reshape(-1, 3)
        out = np.zeros(27)
        self._L.orc_front_end_match(self.h, _p(_f64(pose_prediction), _f64p), _p(_f32(origin), _f32p),
                                    _p(returns, _f32p), len(returns), _p(out, _f64p))
        return dict(dropped=bool(out[0]), pose_estimate=out[1:8].copy(), pose_observation_in_submap=out[8:15].copy(),
                    initial_ceres_pose=out[15:22].copy(), rtcsm_score=float(out[22]), final_cost=float(out[23]),
                    num_iterations=int(out[24]), num_high=int(out[25]), num_low=int(out[26]))

    def insert(self, time_ticks, pose, gravity_alignment):
        return self._L.orc_front_end_insert(self.h, int(time_ticks), _p(_f64(pose), _f64p),
                                            _p(_f64(gravity_alignment), _f64p))

    def num_active_submaps(self):
        return self._L.orc_front_end_num_active_submaps(self.h)

    def matching_index(self):
        return self._L.orc_front_end_matching_index(self.h)

    def active_submap(self, i, resolutions):
        pose = np.zeros(7)
        n = C.c_int()
        hi, lo = C.c_void_p(), C.c_void_p()
        self._L.orc_front_end_active_submap(self.h, i, _p(pose, _f64p), C.byref(n), C.byref(hi), C.byref(lo))
        return dict(local_pose=pose, num_range_data=n.value, hi=HybridGrid(resolutions[0], borrowed_handle=hi),
                    lo=HybridGrid(resolutions[1], borrowed_handle=lo), _keepalive=self)


# ------------------------------------------------------------------ AddRangeData pre-processing
def deskew_and_filter(scan_period, min_range, max_range, voxel_filter_size, prev_pose, cur_pose, ranges_xyzt,
                      origin=(0.0, 0.0, 0.0)):
    """local_trajectory_builder_3d.cc:393-487.  Returns a dict with the per-hit local points and
    gate decisions and the filtered range data in the tracking frame."""
    r = _f32(ranges_xyzt).reshape(-1, 4)
    n = len(r)
    hits = np.zeros((n, 3), dtype=np.float32)
    kind = np.zeros(n, dtype=np.int32)
    counts = np.zeros(3, dtype=np.int32)
    ret = np.zeros((n, 3), dtype=np.float32)
    mis = np.zeros((n, 3), dtype=np.float32)
    cur = np.zeros(7, dtype=np.float32)
    org = np.zeros(3, dtype=np.float32)
    lib().orc_deskew_and_filter(_p(_f64([scan_period, min_range, max_range, voxel_filter_size]), _f64p),
                                _p(_f64(prev_pose), _f64p), _p(_f64(cur_pose), _f64p), _p(r, _f32p), n,
                                _p(_f32(origin), _f32p), _p(hits, _f32p), _p(kind, _i32p), _p(counts, _i32p),
                                _p(ret, _f32p), _p(mis, _f32p), _p(cur, _f32p), _p(org, _f32p))
    return dict(hits_in_local=hits[:counts[0]].copy(), kind=kind[:counts[0]].copy(),
                returns_in_tracking=ret[:counts[1]].copy(), misses_in_tracking=mis[:counts[2]].copy(),
                current_pose=cur, origin_in_tracking=org)


class RangeDataAccumulator:
    """AddRangeData with the synchronizer's origin table and num_accumulated_range_data > 1
    (local_trajectory_builder_3d.cc:393-487)."""

    def __init__(self, scan_period, min_range, max_range, voxel_filter_size):
        self.opts = _f64([scan_period, min_range, max_range, voxel_filter_size])
        self.h = lib().orc_accumulator_new()
        self.capacity = 0

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_accumulator_free(self.h)
            self.h = None

    def add(self, prev_pose, cur_pose, ranges_xyzt, origins=((0.0, 0.0, 0.0),), origin_index=None):
        r = _f32(ranges_xyzt).reshape(-1, 4)
        og = _f32(origins).reshape(-1, 3)
        oi = None if origin_index is None else np.ascontiguousarray(origin_index, dtype=np.int32)
        cur = np.zeros(7, dtype=np.float32)
        lib().orc_accumulator_add(self.h, _p(self.opts, _f64p), _p(_f64(prev_pose), _f64p), _p(_f64(cur_pose), _f64p),
                                  _p(r, _f32p), len(r), None if oi is None else _p(oi, _i32p), _p(og, _f32p), len(og),
                                  _p(cur, _f32p))
        self.capacity += len(r)
        return cur

    def finish(self):
        out = np.zeros((max(self.capacity, 1), 3), dtype=np.float32)
        org = np.zeros(3, dtype=np.float32)
        n = lib().orc_accumulator_finish(self.h, _p(self.opts, _f64p), _p(out, _f32p), len(out), _p(org, _f32p))
        self.capacity = 0
        return out[:n].copy(), org


# ------------------------------------------------------------------ fast correlative scan matcher 3D
def fast_options(o):
    """dict(branch_and_bound_depth, full_resolution_depth, min_rotational_score, min_low_resolution_score,
    linear_xy_search_window, linear_z_search_window, angular_search_window) -> float64[7]."""
    return _f64([o["branch_and_bound_depth"], o["full_resolution_depth"], o["min_rotational_score"],
                 o["min_low_resolution_score"], o["linear_xy_search_window"], o["linear_z_search_window"],
                 o["angular_search_window"]])


def compute_histogram(pts, histogram_size):
    pts = _f32(pts).reshape(-1, 3)
    out = np.zeros(histogram_size, dtype=np.float32)
    lib().orc_compute_histogram(_p(pts, _f32p), len(pts), histogram_size, _p(out, _f32p))
    return out


def histogram_contributions(pts, histogram_size):
    """(buckets, values) of every `histogram(bucket) += value` of ComputeHistogram, in the order they happen."""
    pts = _f32(pts).reshape(-1, 3)
    cap = len(pts) + 1
    buckets = np.zeros(cap, dtype=np.int32)
    values = np.zeros(cap, dtype=np.float32)
    total = lib().orc_histogram_contributions(_p(pts, _f32p), len(pts), histogram_size,
                                              buckets.ctypes.data_as(C.POINTER(C.c_int)), _p(values, _f32p), cap)
    return buckets[:total].copy(), values[:total].copy()


def std_sort_order(keys):
    """Indices in the order std::sort (this machine's libstdc++) leaves (key, index) pairs compared by key only."""
    keys = _f32(keys).reshape(-1)
    out = np.zeros(len(keys), dtype=np.int32)
    lib().orc_std_sort_order(_p(keys, _f32p), len(keys), out.ctypes.data_as(C.POINTER(C.c_int)))
    return out


class MotionFilter:
    """mapping/internal/motion_filter.cc:40-58; time in common::Time ticks (100 ns)."""

    def __init__(self, max_time_seconds, max_distance_meters, max_angle_radians):
        self.h = lib().orc_motion_filter_create(max_time_seconds, max_distance_meters, max_angle_radians)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_motion_filter_destroy(self.h)
            self.h = None

    def is_similar(self, time_ticks, pose7):
        return bool(lib().orc_motion_filter_is_similar(self.h, int(time_ticks), _p(_f64(pose7), _f64p)))


def rotational_match(node_histograms, node_angles, scan_histogram, initial_angle, angles):
    h = _f32(node_histograms).reshape(len(node_angles), -1)
    a = _f32(angles)
    out = np.zeros(len(a), dtype=np.float32)
    lib().orc_rotational_match(_p(h, _f32p), _p(_f32(node_angles), _f32p), h.shape[0], h.shape[1],
                               _p(_f32(scan_histogram), _f32p), C.c_float(initial_angle), _p(a, _f32p), len(a),
                               _p(out, _f32p))
    return out


class FastCorrelativeScanMatcher3D:
    """FastCorrelativeScanMatcher3D(hybrid_grid, low_resolution_hybrid_grid, nodes, options) with the nodes
    given as (histogram, yaw) pairs (HistogramsAtAnglesFromNodes, fast_correlative_scan_matcher_3d.cc:114-127)."""

    def __init__(self, hi_grid, lo_grid, node_histograms, node_angles, options):
        h = _f32(node_histograms).reshape(len(node_angles), -1)
        self.hist_size = h.shape[1]
        self.grids = (hi_grid, lo_grid)  # keep alive
        self.h = lib().orc_fast_csm_new(hi_grid.h, lo_grid.h, _p(h, _f32p), _p(_f32(node_angles), _f32p), h.shape[0],
                                        h.shape[1], _p(fast_options(options), _f64p))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_fast_csm_free(self.h)
            self.h = None

    def max_depth(self):
        return lib().orc_fast_csm_max_depth(self.h)

    def stack_cells(self, depth):
        n = lib().orc_fast_csm_stack_num_cells(self.h, depth)
        xyz = np.zeros((n, 3), dtype=np.int32)
        v = np.zeros(n, dtype=np.uint8)
        lib().orc_fast_csm_stack_cells(self.h, depth, _p(xyz, _i32p), v.ctypes.data_as(C.POINTER(C.c_uint8)))
        return xyz, v

    @staticmethod
    def _result(pose, out):
        return dict(found=bool(out[0]), score=np.float32(out[1]), rotational_score=np.float32(out[2]),
                    low_resolution_score=np.float32(out[3]), num_scored_candidates=int(out[4]),
                    num_discrete_scans=int(out[5]), pose=pose if out[0] else None)

    def _data(self, data):
        hi = _f32(data["high_resolution_point_cloud"]).reshape(-1, 3)
        lo = _f32(data["low_resolution_point_cloud"]).reshape(-1, 3)
        hist = _f32(data["rotational_scan_matcher_histogram"])
        assert len(hist) == self.hist_size
        return _f64(data["gravity_alignment"]), hi, lo, hist

    def Match(self, global_node_pose, global_submap_pose, data, min_score):
        g, hi, lo, hist = self._data(data)
        pose, out = np.zeros(7), np.zeros(6)
        lib().orc_fast_csm_match(self.h, _p(_f64(global_node_pose), _f64p), _p(_f64(global_submap_pose), _f64p),
                                 _p(g, _f64p), _p(hi, _f32p), len(hi), _p(lo, _f32p), len(lo), _p(hist, _f32p), len(hist),
                                 C.c_float(min_score), _p(pose, _f64p), _p(out, _f64p))
        return self._result(pose, out)

    def MatchFullSubmap(self, global_node_rotation, global_submap_rotation, data, min_score):
        g, hi, lo, hist = self._data(data)
        pose, out = np.zeros(7), np.zeros(6)
        lib().orc_fast_csm_match_full_submap(self.h, _p(_f64(global_node_rotation), _f64p),
                                             _p(_f64(global_submap_rotation), _f64p), _p(g, _f64p), _p(hi, _f32p), len(hi),
                                             _p(lo, _f32p), len(lo), _p(hist, _f32p), len(hist), C.c_float(min_score),
                                             _p(pose, _f64p), _p(out, _f64p))
        return self._result(pose, out)

    def MatchWith3DofInitial(self, pose_in_submap_guess, data, min_score):
        g, hi, lo, hist = self._data(data)
        pose, out = np.zeros(7), np.zeros(6)
        lib().orc_fast_csm_match_3dof(self.h, _p(_f64(pose_in_submap_guess), _f64p), _p(g, _f64p), _p(hi, _f32p), len(hi),
                                      _p(lo, _f32p), len(lo), _p(hist, _f32p), len(hist), C.c_float(min_score),
                                      _p(pose, _f64p), _p(out, _f64p))
        return self._result(pose, out)


# ------------------------------------------------------------------ IMU preintegration
class IntegrationBase:
    """mapping/internal/3d/initialization/integration_base.h restated (oracle/src/om_imu.h)."""

    def __init__(self, ba, bg, noise):
        self.h = lib().orc_imu_new(_p(_f64(ba), _f64p), _p(_f64(bg), _f64p), _p(_f64(noise), _f64p))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_imu_free(self.h)
            self.h = None

    def push_back(self, dt, acc, gyr):
        lib().orc_imu_push_back(self.h, float(dt), _p(_f64(acc), _f64p), _p(_f64(gyr), _f64p))

    def repropagate(self, ba, bg):
        lib().orc_imu_repropagate(self.h, _p(_f64(ba), _f64p), _p(_f64(bg), _f64p))

    def get(self):
        o = np.zeros(461)
        lib().orc_imu_get(self.h, _p(o, _f64p))
        return dict(sum_dt=o[0], delta_p=o[1:4].copy(), delta_q=o[4:8].copy(), delta_v=o[8:11].copy(),
                    jacobian=o[11:236].reshape(15, 15).copy(), covariance=o[236:461].reshape(15, 15).copy())

    def evaluate(self, state_i, state_j, gravity):
        r = np.zeros(15)
        lib().orc_imu_evaluate(self.h, _p(_f64(state_i), _f64p), _p(_f64(state_j), _f64p), _p(_f64(gravity), _f64p),
                               _p(r, _f64p))
        return r
