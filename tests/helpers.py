"""Shared scaffolding for the parity tests (scene building, pose helpers)."""
import numpy as np

from dliom import synth

DEFAULT_RTCSM = dict(linear_search_window=0.15, angular_search_window=np.deg2rad(1.0),
                     translation_delta_cost_weight=1e-1, rotation_delta_cost_weight=1e-1)
DEFAULT_CSM = dict(occupied_space_weight=[1.0, 6.0], translation_weight=5.0, rotation_weight=4e2,
                   only_optimize_yaw=False, use_nonmonotonic_steps=False, max_num_iterations=12)
HIT_P, MISS_P, FREE = 0.55, 0.49, 2


def build_oracle_submap(orc, resolution, num_scans=6, beams=16, azimuths=256, max_range=None,
                        first_scan=0):
    """Inserts `num_scans` scans at ground-truth corkscrew poses into an oracle HybridGrid whose
    frame is the world frame (submap pose = identity)."""
    g = orc.HybridGrid(resolution)
    hit = orc.lookup_table_to_apply_odds(orc.odds(HIT_P))
    miss = orc.lookup_table_to_apply_odds(orc.odds(MISS_P))
    for s in range(first_scan, first_scan + num_scans):
        pose = synth.trajectory_pose(0.1 * s)
        pts, _ = synth.scan(pose, beams, azimuths)
        if max_range is not None:
            pts = synth.range_filter(pts, max_range)
        world = synth.transform_points(pose, pts)
        g.insert_tables(pose[:3].astype(np.float32), world, hit, miss, FREE)
    return g


def to_device_grid(dl, ctx, oracle_grid):
    origins, leaves = oracle_grid.export_leaves()
    g = dl.HybridGrid(ctx, oracle_grid.resolution)
    g.upload_blocks(origins, leaves)
    return g


def oracle_cells_dict(oracle_grid):
    xyz, v = oracle_grid.export_cells()
    return {(int(c[0]), int(c[1]), int(c[2])): int(val) for c, val in zip(xyz, v)}


def pose_distance(a, b):
    """(translation distance, rotation angle) between two [t,q] poses."""
    dt = np.linalg.norm(np.asarray(a[:3]) - np.asarray(b[:3]))
    qa = np.asarray(a[3:]) / np.linalg.norm(a[3:])
    qb = np.asarray(b[3:]) / np.linalg.norm(b[3:])
    d = abs(float(np.dot(qa, qb)))
    return dt, 2.0 * np.arccos(min(1.0, d))


# ---- the benchmarked configurations (bench.py builds the same scenes with the same constants) --------
def build_device_scene(dl, ctx, beams, azimuths, res_hi, res_lo, map_scans, num_scans=1, high_res_max_range=20.0):
    """bench.py's scene: `map_scans` scans inserted at ground truth into a high and a low resolution
    device grid, then `num_scans` scans to match (truth, points, perturbed initial pose, device cloud)."""
    ins = dl.RangeDataInserter3D(HIT_P, MISS_P, FREE, ctx=ctx)
    g_hi, g_lo = dl.HybridGrid(ctx, res_hi), dl.HybridGrid(ctx, res_lo)
    centers = synth.bubbles()
    for s in range(map_scans):
        pose = synth.trajectory_pose(0.1 * s)
        pts, _ = synth.scan(pose, beams, azimuths, centers=centers)
        cloud = dl.PointCloud(ctx, pts)
        pf = pose.astype(np.float32)
        ins.InsertCloud(g_hi, cloud, poses=[pf], max_range=high_res_max_range)
        ins.InsertCloud(g_lo, cloud, poses=[pf])
        cloud.close()
    scans = []
    for k in range(num_scans):
        truth = synth.trajectory_pose(0.1 * (map_scans + k))
        pts, _ = synth.scan(truth, beams, azimuths, centers=centers)
        init = synth.perturb_pose(truth, 0.1, 0.5, seed=13 + k)
        scans.append(dict(truth=truth, pts=pts, init=init, cloud=dl.PointCloud(ctx, pts)))
    return ins, g_hi, g_lo, scans


def device_grid_to_oracle(orc, dg):
    """Oracle HybridGrid with exactly the device grid's non-zero cells (one vectorised upload)."""
    og = orc.HybridGrid(dg.resolution)
    origins, values = dg.download_blocks()
    leaf, cell = np.nonzero(values)
    if len(leaf):
        xyz = np.stack([origins[leaf, 0] + (cell & 7), origins[leaf, 1] + ((cell >> 3) & 7), origins[leaf, 2] + (cell >> 6)],
                       axis=1).astype(np.int32)
        og.set_values(xyz, values[leaf, cell])
    return og


def device_cells_sorted(dg):
    """(keys, values) of the device grid's non-zero cells, sorted by a 63-bit cell key."""
    origins, values = dg.download_blocks()
    leaf, cell = np.nonzero(values)
    x = origins[leaf, 0].astype(np.int64) + (cell & 7)
    y = origins[leaf, 1].astype(np.int64) + ((cell >> 3) & 7)
    z = origins[leaf, 2].astype(np.int64) + (cell >> 6)
    key = ((x + (1 << 20)) << 42) | ((y + (1 << 20)) << 21) | (z + (1 << 20))
    order = np.argsort(key)
    return key[order], values[leaf, cell][order]


def oracle_cells_sorted(og):
    xyz, v = og.export_cells()
    xyz = xyz.astype(np.int64)
    key = ((xyz[:, 0] + (1 << 20)) << 42) | ((xyz[:, 1] + (1 << 20)) << 21) | (xyz[:, 2] + (1 << 20))
    order = np.argsort(key)
    return key[order], np.asarray(v)[order]


def slice_angle_arrays(pts):
    """The angle arrays SortSlice hands to std::sort for a cloud (rotational_scan_matcher.cc:97-121,159-165), in exact
    float32 arithmetic: slices by lround(z / 0.2f) in input order, sequential float centroid, points closer than 0.2 m
    to it skipped, atan2f of this machine's libm."""
    pts = np.asarray(pts, np.float32)
    q = (pts[:, 2] / np.float32(0.2)).astype(np.float64)
    keys = np.where(q >= 0, np.floor(q + 0.5), -np.floor(-q + 0.5)).astype(int)
    out = []
    for key in np.unique(keys):
        s = pts[keys == key]
        c = np.zeros(3, np.float32)
        for p in s:
            c = (c + p).astype(np.float32)
        c = (c / np.float32(len(s))).astype(np.float32)
        d = (s[:, :2] - c[:2]).astype(np.float32)
        nrm = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32)).astype(np.float32)
        ang = np.arctan2(d[~(nrm < np.float32(0.2)), 1], d[~(nrm < np.float32(0.2)), 0]).astype(np.float32)
        if len(ang) > 1:
            out.append(ang)
    return out
