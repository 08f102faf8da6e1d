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
