#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

The reference is C++ behind Eigen/Ceres/glog and cannot be built or imported in this environment
(SURVEY.md 8c), so the fixtures are produced by the oracle -- the restatement that is itself pinned
by every known-answer test the reference holds for the path (tests/test_oracle_kat.py).  They
freeze complete input/output pairs of each stage on seeded synthetic scans so that
  * tests/test_golden.py (not gpu) detects any drift of the oracle, and
  * tests/test_golden.py (gpu) checks the device path against files that do not depend on the
    oracle library being rebuilt the same way on the GPU box.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "d-liom_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from dliom import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from helpers import DEFAULT_CSM, DEFAULT_RTCSM, FREE, HIT_P, MISS_P, build_oracle_submap  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def cells(g):
    xyz, v = g.export_cells()
    order = np.lexsort((xyz[:, 0], xyz[:, 1], xyz[:, 2]))
    return xyz[order].astype(np.int32), v[order].astype(np.uint16)


def main():
    # 1. insertion: three 16x128 scans at ground-truth poses into a 10 cm grid
    g = orc.HybridGrid(0.1)
    hit = orc.lookup_table_to_apply_odds(orc.odds(HIT_P))
    miss = orc.lookup_table_to_apply_odds(orc.odds(MISS_P))
    origins, returns = [], []
    for s in range(3):
        pose = synth.trajectory_pose(0.1 * s)
        pts, _ = synth.scan(pose, 16, 128)
        world = synth.transform_points(pose, pts)
        g.insert_tables(pose[:3].astype(np.float32), world, hit, miss, FREE)
        origins.append(pose[:3].astype(np.float32))
        returns.append(world)
    cx, cv = cells(g)
    np.savez_compressed(os.path.join(OUT, "insertion.npz"), resolution=np.float32(0.1), hit_probability=HIT_P,
                        miss_probability=MISS_P, num_free_space_voxels=FREE, origins=np.array(origins),
                        returns=np.array(returns), cell_xyz=cx, cell_value=cv)

    # 2. RTCSM3D + Ceres on a 6-scan submap (hi 10 cm / lo 45 cm), 16x256 query scan
    g_hi = build_oracle_submap(orc, 0.1, num_scans=6, max_range=20.0)
    g_lo = build_oracle_submap(orc, 0.45, num_scans=6)
    truth = synth.trajectory_pose(0.6)
    pts, _ = synth.scan(truth, 16, 256)
    init = synth.perturb_pose(truth, 0.1, 0.5, seed=21)
    r = orc.rtcsm3d_match(DEFAULT_RTCSM, init, pts, g_hi)
    sums = orc.rtcsm3d_value_sums(DEFAULT_RTCSM, init, pts, g_hi)
    c = orc.csm3d_match(DEFAULT_CSM, init[:3], r["pose"], [(pts, g_hi), (pts, g_lo)])
    hx, hv = cells(g_hi)
    lx, lv = cells(g_lo)
    np.savez_compressed(os.path.join(OUT, "matching.npz"), points=pts, initial_pose=init, hi_cell_xyz=hx,
                        hi_cell_value=hv, lo_cell_xyz=lx, lo_cell_value=lv, rtcsm_pose=r["pose"],
                        rtcsm_score=np.float32(r["score"]), score_volume_sums=np.asarray(sums, dtype=np.uint64),
                        csm_pose=c["pose"], csm_final_cost=c["final_cost"], csm_iterations=c["num_iterations"])

    # 3. voxel filters on a 32x512 scan
    pts, _ = synth.scan(synth.trajectory_pose(0.3), 32, 512)
    np.savez_compressed(os.path.join(OUT, "voxel_filter.npz"), points=pts,
                        kept_015=orc.voxel_filter(0.15, pts).astype(np.int32),
                        adaptive_hi=orc.adaptive_voxel_filter(2.0, 150, 15.0, pts),
                        adaptive_lo=orc.adaptive_voxel_filter(4.0, 200, 60.0, pts))
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
