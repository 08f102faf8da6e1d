#!/usr/bin/env python3
"""Randomised differential test of the loop-closure matcher (device vs CPU oracle): random submaps,
pyramid depths, windows, thresholds, node poses, Match / MatchWith3DofInitial (MatchFullSubmap's
whole-submap windows are covered at known-answer-test scale in tests/test_gpu_fast_csm.py)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "d-liom_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=4321)
    ap.add_argument("--seconds", type=float, default=120.0)
    args = ap.parse_args(argv)
    import dliom as dl
    from dliom import synth
    from helpers import build_oracle_submap, to_device_grid
    from oracle import oracle as orc
    ctx = dl.Context(0)
    t0 = time.time()
    done = 0
    ident = np.array([0, 0, 0, 1.0, 0, 0, 0])
    for case in range(args.cases):
        if time.time() - t0 > args.seconds:
            break
        seed = args.seed + case
        rng = np.random.RandomState(seed)
        res = float(rng.choice([0.1, 0.2, 0.3]))
        scans = int(rng.randint(2, 6))
        synth.set_scene("ground" if rng.rand() < 0.35 else "cube")  # round 4: also the yard scene (reset at the end of the case)
        og_hi = build_oracle_submap(orc, res, num_scans=scans, beams=8, azimuths=128, max_range=30.0)
        og_lo = build_oracle_submap(orc, 0.5, num_scans=scans, beams=8, azimuths=128)
        g_hi, g_lo = to_device_grid(dl, ctx, og_hi), to_device_grid(dl, ctx, og_lo)
        depth = int(rng.randint(1, 8))
        opts = dict(branch_and_bound_depth=depth, full_resolution_depth=int(rng.randint(1, depth + 1)),
                    min_rotational_score=float(rng.uniform(0.0, 0.8)), min_low_resolution_score=float(rng.uniform(0.1, 0.5)),
                    linear_xy_search_window=float(rng.uniform(0.3, 4.0)), linear_z_search_window=float(rng.uniform(0.2, 1.5)),
                    angular_search_window=float(np.deg2rad(rng.uniform(1.0, 40.0))))
        hsize = int(rng.choice([10, 30, 120]))
        hists, yaws = [], []
        for s in range(scans):
            pose = synth.trajectory_pose(0.1 * s)
            pts, _ = synth.scan(pose, 8, 128)
            hists.append(orc.compute_histogram(pts, hsize))
            yaws.append(float(rng.uniform(-0.2, 0.2)))
        # keep the search small enough for the CPU oracle (and the box's RAM): bound the number of
        # lowest-resolution candidates the reference itself would allocate
        step = 1 << (depth - 1)
        per_xy = (2 * round(opts["linear_xy_search_window"] / res) + step) // step
        per_z = (2 * round(opts["linear_z_search_window"] / res) + step) // step
        max_scans = 2 * opts["angular_search_window"] / (res / 30.0) + 1
        if per_xy * per_xy * per_z * max_scans > 2e6:
            g_hi.close()
            g_lo.close()
            synth.set_scene("cube")
            continue
        om = orc.FastCorrelativeScanMatcher3D(og_hi, og_lo, np.array(hists), yaws, opts)
        dm = dl.FastCorrelativeScanMatcher3D(ctx, g_hi, g_lo, np.array(hists), yaws, opts)
        truth = synth.trajectory_pose(0.1 * float(rng.uniform(0, scans)))
        pts, _ = synth.scan(truth, 8, 128)
        n_hi = int(rng.choice([30, 150, len(pts)]))
        hi_pts = pts[rng.choice(len(pts), n_hi, replace=False)] if n_hi < len(pts) else pts
        lo_pts = pts[::5]
        g = synth.quat_from_axis_angle(rng.normal(size=3), rng.uniform(0, 0.05))
        data = dict(gravity_alignment=g, high_resolution_point_cloud=hi_pts, low_resolution_point_cloud=lo_pts,
                    rotational_scan_matcher_histogram=orc.compute_histogram(pts, hsize))
        node = synth.perturb_pose(truth, float(rng.uniform(0, 2.0)), float(rng.uniform(0, 10.0)), seed=seed)
        submap = synth.perturb_pose(ident, 0.3, 3.0, seed=seed + 7)
        min_score = float(rng.uniform(0.1, 0.6))
        pairs = [(dm.Match(node, submap, data, min_score), om.Match(node, submap, data, min_score), "Match"),
                 (dm.MatchWith3DofInitial(node, data, min_score), om.MatchWith3DofInitial(node, data, min_score), "3dof")]
        for rd, ro, what in pairs:
            same = rd["found"] == ro["found"] and rd["num_discrete_scans"] == ro["num_discrete_scans"]
            if same and ro["found"]:
                same = (np.float32(rd["score"]) == np.float32(ro["score"]) and np.array_equal(rd["pose"], ro["pose"]) and
                        np.float32(rd["low_resolution_score"]) == np.float32(ro["low_resolution_score"]) and
                        np.float32(rd["rotational_score"]) == np.float32(ro["rotational_score"]))
            if not same:
                print("MISMATCH", what, "seed", seed, opts, rd, ro)
                synth.set_scene("cube")
                return 1
        dm.close()
        g_hi.close()
        g_lo.close()
        synth.set_scene("cube")
        done += 1
    print("fast csm fuzz ok: %d cases in %.1f s" % (done, time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
