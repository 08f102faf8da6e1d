#!/usr/bin/env python3
"""Loop-closure matcher (FastCorrelativeScanMatcher3D) timing: device vs the CPU oracle on a synthetic
submap with D-LIOM-like options (basic_config_3d.lua:125-135: 0.2 m submaps, depth 8 / full-resolution
depth 3, 15 m x 8 m x 45 deg window).  One JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "d-liom_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dense", action="store_true", help="match every return of a 64x1024 scan instead of the filtered cloud")
    ap.add_argument("--xy", type=float, default=15.0)
    ap.add_argument("--z", type=float, default=8.0)
    ap.add_argument("--angle", type=float, default=45.0)
    ap.add_argument("--full", action="store_true", help="MatchFullSubmap (window = whole submap, 360 deg)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import dliom as dl
    from dliom import synth
    from helpers import build_oracle_submap, to_device_grid
    from oracle import oracle as orc

    ctx = dl.Context(0)
    og_hi = build_oracle_submap(orc, 0.2, num_scans=10, beams=32, azimuths=512, max_range=60.0)
    og_lo = build_oracle_submap(orc, 0.5, num_scans=10, beams=32, azimuths=512)
    g_hi, g_lo = to_device_grid(dl, ctx, og_hi), to_device_grid(dl, ctx, og_lo)
    opts = dict(branch_and_bound_depth=8, full_resolution_depth=3, min_rotational_score=0.77,
                min_low_resolution_score=0.55, linear_xy_search_window=args.xy, linear_z_search_window=args.z,
                angular_search_window=np.deg2rad(args.angle))
    hists, yaws = [], []
    for s in range(10):
        pose = synth.trajectory_pose(0.1 * s)
        pts, _ = synth.scan(pose, 32, 512)
        hists.append(orc.compute_histogram(pts, 120))
        yaws.append(float(np.arctan2(2 * (pose[3] * pose[6] + pose[4] * pose[5]), 1 - 2 * (pose[5] ** 2 + pose[6] ** 2))))
    t0 = time.perf_counter()
    dm = dl.FastCorrelativeScanMatcher3D(ctx, g_hi, g_lo, np.array(hists), yaws, opts)
    ctx.synchronize()
    t_build = time.perf_counter() - t0
    truth = synth.trajectory_pose(0.45)
    pts, _ = synth.scan(truth, 64, 1024)
    hi_pts = pts if args.dense else orc.adaptive_voxel_filter(2.0, 150, 15.0, pts)
    lo_pts = orc.adaptive_voxel_filter(4.0, 200, 60.0, pts)
    data = dict(gravity_alignment=[1, 0, 0, 0], high_resolution_point_cloud=hi_pts, low_resolution_point_cloud=lo_pts,
                rotational_scan_matcher_histogram=orc.compute_histogram(pts, 120))
    node_pose = synth.perturb_pose(truth, 4.0, 15.0, seed=9)
    ident = np.array([0, 0, 0, 1.0, 0, 0, 0])
    times = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        rd = dm.MatchFullSubmap(node_pose[3:], ident[3:], data, 0.55) if args.full else dm.Match(node_pose, ident, data, 0.55)
        times.append(time.perf_counter() - t0)
    out = {"mode": "MatchFullSubmap" if args.full else "Match",
           "workload": "FastCorrelativeScanMatcher3D::Match, 0.2 m submap of 10 scans, window %.0f m x %.0f m x %.0f deg, "
                       "depth 8 / full-resolution depth 3" % (args.xy, args.z, args.angle),
           "N_hi": len(hi_pts), "N_lo": len(lo_pts), "pyramid_build_ms": 1e3 * t_build,
           "device": {"match_ms_p50": 1e3 * float(np.median(times)), "found": rd["found"], "score": float(rd["score"]),
                      "discrete_scans": rd["num_discrete_scans"], "scored_candidates": rd["num_scored_candidates"],
                      "score_launches": rd["num_score_launches"]}}
    if not args.no_cpu:
        t0 = time.perf_counter()
        om = orc.FastCorrelativeScanMatcher3D(og_hi, og_lo, np.array(hists), yaws, opts)
        t_build_cpu = time.perf_counter() - t0
        t0 = time.perf_counter()
        ro = om.MatchFullSubmap(node_pose[3:], ident[3:], data, 0.55) if args.full else om.Match(node_pose, ident, data, 0.55)
        t_cpu = time.perf_counter() - t0
        same = ro["found"] == rd["found"] and (not ro["found"] or (np.float32(ro["score"]) == np.float32(rd["score"]) and
                                                                  np.array_equal(ro["pose"], rd["pose"])))
        out["cpu_oracle_1_thread"] = {"match_ms": 1e3 * t_cpu, "pyramid_build_ms": 1e3 * t_build_cpu,
                                      "scored_candidates": ro["num_scored_candidates"], "identical_result": bool(same)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
