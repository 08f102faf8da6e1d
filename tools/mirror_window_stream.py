#!/usr/bin/env python3
"""A moving sensor on a grid beyond bits = 4: how often is the windowed dense mirror of the
correlative matcher rebuilt, and what does a rebuild cost?

Yard scene (dliom.synth "ground": returns to 80 m), a 10 cm HybridGrid that every scan is inserted into without a range
cut (DynamicGrid bits 5), the sensor driving straight at `--speed` m per scan for `--scans` scans.  Every scan is
matched (RTCSM3D, the scan cut at `--cut` m so that its search cube fits the window's 1264 cells) and inserted.
dliom_grid_mirror_stats counts the (re)builds; a match that rebuilt the window is timed apart from one that reused it.
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "d-liom_amd")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=40)
    ap.add_argument("--speed", type=float, default=1.0, help="metres per scan (10 Hz: 1.0 = 36 km/h)")
    ap.add_argument("--cut", type=float, default=40.0)
    ap.add_argument("--resolution", type=float, default=0.10)
    a = ap.parse_args()
    import dliom as dl
    from dliom import synth
    dl.load_library()
    ctx = dl.Context(0)
    ins = dl.RangeDataInserter3D(0.55, 0.49, 2, ctx=ctx)
    grid = dl.HybridGrid(ctx, a.resolution)
    opts = dict(linear_search_window=0.15, angular_search_window=float(np.deg2rad(0.35)),
                translation_delta_cost_weight=1e-1, rotation_delta_cost_weight=1e-1)
    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, opts)

    def pose_at(k):
        return np.array([a.speed * k, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])

    rows = []
    with synth.scene("ground"):
        for k in range(3):  # a submap to match against
            pts, _ = synth.scan(pose_at(k), 32, 512)
            c = dl.PointCloud(ctx, pts)
            ins.InsertCloud(grid, c, poses=[pose_at(k).astype(np.float32)])
            c.close()
        for k in range(3, 3 + a.scans):
            truth = pose_at(k)
            pts, _ = synth.scan(truth, 32, 512)
            near = pts[np.linalg.norm(pts.astype(np.float64), axis=1) <= a.cut]
            init = synth.perturb_pose(truth, 0.05, 0.2, seed=k)
            cloud = dl.PointCloud(ctx, near)
            before = grid.mirror_stats()[0]
            ctx.synchronize()
            t0 = time.perf_counter()
            score, pose = rt.Match(init, cloud, grid)
            ctx.synchronize()
            ms = 1e3 * (time.perf_counter() - t0)
            rebuilds, nbytes, windowed = grid.mirror_stats()
            st = rt.last_stats()
            rows.append(dict(scan=k, ms=ms, rebuilt=rebuilds > before, score_kernel=int(st.score_kernel), bits=int(grid.bits),
                             err_m=float(np.linalg.norm(pose[:3] - truth[:3]))))
            cloud.close()
            full = dl.PointCloud(ctx, pts)
            ins.InsertCloud(grid, full, poses=[pose.astype(np.float32)])  # written through to the cells inside the window
            full.close()
    rebuilt = [r["ms"] for r in rows[1:] if r["rebuilt"]]
    reused = [r["ms"] for r in rows[1:] if not r["rebuilt"]]
    rebuilds, nbytes, windowed = grid.mirror_stats()
    out = {"workload": "yard scene, %g cm grid, returns inserted to 80 m, sensor moving %.2f m per scan, scans cut at %g m for matching"
                       % (100 * a.resolution, a.speed, a.cut),
           "scans": len(rows), "grid_bits": int(grid.bits), "mirror_windowed": bool(windowed), "mirror_bytes": int(nbytes),
           "mirror_builds_total": int(rebuilds), "scans_per_rebuild": (len(rows) / max(1, rebuilds - 1)) if rebuilds > 1 else None,
           "match_ms_window_reused_p50": float(np.median(reused)) if reused else None,
           "match_ms_window_rebuilt_p50": float(np.median(rebuilt)) if rebuilt else None,
           "rebuild_cost_ms": (float(np.median(rebuilt)) - float(np.median(reused))) if rebuilt and reused else None,
           "amortised_ms_per_scan": float(np.mean([r["ms"] for r in rows[1:]])),
           "box_kernel_on_every_match": all(r["score_kernel"] == 3 for r in rows),
           "max_pose_error_m": max(r["err_m"] for r in rows), "box_kernel_flags": int(rt.box_error())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
