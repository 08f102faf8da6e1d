"""Latency of CeresScanMatcher3D::Match on the W-ref clouds against max_num_iterations, for the
one-launch Levenberg-Marquardt kernel and the launch-per-evaluation loop (slope = cost per iteration,
intercept = fixed cost)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "d-liom_amd"))
import dliom as dl  # noqa: E402
from dliom import synth  # noqa: E402


def main():
    ctx = dl.Context()
    truth = synth.trajectory_pose(0.3)
    ins = dl.RangeDataInserter3D(0.55, 0.49, 2, ctx=ctx)
    g_hi, g_lo = dl.HybridGrid(ctx, 0.1), dl.HybridGrid(ctx, 0.45)
    for s in range(6):
        pose = synth.trajectory_pose(0.025 * s)
        pts, _ = synth.scan(pose, 64, 1024)
        c = dl.PointCloud(ctx, pts)
        dl.insert_cloud_multi(ins, c, [(g_hi, [pose.astype(np.float32)], 20.0), (g_lo, [pose.astype(np.float32)], 0.0)])
        c.close()
    pts, _ = synth.scan(truth, 64, 1024)
    init = synth.perturb_pose(truth, 0.03, 0.2, seed=3)
    f = dl.PointCloud(ctx, pts).voxel_filter(0.15)
    hi = f.adaptive_voxel_filter(2.0, 150, 15.0)
    lo = f.adaptive_voxel_filter(4.0, 200, 60.0)
    out = {}
    for mode, env in (("one_launch", "4096"), ("per_evaluation", "0")):
        os.environ["DLIOM_CSM_PERSISTENT_MAX"] = env
        rows = []
        for iters in (0, 1, 2, 4, 8, 12, 24):
            cs = dl.CeresScanMatcher3D(ctx, dict(occupied_space_weight=[1.0, 6.0], translation_weight=5.0, rotation_weight=4e2,
                                                 only_optimize_yaw=False, use_nonmonotonic_steps=False, max_num_iterations=iters))
            ts = []
            for rep in range(30):
                a = time.perf_counter()
                pose, summ = cs.Match(init[:3], init, [(hi, g_hi), (lo, g_lo)])
                ts.append(time.perf_counter() - a)
            rows.append(dict(max_iters=iters, p50_us=1e6 * float(np.median(ts[5:])), evals=summ["num_residual_evaluations"],
                             iterations=summ["num_iterations"], pose=[float(v) for v in pose]))
        out[mode] = rows
    print(json.dumps(out))


if __name__ == "__main__":
    main()
