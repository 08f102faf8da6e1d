#!/usr/bin/env python3
"""Kernel micro-benchmark for fast iteration on the GPU box: times the RTCSM3D score-volume
kernel, the full match, one Ceres match and one insertion on the bench.py workload."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "d-liom_amd")):
    sys.path.insert(0, p)
import benchlib as bench  # noqa: E402  (constants + build_scene)
import dliom as dl  # noqa: E402
from dliom import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--beams", type=int, default=64)
    ap.add_argument("--azimuths", type=int, default=1024)
    ap.add_argument("--map-scans", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--max-range", type=float, default=0.0)
    ap.add_argument("--check", action="store_true", help="compare against the oracle on sampled candidates")
    a = ap.parse_args()
    ctx = dl.Context(0)
    ins = dl.RangeDataInserter3D(bench.HIT_P, bench.MISS_P, bench.FREE, ctx=ctx)
    g_hi, g_lo = dl.HybridGrid(ctx, 0.1), dl.HybridGrid(ctx, 0.45)
    for s in range(a.map_scans):
        pose = synth.trajectory_pose(0.1 * s)
        pts, _ = synth.scan(pose, a.beams, a.azimuths)
        c = dl.PointCloud(ctx, pts)
        pf = pose.astype(np.float32)
        ins.InsertCloud(g_hi, c, poses=[pf], max_range=bench.HIGH_RES_MAX_RANGE)
        ins.InsertCloud(g_lo, c, poses=[pf])
        c.close()
    truth = synth.trajectory_pose(0.1 * a.map_scans)
    pts, _ = synth.scan(truth, a.beams, a.azimuths)
    if a.max_range > 0:
        pts = synth.range_filter(pts, a.max_range)
    init = synth.perturb_pose(truth, 0.1, 0.5, seed=13)
    cloud = dl.PointCloud(ctx, pts)
    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, bench.RTCSM_OPTS)
    cs = dl.CeresScanMatcher3D(ctx, bench.CSM_OPTS)
    ctx.set_profiling(True)
    for name, fn in (("rtcsm", lambda: rt.Match(init, cloud, g_hi)),
                     ("ceres", lambda: cs.Match(init[:3], init, [(cloud, g_hi), (cloud, g_lo)])),
                     ("insert", lambda: ins.InsertCloud(g_lo, cloud, poses=[truth.astype(np.float32)]))):
        fn()
        ctx.reset_profiling()
        t = time.perf_counter()
        for _ in range(a.reps):
            r = fn()
        ctx.synchronize()
        wall = (time.perf_counter() - t) / a.reps * 1e3
        ks = {k: ctx.kernel_time(i) for k, i in (("score", 0), ("select", 1), ("rescore", 2), ("csm", 3), ("insert", 4))}
        msg = ", ".join("%s %.3f ms/%d" % (k, v[0] / a.reps, v[1] // a.reps) for k, v in ks.items() if v[1])
        print("%-7s wall %.3f ms | %s" % (name, wall, msg))
        if name == "rtcsm":
            st = rt.last_stats()
            C, n = st.window.num_candidates, st.num_points
            k_ms = ks["score"][0] / a.reps
            print("        C=%d N=%d rescored=%d  pairs/s=%.3e  alg GB/s=%.0f" %
                  (C, n, st.num_rescored, C * n / (k_ms * 1e-3), 14.0 * C * n / (k_ms * 1e-3) / 1e9))
            fn_stats = getattr(dl.load_library(), "dliom_exp_box_stats", None)
            if fn_stats is not None and (int(os.environ.get("DLIOM_BOX_DEBUG", "0")) & 128):
                import ctypes
                buf = (ctypes.c_uint32 * 8)()
                fn_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
                fn_stats(ctx.h, buf)
                names = ("boxes", "staged_quads", "box_points", "exact_points", "l1_rounds", "l1_entries", "l2_entries", "bbox_tries")
                print("        stats/launch: " + ", ".join("%s %.0f" % (k, v / (a.reps + 1.0)) for k, v in zip(names, buf)))
        if name == "ceres":
            print("        evals=%d iterations=%d" % (r[1]["num_residual_evaluations"], r[1]["num_iterations"]))
    if a.check:
        from oracle import oracle as orc
        og = orc.HybridGrid(0.1)
        origins, values = g_hi.download_blocks()
        for o, v in zip(origins, values):
            nz = np.nonzero(v)[0]
            if len(nz):
                og.set_values(np.stack([o[0] + (nz & 7), o[1] + ((nz >> 3) & 7), o[2] + (nz >> 6)], axis=1), v[nz])
        sums = rt.score_volume(init, pts, g_hi)
        rng = np.random.RandomState(0)
        for c in rng.randint(0, len(sums), size=int(os.environ.get('KBENCH_CHECK', '8'))):
            want = orc.rtcsm3d_value_sums(bench.RTCSM_OPTS, init, pts, og, first=int(c), count=1)[0]
            assert sums[c] == want, (c, sums[c], want)
        print("check ok: sampled candidates equal the oracle; box_error flags = %d" % rt.box_error())


if __name__ == "__main__":
    main()
