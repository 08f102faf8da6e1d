#!/usr/bin/env python3
"""Randomised differential test: device vs CPU oracle on many random (grid, cloud, pose, options)
cases -- score volumes, matches, insertion, voxel filters.  Exits non-zero on the first mismatch and
prints the seed.  Run on the GPU box: python tools/fuzz_parity.py --cases 150"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "d-liom_amd"))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--seconds", type=float, default=150.0)
    args = ap.parse_args(argv)
    import dliom as dl
    from dliom import synth
    from oracle import oracle as orc
    ctx = dl.Context(0)
    t_start = time.time()
    done = 0
    for case in range(args.cases):
        if time.time() - t_start > args.seconds:
            break
        seed = args.seed + case
        rng = np.random.RandomState(seed)
        res = float(rng.choice([0.05, 0.1, 0.2, 0.45]))
        extent = float(rng.choice([4.0, 10.0, 24.0])) * (res / 0.1) ** 0.5
        # a random occupied shell + clutter, inserted through both inserters
        og, dg = orc.HybridGrid(res), dl.HybridGrid(ctx, res)
        hit_p, miss_p, free = float(rng.uniform(0.52, 0.8)), float(rng.uniform(0.3, 0.49)), int(rng.randint(0, 4))
        ins = dl.RangeDataInserter3D(hit_p, miss_p, free)
        for s in range(int(rng.randint(1, 4))):
            n_ins = int(rng.randint(50, 4000))
            d = rng.normal(size=(n_ins, 3))
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            returns = (d * rng.uniform(0.3 * extent, extent, size=(n_ins, 1))).astype(np.float32)
            origin = rng.uniform(-0.1 * extent, 0.1 * extent, 3).astype(np.float32)
            og.insert_tables(origin, returns, ins.hit_table, ins.miss_table, free)
            ins.Insert(origin, returns, dg)
        xyz, v = og.export_cells()
        want = {(int(c[0]), int(c[1]), int(c[2])): int(val) for c, val in zip(xyz, v)}
        if dg.cells() != want:
            print("MISMATCH insertion seed", seed)
            return 1
        # a cloud near the shell, random pose, random window
        n = int(rng.choice([1, 5, 40, 167, 600, 3000]))
        d = rng.normal(size=(n, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        pts = (d * rng.uniform(0.2 * extent, 1.15 * extent, size=(n, 1))).astype(np.float32)
        init = np.concatenate([rng.uniform(-0.2 * extent, 0.2 * extent, 3),
                               synth.quat_from_axis_angle(rng.normal(size=3), rng.uniform(0, 3.1))])
        opts = dict(linear_search_window=float(rng.choice([0.0, 0.6, 1.5, 2.4])) * res,
                    angular_search_window=float(np.deg2rad(rng.uniform(0.1, 1.2))),
                    translation_delta_cost_weight=float(rng.uniform(0.01, 1.0)),
                    rotation_delta_cost_weight=float(rng.uniform(0.01, 1.0)))
        m = dl.RealTimeCorrelativeScanMatcher3D(ctx, opts)
        w = m.window(res, pts)
        if w.num_candidates * n > 4e7:  # keep the oracle side cheap
            dg.close()
            continue
        got = m.score_volume(init, pts, dg)
        ref = orc.rtcsm3d_value_sums(opts, init, pts, og)
        if not np.array_equal(got, ref):
            print("MISMATCH score volume seed", seed, "res", res, "n", n, "C", w.num_candidates)
            return 1
        score, pose = m.Match(init, pts, dg)
        r = orc.rtcsm3d_match(opts, init, pts, og)
        if not (np.array_equal(pose, r["pose"]) and np.float32(score) == np.float32(r["score"])):
            print("MISMATCH match seed", seed, score, r["score"], m.last_stats().best_index, r["best_index"])
            return 1
        # voxel filters on the same cloud
        size = float(rng.choice([0.05, 0.15, 0.5, 2.0]))
        cloud = dl.PointCloud(ctx, pts)
        f = cloud.voxel_filter(size)
        if not np.array_equal(f.download(), pts[orc.voxel_filter(size, pts)]):
            print("MISMATCH voxel filter seed", seed)
            return 1
        ao = (float(rng.choice([0.5, 2.0, 4.0])), float(rng.choice([5, 50, 150])), float(rng.uniform(0.5, 1.2) * extent))
        a = cloud.adaptive_voxel_filter(*ao)
        if not np.array_equal(a.download(), orc.adaptive_voxel_filter(ao[0], ao[1], ao[2], pts)):
            print("MISMATCH adaptive voxel filter seed", seed, ao)
            return 1
        # Ceres scan matcher on the same grid (two clouds on one grid), pose within 1e-6 of the oracle
        if n >= 40:
            copts = dict(occupied_space_weight=[1.0, float(rng.uniform(0.5, 6.0))], translation_weight=float(rng.uniform(0.1, 10)),
                         rotation_weight=float(rng.uniform(1, 400)), only_optimize_yaw=bool(rng.randint(0, 2)),
                         use_nonmonotonic_steps=bool(rng.randint(0, 2)), max_num_iterations=int(rng.randint(3, 15)))
            p2, _ = dl.CeresScanMatcher3D(ctx, copts).Match(init[:3], pose, [(pts, dg), (pts[::2], dg)])
            r2 = orc.csm3d_match(copts, init[:3], r["pose"], [(pts, og), (pts[::2], og)])
            if np.linalg.norm(p2[:3] - r2["pose"][:3]) > 1e-6 or np.abs(p2[3:] - r2["pose"][3:]).max() > 1e-6:
                print("MISMATCH ceres seed", seed, p2, r2["pose"])
                return 1
        f.close()
        a.close()
        cloud.close()
        dg.close()
        done += 1
    print("fuzz ok: %d cases in %.1f s" % (done, time.time() - t_start))
    return 0


if __name__ == "__main__":
    sys.exit(main())
