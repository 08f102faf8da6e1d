#!/usr/bin/env python3
"""Randomised differential test of what round 3 put on the device: ComputeHistogram (random scans, rotations, crops,
duplicated and collinear points), the std::sort order it rests on, AddRangeData (random motion, ranges, filter sizes)
and the box score kernel on ragged cloud sizes -- device vs CPU oracle, bit for bit.  Exits non-zero on the first
mismatch and prints the case.  Run on the GPU box: python tools/fuzz_round3.py --seconds 120"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "d-liom_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--cases", type=int, default=0, help="stop after this many cases (0: run for --seconds)")
    ap.add_argument("--big-sorts", action="store_true", help="kind 1: every array above 4096 keys (the big slices' sort), up to 40 000")
    ap.add_argument("--kinds", default="0,1,2", help="which case kinds run (0 histogram, 1 std::sort order, 2 AddRangeData); the others are skipped")
    args = ap.parse_args(argv)
    import dliom as dl
    from dliom import synth
    from oracle import oracle as orc
    ctx = dl.Context(0)
    t0 = time.time()
    counts = {"histogram": 0, "sort": 0, "add_range_data": 0}
    case = 0
    kinds = {int(k) for k in args.kinds.split(",")}
    while time.time() - t0 < args.seconds and (args.cases <= 0 or case < args.cases):
        seed = args.seed + case
        rng = np.random.RandomState(seed)
        kind = case % 3
        case += 1
        if kind not in kinds:
            continue
        family = "ground" if rng.rand() < 0.4 else "cube"  # round 4: the yard scene (a floor, ragged scans, far returns)
        if kind == 0:  # ComputeHistogram
            beams, az = int(rng.choice([16, 32, 64])), int(rng.choice([256, 512, 1024]))
            with synth.scene(family):
                raw, _ = synth.scan(synth.trajectory_pose(float(rng.uniform(0, 3))), beams, az,
                                    noise_sigma=0.02 if rng.rand() < 0.3 else 0.0, noise_seed=seed)
            size = float(rng.choice([0.1, 0.15, 0.3]))
            pts = raw[orc.voxel_filter(size, raw)]
            mode = int(rng.randint(0, 4))
            if mode == 1:  # a crop: few slices, small ones
                pts = pts[np.abs(pts[:, 2]) < rng.uniform(0.3, 2.0)]
            elif mode == 2:  # duplicated returns
                pts = np.concatenate([pts, pts[rng.randint(0, len(pts), len(pts) // 3)]])
            elif mode == 3:  # returns on common rays: equal angles everywhere
                k = int(rng.randint(20, 200))
                ang = np.repeat(rng.uniform(0, 2 * np.pi, k), 6)
                rad = np.tile(rng.uniform(2, 20, 6), k)
                z = np.tile(np.repeat(rng.uniform(-1, 1, 2), 3), k)
                extra = np.stack([rad * np.cos(ang), rad * np.sin(ang), z], axis=1).astype(np.float32)
                pts = np.concatenate([pts[: len(pts) // 4], extra]).astype(np.float32)
            pts = pts[rng.permutation(len(pts))] if rng.rand() < 0.3 else pts
            if len(pts) == 0:
                continue
            rot = None
            if rng.rand() < 0.7:
                rot = synth.perturb_pose(np.array([0, 0, 0, 1, 0, 0, 0], float), 0.0, float(rng.uniform(0, 10)), seed=seed)[3:].astype(np.float32)
            hsize = int(rng.choice([1, 17, 120, 255]))
            cloud = dl.PointCloud(ctx, pts)
            try:
                got = dl.cloud_rotational_histogram(ctx, cloud, hsize, rotation_wxyz=rot)
            except dl.DliomError as e:
                print("REFUSED histogram seed %d mode %d n %d status %d (slices of any size are in contract since round 4)" %
                      (seed, mode, len(pts), e.status))
                return 1
            aligned = pts if rot is None else orc.transform_points(np.concatenate([np.zeros(3, np.float32), rot]), pts)
            want = np.asarray(orc.compute_histogram(aligned, hsize), np.float32)
            if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                print("MISMATCH histogram seed %d mode %d n %d size %d: %d buckets, max %g" % (seed, mode, len(pts), hsize, int((got != want).sum()), float(np.abs(got - want).max())))
                return 1
            # ... and addition by addition: a bucket whose sum is in the hundreds hides a contribution of 1e-5 gone astray
            gb, gv = dl.diag_histogram_contributions(ctx, cloud, hsize, rotation_wxyz=rot)
            wb, wv = orc.histogram_contributions(aligned, hsize)
            if len(gb) != len(wb) or not np.array_equal(gb, wb) or not np.array_equal(gv.view(np.uint32), wv.view(np.uint32)):
                print("MISMATCH contributions seed %d mode %d n %d size %d: %d on the device, %d in the oracle" % (seed, mode, len(pts), hsize, len(gb), len(wb)))
                return 1
            cloud.close()
            counts["histogram"] += 1
        elif kind == 1:  # std::sort's order
            n = int(rng.randint(1, 4097)) if rng.rand() < 0.85 else int(rng.randint(4097, 30000))  # above 4096: the HBM path
            if args.big_sorts:
                n = int(rng.randint(4097, 40000))
            choice = int(rng.randint(0, 5))
            if choice == 0:
                keys = rng.randint(0, max(1, n // int(rng.randint(1, 40))), n)
            elif choice == 1:
                keys = np.sort(rng.uniform(-3, 3, n))
                keys[rng.randint(0, n, n // 5)] = keys[rng.randint(0, n, n // 5)]
            elif choice == 2:
                keys = np.arange(n) % int(rng.randint(1, 70))
            elif choice == 3:
                keys = -np.arange(n) // int(rng.randint(1, 9))
            else:
                keys = rng.uniform(-3.2, 3.2, n)
            keys = keys.astype(np.float32)
            try:
                got = dl.diag_std_sort_order(ctx, keys)
            except dl.DliomError as e:
                print("REFUSED sort seed %d n %d choice %d status %d" % (seed, n, choice, e.status))
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.save(os.path.join(ROOT, "gpurun_out", "refused_sort_keys.npy"), keys)
                return 1
            if not np.array_equal(got, orc.std_sort_order(keys)):
                print("MISMATCH sort seed %d n %d choice %d" % (seed, n, choice))
                return 1
            counts["sort"] += 1
        else:  # AddRangeData
            beams, az = int(rng.choice([16, 64])), int(rng.choice([256, 1024]))
            t_end = float(rng.uniform(0.2, 3.0))
            T = 0.1
            with synth.scene(family):
                prev = synth.trajectory_pose(t_end - T)
                cur = synth.perturb_pose(synth.trajectory_pose(t_end), float(rng.uniform(0, 0.05)), float(rng.uniform(0, 1.0)), seed=seed)
                pts, t_rel = synth.scan(synth.trajectory_pose(t_end), beams, az)
            xyzt = np.concatenate([pts, t_rel.reshape(-1, 1)], axis=1).astype(np.float32)
            vfs = float(rng.choice([0.1, 0.15, 0.3]))
            rmin, rmax = float(rng.uniform(0.5, 3.0)), float(rng.uniform(15.0, 120.0))
            cloud, origin, cpose = dl.add_range_data(ctx, prev, cur, T, xyzt, (0, 0, 0), rmin, rmax, vfs)
            ref = orc.deskew_and_filter(T, rmin, rmax, vfs, prev, cur, xyzt)
            got = cloud.download()
            want_pts = np.asarray(ref["returns_in_tracking"], np.float32)
            ok = (got.shape == want_pts.shape and np.array_equal(got.view(np.uint32), want_pts.view(np.uint32)) and
                  np.array_equal(np.asarray(cpose, np.float32).view(np.uint32), np.asarray(ref["current_pose"], np.float32).view(np.uint32)))
            if not ok:
                print("MISMATCH add_range_data seed %d n %d vfs %g: %d vs %d returns" % (seed, len(xyzt), vfs, len(got), len(ref["returns_in_tracking"])))
                return 1
            cloud.close()
            counts["add_range_data"] += 1
    print("round-3 fuzz ok: %s in %.1f s" % (counts, time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
