#!/usr/bin/env python3
"""Randomised differential test of the LDS-box score kernel's instantiations (round 6: 27 translations per pass at four and
at three waves per SIMD, 49 per pass) against the oracle's FULL candidate loop: random resolutions, linear windows of 1 to
4 cells (27 ... 729 translations), angular windows of one or two steps, random initial orientations, clouds of
20 000 ... 70 000 returns on random shells -- every candidate's integer sum, the winner's index, score bits and pose.
Exits non-zero on the first mismatch and prints the seed.  python tools/fuzz_box_variants.py --cases 40"""
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
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=6000)
    ap.add_argument("--seconds", type=float, default=180.0)
    args = ap.parse_args(argv)
    import dliom as dl
    from dliom import synth
    from oracle import oracle as orc
    ctx = dl.Context(0)
    threads = min(32, os.cpu_count() or 1)
    t_start = time.time()
    done, seen = 0, {0: 0, 1: 0, 2: 0}
    for case in range(args.cases):
        if time.time() - t_start > args.seconds:
            break
        seed = args.seed + case
        rng = np.random.RandomState(seed)
        res = float(rng.choice([0.05, 0.1, 0.2]))
        extent = float(rng.uniform(6.0, 22.0))
        og, dg = orc.HybridGrid(res), dl.HybridGrid(ctx, res)
        ins = dl.RangeDataInserter3D(0.55, 0.49, 2)
        centre = rng.uniform(-2.0, 2.0, 3)
        for s in range(2):  # an occupied shell with clutter: what the scan is matched against
            d = rng.normal(size=(20000, 3))
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            returns = (centre + d * rng.uniform(0.5 * extent, extent, size=(len(d), 1))).astype(np.float32)
            origin = (centre + rng.uniform(-0.3, 0.3, 3)).astype(np.float32)
            og.insert_tables(origin, returns, ins.hit_table, ins.miss_table, 2)
            ins.Insert(origin, returns, dg)
        n = int(rng.choice([20000, 32768, 50000, 70000]))
        d = rng.normal(size=(n, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        pts = (d * rng.uniform(0.45 * extent, 1.05 * extent, size=(n, 1))).astype(np.float32)  # in the sensor frame
        init = np.concatenate([centre + rng.uniform(-0.2, 0.2, 3), synth.quat_from_axis_angle(rng.normal(size=3), rng.uniform(0, 3.1))])
        cells = float(rng.choice([1.4, 2.4, 3.4, 4.4]))  # lround -> 1 .. 4 cells: 27, 125, 343, 729 translations
        opts = dict(linear_search_window=cells * res, angular_search_window=0.0,
                    translation_delta_cost_weight=float(rng.uniform(0.01, 1.0)), rotation_delta_cost_weight=float(rng.uniform(0.01, 1.0)))
        m0 = dl.RealTimeCorrelativeScanMatcher3D(ctx, dict(opts, angular_search_window=1.0))
        step = float(m0.window(res, pts).angular_step_size)
        opts["angular_search_window"] = step * float(rng.choice([1.2, 2.2]))  # one or two steps: 27 or 125 rotations
        m = dl.RealTimeCorrelativeScanMatcher3D(ctx, opts)
        w = m.window(res, pts)
        if float(w.num_candidates) * n > 3.0e9:
            dg.close()
            continue
        score, pose = m.Match(init, pts, dg)
        st = m.last_stats()
        got = m.score_volume(init, pts, dg)
        flat = orc.FlatGridIndex(og)
        want_sums, want_scores = orc.rtcsm3d_volume_fair(opts, init, pts, flat, threads=threads)
        tag = "seed %d res %g n %d T %d R %d kernel %d variant %d" % (seed, res, n, w.num_translations, w.num_rotations,
                                                                       st.score_kernel, st.box_kernel_variant)
        if not np.array_equal(got.astype(np.uint64), want_sums):
            print("MISMATCH score volume", tag, "differing", int(np.count_nonzero(got.astype(np.uint64) != want_sums)))
            return 1
        best = int(np.argmax(want_scores))
        _, ca = orc.rtcsm3d_candidates(opts, res, pts, init)
        if st.best_index != best or np.float32(score).tobytes() != want_scores[best].tobytes() or not np.array_equal(pose, ca[best].astype(np.float64)):
            print("MISMATCH match", tag, st.best_index, best)
            return 1
        if m.box_error() != 0:
            print("box kernel flags", tag)
            return 1
        if st.score_kernel == 3:
            seen[int(st.box_kernel_variant)] += 1
        dg.close()
        done += 1
    print("box variant fuzz ok: %d cases in %.1f s, box-kernel instantiations seen %s" % (done, time.time() - t_start, seen))
    return 0


if __name__ == "__main__":
    sys.exit(main())
