"""Round-4 diagnosis: the yard 128 x 2048 scan without range noise (floor slice of 20 090 returns) whose device histogram
differed from the oracle's in tools/hist_bench.py --check.  Narrows it down: whole cloud / floor slice alone / its prefixes,
the slice's std::sort order and its sequential sums on their own.  Prints JSON lines."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "d-liom_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import dliom as dl
    from dliom import synth
    from oracle import oracle as orc
    from helpers import slice_angle_arrays
    ctx = dl.Context()
    with synth.scene("ground"):
        raw, _ = synth.scan(synth.trajectory_pose(0.5), 128, 2048)
    pts = raw[orc.voxel_filter(0.15, raw)]
    keys = np.round(pts[:, 2].astype(np.float64) / 0.2)
    u, c = np.unique(keys, return_counts=True)
    floor = pts[keys == u[np.argmax(c)]]
    rest = pts[keys != u[np.argmax(c)]]

    def compare(name, cloud_pts, reps=3):
        want = np.asarray(orc.compute_histogram(cloud_pts, 120), np.float32)
        cloud = dl.PointCloud(ctx, cloud_pts)
        rec = {"case": name, "points": int(len(cloud_pts)), "runs": []}
        for _ in range(reps):
            try:
                got = dl.cloud_rotational_histogram(ctx, cloud, 120)
            except Exception as e:  # noqa: BLE001
                rec["runs"].append("error: %s" % e)
                continue
            bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
            rec["runs"].append({"bad_buckets": [int(b) for b in bad[:8]], "n_bad": int(len(bad)),
                                "max_abs": float(np.abs(got - want).max()),
                                "first": [float(got[bad[0]]), float(want[bad[0]])] if len(bad) else None})
        cloud.close()
        print(json.dumps(rec), flush=True)

    compare("whole", pts)
    compare("floor_only", floor)
    compare("rest_only", rest)
    for n in (15000, 15300, 15400, 16000, 16384, 16400, 18000, 19000, 20000):
        compare("floor_prefix_%d" % n, floor[:n], reps=1)
    # a floor of the same size with range noise passes in the tests: noise on the floor alone
    rng = np.random.RandomState(3)
    noisy = floor.copy()
    noisy[:, :2] += rng.normal(0, 0.01, (len(floor), 2)).astype(np.float32)
    compare("floor_xy_noise", noisy, reps=1)
    for a in slice_angle_arrays(floor):
        got = dl.diag_std_sort_order(ctx, a)
        want = orc.std_sort_order(a)
        print(json.dumps({"case": "sort_order", "n": int(len(a)), "mismatches": int((got != want).sum())}), flush=True)
    for order_name, arr in (("input", floor), ("reversed", floor[::-1])):
        vals = np.ascontiguousarray(arr[:, :2].T)
        got = dl.diag_sequential_sums(ctx, vals)
        want = np.array([np.add.accumulate(vals[0], dtype=np.float32)[-1], np.add.accumulate(vals[1], dtype=np.float32)[-1]], np.float32)
        print(json.dumps({"case": "sums_" + order_name, "equal": bool(np.array_equal(got.view(np.uint32), want.view(np.uint32))),
                          "got": [float(v) for v in got], "want": [float(v) for v in want]}), flush=True)


if __name__ == "__main__":
    main()
