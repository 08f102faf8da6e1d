"""Experiments build: where csm_lm_kernel (the whole Ceres loop in one launch, W-ref's ~170 + ~210 points) spends its time."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "d-liom_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import dliom as dl
    from dliom import synth
    import wref
    ctx = dl.Context(0)
    L = dl.load_library()
    for scene in ("cube", "ground"):
        with synth.scene(scene):
            fe = dl.LocalTrajectoryBuilder3D(ctx, wref.OPTS)
            gravity = np.array([1.0, 0, 0, 0])
            origin = np.zeros(3, np.float32)
            for s in range(12):
                truth = synth.trajectory_pose(0.025 * s)
                pts, _ = synth.scan(truth, 64, 1024)
                pred = synth.perturb_pose(truth, 0.03, 0.2, seed=100 + s)
                raw = dl.PointCloud(ctx, pts)
                f = raw.voxel_filter(0.15)
                t0 = time.perf_counter()
                r = fe.match_cloud(pred, origin, f)
                ctx.synchronize()
                t1 = time.perf_counter()
                fe.insert(int(s * 250000), r["pose_estimate"], gravity)
                ctx.synchronize()
                raw.close()
                f.close()
            buf = (ctypes.c_ulonglong * 128)()
            L.dliom_exp_lm_stamps(buf)
            a = [int(v) for v in buf]
            n = a[127]
            print(scene, "match ms", 1e3 * (t1 - t0), "N", r["num_high"], r["num_low"], "evaluations", n, "kernel cycles", a[126] - a[0])
            for e in range(min(n, 24)):
                b = a[5 * e:5 * e + 4]
                nxt = a[5 * (e + 1)] if e + 1 < min(n, 24) else a[126]
                print("   eval %2d: points %6d  reduce %6d  finish %6d | LM step until next eval %6d" % (e, b[1] - b[0], b[2] - b[1], b[3] - b[2], nxt - b[3]))
            b2 = (ctypes.c_ulonglong * 256)()
            L.dliom_exp_lm2_stamps(b2)
            c = [int(v) for v in b2]
            for it in range(1, 5):
                q = c[8 * it:8 * it + 7]
                print("   iteration %d: scale %5d solve %5d model+plus %5d [eval %6d] quality %5d accept %5d" %
                      (it, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[6] - q[5]))
            fe.close()


if __name__ == "__main__":
    main()
