#!/usr/bin/env python3
"""How accurate is the reference's own scan-to-submap matching on the synthetic scene?  CPU oracle chain only
(no GPU): every scan is de-skewed and matched starting from the GROUND-TRUTH pose, so whatever error comes out is the
matcher's (RTCSM3D at 10 cm / 1 deg + CeresScanMatcher3D against a submap built from the few scans before it).
Context for tools/stream.py's pose errors: with a noise-free IMU the prediction is better than any match on this
scene (2-5 cm per scan, mostly along z: +-15 degree beams in a 30 m cube), so the fixed-lag window's output lies
between the matched pose and the prediction by construction."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "d-liom_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gentle", action="store_true", help="4 m/s on a 10 m radius instead of the 1 m corkscrew")
    ap.add_argument("--static-scans", action="store_true", help="scans without motion distortion")
    ap.add_argument("--scans", type=int, default=6)
    a = ap.parse_args()
    from dliom import synth
    from oracle import oracle as orc
    from tools.wref import OPTS
    if a.gentle:
        synth.set_trajectory(10.0, 0.4)
    T = 0.1
    centers = synth.bubbles()
    fe = orc.FrontEnd(OPTS)
    gravity = np.array([1.0, 0, 0, 0])
    errs = []
    for k in range(1, a.scans + 1):
        truth, prev = synth.trajectory_pose(T * k), synth.trajectory_pose(T * (k - 1))
        if a.static_scans:
            pts, _ = synth.scan(truth, 64, 1024, centers=centers)
            ranges = np.concatenate([pts, np.zeros((len(pts), 1), np.float32)], axis=1).astype(np.float32)
        else:
            ranges = synth.moving_scan(T * k, 64, 1024, centers)
        ref = orc.deskew_and_filter(T, 1.0, 100.0, 0.15, prev, truth, ranges)
        r = fe.match(ref["current_pose"].astype(np.float64), ref["origin_in_tracking"], ref["returns_in_tracking"])
        est = r["pose_estimate"]
        fe.insert(int(k * 1e6), est, gravity)
        errs.append(est[:3] - truth[:3])
        print("scan %d: matched - truth = %s  (|.| = %.3f m)" % (k, np.round(errs[-1], 3), np.linalg.norm(errs[-1])))
    e = np.array(errs[1:])  # the first scan meets an empty submap and keeps the prediction
    print("mean |error| %.3f m, mean error vector %s" % (float(np.linalg.norm(e, axis=1).mean()), np.round(e.mean(axis=0), 3)))


if __name__ == "__main__":
    main()
