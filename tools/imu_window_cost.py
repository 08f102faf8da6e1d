#!/usr/bin/env python3
"""What WindowOptimize costs per scan on the host (d-liom_amd/csrc/imu_window.cc, no GPU involved), by mode:

  fixed lag 4 / 8 (+ gravity)       the round 3-5 smoother: Gauss-Newton over the window, older keys marginalised
  reference rule, threshold 0.1     every key until the graph reset at num_range_data (100: dlio/config/basic_config_3d.lua),
                                    ISAM2's relinearisation rule on the chain solver: what the adapter runs by default
  reference rule, threshold 0       the same graph with every key relinearised at every update (batch Gauss-Newton)

per key count (the graph's size when the scan arrives): mean over all scans and the scans that find >= 90 keys.

    python tools/imu_window_cost.py [--scans 400] > profiles/r6_imu_window_cost.json"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "d-liom_amd"))
sys.path.insert(0, ROOT)


def run(dl, synth, scans, **opts):
    w = dl.ImuWindow(**opts)
    st = synth.trajectory_state(0.0)
    w.initialize(st[:7], st[7:10], np.zeros(6))
    T = 0.1
    rows = []
    for k in range(1, scans + 1):
        dt, acc, gyr = synth.imu_samples(T * (k - 1), T * k, 200.0, (0.02, 0.002), seed=11 + k)
        w.add_imu_batch(acc[:-1], gyr[:-1], dt)
        matched = synth.perturb_pose(synth.trajectory_pose(T * k), 0.02, 0.1, seed=70 + k)
        keys = len(w)
        t0 = time.perf_counter()
        _, _, _, status = w.add_pose(matched)
        rows.append((keys, 1e6 * (time.perf_counter() - t0)))
        assert status == 0, status
    relin, blocks = w.solver_stats()
    rows = np.array(rows)
    big = rows[rows[:, 0] >= 90]
    return {"options": opts, "scans": scans, "mean_us_per_scan": float(rows[:, 1].mean()), "p50_us": float(np.median(rows[:, 1])),
            "max_us": float(rows[:, 1].max()), "mean_us_at_90_or_more_keys": float(big[:, 1].mean()) if len(big) else None,
            "largest_graph_keys": int(rows[:, 0].max()) + 1, "relinearizations": relin, "blocks_eliminated_per_scan": blocks / scans}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=400)
    a = ap.parse_args()
    import dliom as dl
    from dliom import synth
    dl.load_library()
    grav = dict(enable_gravity_factor=1, frames_for_online_gravity_estimate=7)
    out = {
        "what": "dliom_imu_window_add_pose, microseconds per scan on this host (one thread), corkscrew stream with IMU and "
                "matcher noise; ctypes call overhead (~2 us) included",
        "fixed_lag_4": run(dl, synth, a.scans, window_size=4),
        "fixed_lag_8_gravity": run(dl, synth, a.scans, window_size=8, **grav),
        "fixed_lag_8_gravity_reset_100": run(dl, synth, a.scans, window_size=8, graph_reset_every=100, **grav),
        "reference_rule_100_keys": run(dl, synth, a.scans, window_size=0, graph_reset_every=100),
        "reference_rule_100_keys_gravity": run(dl, synth, a.scans, window_size=0, graph_reset_every=100, **grav),
        "reference_rule_160_keys": run(dl, synth, a.scans, window_size=0, graph_reset_every=160),
        "reference_rule_100_keys_threshold_0": run(dl, synth, a.scans, window_size=0, graph_reset_every=100, relinearize_threshold=0.0),
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
