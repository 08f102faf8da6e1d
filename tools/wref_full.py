#!/usr/bin/env python3
"""W-ref, complete: EVERYTHING LocalTrajectoryBuilder3D does with an incoming 64 x 1024 scan, per scan, with the
reference's own option sets -- not only the matchers.

  AddImuData x 20 + predict                      (local_trajectory_builder_3d.cc:179-199)    host (15-state problem)
  AddRangeData: voxel filter, de-skew, range gate, voxel filter, transform  (:393-487)      device
  AddAccumulatedRangeData: adaptive filters, [RTCSM3D], CeresScanMatcher3D  (:493-572)      device
  WindowOptimize (+ EstimateGravity / gravity factor when enabled)          (:693-863)      host
  InsertIntoSubmap: ActiveSubmaps3D::InsertRangeData                        (:584-622)      device
  RotationalScanMatcher::ComputeHistogram of the gravity-aligned returns    (:605-610)      device

Two option sets:
  trajectory_builder_3d  cartographer/configuration_files/trajectory_builder_3d.lua (RTCSM3D on, 0.15 / 0.10 / 20 m)
  basic_config_3d        dlio/config/basic_config_3d.lua, what D-LIOM ships (RTCSM3D OFF, voxel filter 0.3, 0.2 m grid,
                         60 m, 100 scans per submap, Ceres weights 6 / 45, gravity factor on, 7-frame estimator)

and the same chain on the CPU oracle (one thread, like the reference runs it; WindowOptimize is the same host code in both
TIMED legs -- GTSAM is not in the tree), with the largest pose difference between the two legs as the parity figure, plus an
untimed INDEPENDENT leg whose WindowOptimize is the numpy batch smoother of oracle/imu_window_ref.py (NumpyWindow).  The raw
scan crosses PCIe inside AddRangeData, as it would in cartographer_ros: these rates are PCIe-inclusive."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "d-liom_amd"))
sys.path.insert(0, ROOT)

NOISE = [0.08, 0.004, 4e-5, 2e-6]


def option_sets():
    base = dict(
        high_resolution_adaptive_voxel_filter=dict(max_length=2.0, min_num_points=150, max_range=15.0),
        low_resolution_adaptive_voxel_filter=dict(max_length=4.0, min_num_points=200, max_range=60.0),
        use_online_correlative_scan_matching=True,
        real_time_correlative_scan_matcher=dict(linear_search_window=0.15, angular_search_window=np.deg2rad(1.0),
                                                translation_delta_cost_weight=1e-1, rotation_delta_cost_weight=1e-1),
        ceres_scan_matcher=dict(occupied_space_weight=[1.0, 6.0], translation_weight=5.0, rotation_weight=4e2,
                                only_optimize_yaw=False, use_nonmonotonic_steps=False, max_num_iterations=12,
                                num_threads=1),
        motion_filter=dict(max_time_seconds=0.5, max_distance_meters=0.1, max_angle_radians=0.004),
        submaps=dict(high_resolution=0.10, high_resolution_max_range=20.0, low_resolution=0.45, num_range_data=160,
                     hit_probability=0.55, miss_probability=0.49, num_free_space_voxels=2))
    dlio = json.loads(json.dumps(base))  # deep copy
    dlio["use_online_correlative_scan_matching"] = False               # basic_config_3d.lua:56
    dlio["real_time_correlative_scan_matcher"].update(linear_search_window=0.1, angular_search_window=float(np.deg2rad(3.0)),
                                                      rotation_delta_cost_weight=0.3)  # :57-60 (unused while off)
    dlio["ceres_scan_matcher"].update(translation_weight=6.0, rotation_weight=4.5e1)  # :94-95
    dlio["motion_filter"] = dict(max_time_seconds=0.5, max_distance_meters=0.2, max_angle_radians=float(np.deg2rad(5.0)))  # :91-93
    dlio["submaps"].update(high_resolution=0.2, high_resolution_max_range=60.0, num_range_data=100)  # :63-67
    return {
        "trajectory_builder_3d": dict(front_end=base, voxel_filter_size=0.15, min_range=1.0, max_range=100.0,
                                      # WindowOptimize by the reference's rule: every key until the graph reset at
                                      # submaps.num_range_data (160), ISAM2's relinearisation threshold (0.1)
                                      window=dict(window_size=0, graph_reset_every=160)),
        "basic_config_3d": dict(front_end=dlio, voxel_filter_size=0.3, min_range=0.5, max_range=100.0,      # :62, :88-89
                                window=dict(enable_gravity_factor=1, frames_for_online_gravity_estimate=7,  # :77-81
                                            window_size=0, graph_reset_every=100)),  # the reference's rule, num_range_data 100
    }


_STREAMS = {}


def make_stream(synth, scans, beams, azimuths):
    key = (synth.SCENE["name"], tuple(sorted((k, str(v)) for k, v in synth.TRAJ.items())), scans, beams, azimuths)
    if key not in _STREAMS:  # ray casting the yard's 140 boxes costs a second per scan: both option sets share a stream
        _STREAMS[key] = _make_stream(synth, scans, beams, azimuths)
    return _STREAMS[key]


def _make_stream(synth, scans, beams, azimuths):
    T = 0.1
    centers = synth.bubbles()
    clouds = [synth.moving_scan(T * k, beams, azimuths, centers) for k in range(1, scans + 1)]
    imus = [synth.imu_samples(T * (k - 1), T * k, 200.0, NOISE[:2], seed=11 + k) for k in range(1, scans + 1)]
    return T, clouds, imus, synth.trajectory_state(0.0)


OPT_NAMES = ("acc_noise", "gyr_noise", "acc_bias_noise", "gyr_bias_noise", "gravity", "integration_sigma",
             "prior_pose_noise", "prior_velocity_sigma", "prior_bias_sigma", "ceres_pose_noise_t", "ceres_pose_noise_r",
             "ceres_pose_noise_t_drift", "ceres_pose_noise_r_drift", "prior_gravity_noise", "tangent_preintegration")


LAST_RUN = {}  # what the last run_chain fed its window: matched poses, statuses, predicted velocities (for the shadow run)


def shadow_window(dl, cfg, imus, state0, matched, status, pv):
    """The numpy window fed with EXACTLY what another leg's window was fed (IMU samples, matched poses, its
    re-initialisations): the two windows' estimates then differ by the windows alone -- no feedback through the matcher,
    which on a scene like the yard turns a 1e-5 m difference of the prediction into another 10 cm candidate."""
    w = NumpyWindow(dl, dict(acc_noise=NOISE[0], gyr_noise=NOISE[1], acc_bias_noise=NOISE[2], gyr_bias_noise=NOISE[3], **cfg["window"]))
    w.initialize(state0[:7], state0[7:10], np.zeros(6))
    w.window_optimize(state0[:7])  # the graph starts at the initial state, as in run_chain
    out = []
    for (dt, acc, gyr), m, st, v in zip(imus, matched, status, pv):
        w.add_imu_batch(acc[:-1], gyr[:-1], dt)
        est, _, _, _ = w.window_optimize(m)
        if st != 0:
            w.initialize(m, v, np.zeros(6))
            w.window_optimize(m)
            est = m
        out.append(est)
    return np.array(out)


class NumpyWindow:
    """WindowOptimize of the INDEPENDENT parity leg: oracle/imu_window_ref.py's ReferenceRuleSmoother -- every key since
    the last reset in one batch problem, numerical Jacobians, solved to convergence -- behind the interface of
    dl.ImuWindow.  Not the product's imu_window.cc and not derived from it: a difference between the two legs' poses is a
    difference between two implementations of the IMU window (a16 / f4 stay PARITY UNPINNED against GTSAM itself)."""

    def __init__(self, dl, overrides):
        from oracle.imu_window_ref import ReferenceRuleSmoother
        w = dl.ImuWindow(**overrides)  # only to read the option values the product runs with
        opts = {n: getattr(w.options, n) for n in OPT_NAMES}
        for n in ("enable_gravity_factor", "frames_for_online_gravity_estimate"):
            opts[n] = int(getattr(w.options, n))
        reset = int(w.options.graph_reset_every)
        rule = int(w.options.window_size) == 0  # the reference's rule: ISAM2's relinearisation threshold, two updates a scan
        self.s = ReferenceRuleSmoother(opts, num_range_data=reset if reset > 0 else 10 ** 9,
                                       relinearize_threshold=float(w.options.relinearize_threshold) if rule else None,
                                       updates=int(w.options.iterations))
        w.close()

    def initialize(self, pose7, vel, bias6):
        self.s.initialize(pose7, vel, bias6)
        self._started = False

    def add_imu_batch(self, acc, gyr, dt):
        for a, g in zip(acc, gyr):
            self.s.add_imu(a, g, dt)

    @staticmethod
    def _pose7(R, p):
        from scipy.spatial.transform import Rotation as Rot
        q = Rot.from_matrix(R).as_quat()
        return np.concatenate([p, [q[3], q[0], q[1], q[2]]])

    def predict(self):
        from oracle.imu_window_ref import BatchSmoother
        R, p, v, _, _ = BatchSmoother._predict(self.s, self.s.estimate(), self.s.cur)
        return self._pose7(R, p), v

    def add_pose(self, matched):
        R, p, v, ba, bg = self.s.add_pose(matched, iterations=5)
        return self._pose7(R, p), v, np.concatenate([ba, bg]), 0

    def window_optimize(self, matched):
        """WindowOptimize as the reference calls it: the first call after initialize() only starts the graph (the
        preintegration since then is dropped, the initial state is returned, .cc:712-745); later ones add a key."""
        if not getattr(self, "_started", False):
            from oracle.imu_window_ref import make_preintegration
            self._started = True
            R, p, v, ba, bg = self.s.estimate()
            self.s.cur = make_preintegration(ba, bg, self.s.o)
            return self._pose7(R, p), v, np.concatenate([ba, bg]), 0
        return self.add_pose(matched)

    def gravity_estimate(self):
        return self.s.g_est, bool(self.s.g_valid), int(self.s.gravity_factors)


def run_chain(dl, cfg, T, clouds, imus, state0, device, ctx=None, orc=None, histogram_size=120, cpu_threads=1,
              numpy_window=False):
    """One pass over the stream; returns (per-scan stage seconds [n x 6], poses [n x 7], histograms, gravity factors).
    numpy_window: WindowOptimize by NumpyWindow instead of the product's host code (the independent parity leg)."""
    overrides = dict(acc_noise=NOISE[0], gyr_noise=NOISE[1], acc_bias_noise=NOISE[2], gyr_bias_noise=NOISE[3], **cfg["window"])
    window = NumpyWindow(dl, overrides) if numpy_window else dl.ImuWindow(**overrides)
    window.initialize(state0[:7], state0[7:10], np.zeros(6))
    # The reference starts its factor graph at the first WindowOptimize call after InitializeIMU and reports that scan at
    # the initial pose (.cc:712-745).  The stream begins in motion and state0 IS the state at its first instant: the graph
    # is started here, so that every streamed scan is fused (the adapter's SetInitialState(..., start_graph = true)).
    window.window_optimize(state0[:7])
    fe = dl.LocalTrajectoryBuilder3D(ctx, cfg["front_end"]) if device else orc.FrontEnd(cfg["front_end"])
    if not device and cpu_threads > 1:
        fe.set_threads(cpu_threads)  # the candidate loop on 8 threads (the reference's is serial)
    vfs, rmin, rmax = cfg["voxel_filter_size"], cfg["min_range"], cfg["max_range"]
    state = state0.copy()
    rows, poses, hists = [], [], []
    LAST_RUN["matched"], LAST_RUN["status"], LAST_RUN["pv"] = [], [], []
    for k, (scan, (dt, acc, gyr)) in enumerate(zip(clouds, imus), start=1):
        t0 = time.perf_counter()
        window.add_imu_batch(acc[:-1], gyr[:-1], dt)  # the scan interval's 20 samples (one call: no Python per sample)
        pp, pv = window.predict()
        t1 = time.perf_counter()
        if device:
            cloud, origin, cur = dl.add_range_data(ctx, state[:7], pp, T, scan, (0, 0, 0), rmin, rmax, vfs)
            t2 = time.perf_counter()
            r = fe.match_cloud(cur.astype(np.float64), origin, cloud)
        else:
            ref = orc.deskew_and_filter(T, rmin, rmax, vfs, state[:7], pp, scan)
            t2 = time.perf_counter()
            r = fe.match(ref["current_pose"].astype(np.float64), ref["origin_in_tracking"], ref["returns_in_tracking"])
        t3 = time.perf_counter()
        matched = r["pose_estimate"] if not r["dropped"] else pp
        est, vel, bias, status = window.window_optimize(matched)  # (the first call only starts the graph, like the reference's)
        LAST_RUN["matched"].append(np.array(matched, dtype=np.float64))
        LAST_RUN["status"].append(int(status))
        LAST_RUN["pv"].append(np.array(pv, dtype=np.float64))
        if status != 0:  # FailureDetection / solver: re-initialise at the matched pose like ResetParams() + InitializeIMU
            window.initialize(matched, pv, np.zeros(6))
            window.window_optimize(matched)
            est, vel, bias = matched, pv, np.zeros(6)
        t4 = time.perf_counter()
        if device:  # the histogram's kernels run beside the insertion (both only read the filtered cloud): begin / finish
            dl.cloud_rotational_histogram_begin(ctx, cloud, histogram_size, rotation_wxyz=est[3:].astype(np.float32))
        ins = fe.insert(int(k * 1e6), est, est[3:])
        # (no synchronisation here: like the C++ adapter -- histogram begin, InsertIntoSubmap, histogram finish -- the
        # insertion's launches are only enqueued; the scan's cloud is released below, inside the timing, and releasing it
        # waits for the device)
        t5 = time.perf_counter()
        hist = None
        inserted = bool(ins["inserted"]) if isinstance(ins, dict) else bool(ins)  # the oracle's insert returns the flag itself
        if device:
            hist = dl.cloud_rotational_histogram_finish(ctx, histogram_size)
            if not inserted:
                hist = None
        if inserted:  # the histogram belongs to the TrajectoryNode of an inserted scan (:605-610)
            if device:
                pass
            else:
                rot = np.concatenate([np.zeros(3), est[3:]]).astype(np.float32)
                hist = orc.compute_histogram(orc.transform_points(rot, ref["returns_in_tracking"]), histogram_size)
        if device:
            cloud.close()  # inside the timing: handing the cloud's block back waits for the device (the scan's last launches)
        t6 = time.perf_counter()
        state = np.concatenate([est, vel, bias])
        rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5))
        poses.append(est)
        hists.append(hist)
    return np.array(rows), np.array(poses), hists, window.gravity_estimate()[2]


STAGES = ("imu", "add_range_data", "match", "window_optimize", "insert", "histogram")


def line(dl, ctx, name, scans=24, warmup=4, cpu_scans=20, beams=64, azimuths=1024, cpu=True, scene="cube"):
    """scene: "cube" (the reference test's closed room) or "ground" (the yard of dliom.synth: a floor 1.8 m below
    the sensor, walls, boxes, -25..+15 degree beams, returns to 80 m, rays that do not return)."""
    from dliom import synth
    cfg = option_sets()[name]
    with synth.scene(scene):
        if scene == "cube":
            synth.set_trajectory(10.0, 0.4)  # a vehicle-like arc: 4 m/s on a 10 m radius (the yard's own is level)
        T, clouds, imus, state0 = make_stream(synth, scans, beams, azimuths)
    import gc
    gc.collect()
    gc.disable()  # harness only: a full collection of CPython's cyclic collector is ~35 ms with torch imported
    try:
        read_backs0 = ctx.read_backs()
        rows, poses, hists, g_factors = run_chain(dl, cfg, T, clouds, imus, state0, True, ctx=ctx)
        read_backs = ctx.read_backs() - read_backs0
        dev_fed = {k: list(v) for k, v in LAST_RUN.items()}  # what the device leg's window was fed (for the shadow run)
    finally:
        gc.enable()
    rows = rows[warmup:]
    out = {"options": name, "scene": scene, "returns_per_scan": int(np.mean([len(c) for c in clouds])),
           "workload": "W-ref complete chain (%s): %dx%d motion-distorted scans at 10 Hz + 200 Hz IMU: AddImuData, AddRangeData, "
                       "adaptive filters + %sCeres, WindowOptimize%s, InsertIntoSubmap, ComputeHistogram; raw scans cross "
                       "PCIe inside AddRangeData; the C++ adapter's call sequence (histogram begin, insertion, histogram "
                       "finish, the scan's cloud released -- which waits for the device -- all inside the timing)" % (name, beams, azimuths,
                                                     "RTCSM3D + " if cfg["front_end"]["use_online_correlative_scan_matching"] else "",
                                                     " with gravity factor" if cfg["window"].get("enable_gravity_factor") else ""),
           "scans_per_s": 1.0 / float(np.mean(rows.sum(axis=1))),
           "p50_ms": dict({s: 1e3 * float(np.median(rows[:, i])) for i, s in enumerate(STAGES)},
                          total=1e3 * float(np.median(rows.sum(axis=1)))),
           "read_backs_per_scan": read_backs / float(len(rows) + warmup),  # polled host round trips (dliom_ctx_read_backs)
           "gravity_factors_added": int(g_factors), "scans": int(len(rows))}
    if cpu:
        from oracle import oracle as orc
        n = min(cpu_scans + 1, scans)
        crows, cposes, chists, _ = run_chain(dl, cfg, T, clouds[:n], imus[:n], state0, False, orc=orc)
        crows = crows[1:]
        per_scan = float(np.mean(crows.sum(axis=1)))
        dpos = float(np.max(np.linalg.norm(poses[:n, :3] - cposes[:, :3], axis=1)))
        dang = float(np.max([2.0 * np.arccos(min(1.0, abs(float(np.dot(a[3:], b[3:]))))) for a, b in zip(poses[:n], cposes)]))
        same_presence = all((a is None) == (b is None) for a, b in zip(hists[:n], chists))
        hdiff = max([float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) for a, b in zip(hists[:n], chists)
                     if a is not None and b is not None] or [0.0])
        out["cpu_baseline"] = {"value": 1.0 / per_scan, "unit": "scans/s", "cores": 1, "kind": "port",
                               "p50_ms": dict({s: 1e3 * float(np.median(crows[:, i])) for i, s in enumerate(STAGES)},
                                              total=1e3 * float(np.median(crows.sum(axis=1)))),
                               "sample": "the same stream through the CPU oracle (C++ restatement, 1 thread), %d scans after one "
                                         "warm-up; WindowOptimize is the same host code in both legs" % len(crows)}
        out["speedup_vs_cpu"] = out["scans_per_s"] * per_scan
        if cfg["front_end"]["use_online_correlative_scan_matching"]:
            threads = min(8, os.cpu_count() or 1)
            mrows, _, _, _ = run_chain(dl, cfg, T, clouds[:n], imus[:n], state0, False, orc=orc, cpu_threads=threads)
            per_scan_mt = float(np.mean(mrows[1:].sum(axis=1)))
            out["cpu_baseline_%d_threads" % threads] = {
                "value": 1.0 / per_scan_mt, "unit": "scans/s", "cores": threads, "kind": "port",
                "what": "the same chain with the RTCSM3D candidate loop on %d threads (BASELINE.md section 2; everything else "
                        "is serial in the reference and stays so)" % threads}
            out["speedup_vs_best_cpu"] = out["scans_per_s"] * min(per_scan, per_scan_mt)
        # the independent leg: the CPU oracle chain once more, untimed, with WindowOptimize by the numpy
        # batch solver instead of the product's imu_window.cc -- until now both legs ran the same host code for that stage
        # and their agreement said nothing about it.  Bounded: the solver's numerical Jacobians cost seconds per dozen
        # scans.  Round 6: it follows the same rule as the product's window (every key kept, ISAM2's relinearisation
        # threshold, the gravity factor), so the compared scans may carry gravity factors.
        n_ind = min(n, 13)
        t_ind = time.perf_counter()
        shadow = shadow_window(dl, cfg, imus[:n_ind], state0, dev_fed["matched"][:n_ind], dev_fed["status"][:n_ind], dev_fed["pv"][:n_ind])
        _, iposes, _, _ = run_chain(dl, cfg, T, clouds[:n_ind], imus[:n_ind], state0, False, orc=orc, numpy_window=True)
        _, gposes, _, g_ind = (None, poses[:n_ind], None, None)
        win_probe = dl.ImuWindow(acc_noise=NOISE[0], gyr_noise=NOISE[1], acc_bias_noise=NOISE[2], gyr_bias_noise=NOISE[3], **cfg["window"])
        out["parity_independent_imu_window"] = {
            "what": "device leg (product window, imu_window.cc: the reference's rule -- every key until the graph reset, "
                    "ISAM2's relinearisation threshold -- on the chain solver) against the CPU oracle chain whose WindowOptimize "
                    "is oracle/imu_window_ref.py's numpy smoother following the same rule (dense solves, numerical Jacobians)",
            "scans_compared": n_ind,
            "same_inputs": {"what": "the numpy window fed with the device leg's own IMU samples and matched poses (no feedback "
                                    "through the matcher): the difference of the two WINDOWS",
                            "max_translation_difference_m": float(np.max(np.linalg.norm(gposes[:, :3] - shadow[:, :3], axis=1))),
                            "max_rotation_difference_rad": float(np.max([2.0 * np.arccos(min(1.0, abs(float(np.dot(a[3:], b[3:])))))
                                                                         for a, b in zip(gposes, shadow)]))},
            "closed_loop": {"what": "the whole CPU oracle chain run with the numpy window: differences feed back through the "
                                    "matchers (on the yard's flat floor RTCSM3D turns 1e-5 m into another 10 cm candidate)",
                            "max_translation_difference_m": float(np.max(np.linalg.norm(gposes[:, :3] - iposes[:, :3], axis=1))),
                            "max_rotation_difference_rad": float(np.max([2.0 * np.arccos(min(1.0, abs(float(np.dot(a[3:], b[3:])))))
                                                                         for a, b in zip(gposes, iposes)]))},
            "tolerance_m": 1e-4, "window_size": int(win_probe.options.window_size),
            "graph_reset_every": int(win_probe.options.graph_reset_every),
            "relinearize_threshold": float(win_probe.options.relinearize_threshold),
            "gravity_factor_enabled": bool(cfg["window"].get("enable_gravity_factor")),
            "note": "PARITY UNPINNED against GTSAM itself (absent from the reference tree)",
            "seconds": time.perf_counter() - t_ind}
        win_probe.close()
        p_ind = out["parity_independent_imu_window"]
        p_ind["ok"] = bool(p_ind["same_inputs"]["max_translation_difference_m"] <= 1e-4 and
                           p_ind["same_inputs"]["max_rotation_difference_rad"] <= 1e-4)
        out["parity"] = {"scans_compared": n, "max_translation_difference_m": dpos, "max_rotation_difference_rad": dang,
                         "tolerance_m": 1e-4, "ok": bool(dpos <= 1e-4 and dang <= 1e-4),
                         "histograms_same_scans": bool(same_presence), "histograms_max_abs_difference": hdiff,
                         "note": "the two legs start every scan from their OWN previous estimate (Ceres agrees to ~1e-9, not to "
                                 "the bit), so the histograms are those of slightly different rotations; bit equality of the "
                                 "histogram kernels on identical input is tests/test_gpu_parity.py's job"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--cpu-scans", type=int, default=20)
    ap.add_argument("--scene", default="cube", choices=("cube", "ground"))
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--options", default="both", choices=("both", "trajectory_builder_3d", "basic_config_3d"))
    a = ap.parse_args()
    import dliom as dl
    ctx = dl.Context(0)
    names = ("trajectory_builder_3d", "basic_config_3d") if a.options == "both" else (a.options,)
    print(json.dumps({n: line(dl, ctx, n, a.scans, a.warmup, a.cpu_scans, cpu=not a.no_cpu, scene=a.scene) for n in names}))


if __name__ == "__main__":
    main()
