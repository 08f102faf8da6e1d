#!/usr/bin/env python3
"""W-ref through the C++ adapters (tools/wref_cpp.cc): the streams and option sets of tools/wref_full.py written to a file,
the harness compiled with g++ against libdliom.so and run; one JSON object per (options, scene).  That the adapters
give the poses of the Python-driven chain bit for bit is tests/test_gpu_parity.py::test_cpp_local_trajectory_builder_adapter's
job (there both sides time-step their IMU samples alike; here AddImuData takes the steps from the timestamps).

    python tools/wref_cpp.py [--scans 24] [--warmup 4] > gpurun_out/r6_wref_cpp.json"""
import argparse
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "d-liom_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def build(tmp, stages=False):
    exe = os.path.join(tmp, "wref_cpp")
    libdir = os.path.join(ROOT, "d-liom_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2"] + (["-DDLIOM_ADAPTER_STAGE_TIMES"] if stages else []) + ["-o", exe, os.path.join(ROOT, "tools", "wref_cpp.cc"), "-L", libdir, "-ldliom",
                           "-Wl,-rpath," + libdir])
    return exe


def write_stream(dl, path, cfg, T, clouds, imus, state0, warmup, histogram_size=120):
    import wref_full
    w = dl.ImuWindow(acc_noise=wref_full.NOISE[0], gyr_noise=wref_full.NOISE[1], acc_bias_noise=wref_full.NOISE[2],
                     gyr_bias_noise=wref_full.NOISE[3], **cfg["window"])
    with open(path, "wb") as f:
        per = len(imus[0][1]) - 1
        f.write(struct.pack("6i", len(clouds), per, warmup, histogram_size, 0, 0))
        f.write(np.asarray(np.concatenate([state0[:10], np.zeros(6)]), dtype=np.float64).tobytes())
        f.write(bytes(dl.front_end_options_struct(cfg["front_end"])))
        f.write(bytes(w.options))
        f.write(struct.pack("4f", cfg["voxel_filter_size"], cfg["min_range"], cfg["max_range"], T))
        for (dt, acc, gyr), sc in zip(imus, clouds):
            assert len(acc) - 1 == per
            f.write(struct.pack("i", len(sc)))
            rows = np.concatenate([np.full((per, 1), dt), acc[:-1], gyr[:-1]], axis=1)
            f.write(np.ascontiguousarray(rows, dtype=np.float64).tobytes())
            f.write(np.ascontiguousarray(sc, dtype=np.float32).tobytes())
    w.close()


def measure(dl, synth, scans=24, warmup=4, runs=3, compare=False, pinned_scans=False, stages=False):
    """{options[_yard]: line} for both option sets on both scenes; the harness is compiled once."""
    import wref_full
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        exe = build(tmp, stages)
        for scene in ("cube", "ground"):
            with synth.scene(scene):
                if scene == "cube":
                    synth.set_trajectory(10.0, 0.4)
                T, clouds, imus, state0 = wref_full.make_stream(synth, scans, 64, 1024)
            for name, cfg in wref_full.option_sets().items():
                path, poses_path = os.path.join(tmp, "stream.bin"), os.path.join(tmp, "poses.bin")
                write_stream(dl, path, cfg, T, clouds, imus, state0, warmup)
                got_runs = []
                for _ in range(runs):  # a fresh process each: the best run is the harness's figure, all of them are printed
                    r = subprocess.run([exe, path, poses_path], capture_output=True, text=True, timeout=600)
                    if r.returncode != 0:
                        raise RuntimeError("wref_cpp failed: " + (r.stdout + r.stderr)[-400:])
                    got_runs.append(json.loads(r.stdout.strip().splitlines()[-1]))
                    if stages:
                        got_runs[-1]["stages"] = [l for l in r.stderr.splitlines() if l.startswith("stages")][-1:]
                best = max(got_runs, key=lambda x: x["scans_per_s"])
                line = dict(best, options=name, scene=scene, returns_per_scan=int(np.mean([len(c) for c in clouds])),
                            scans_per_s_all_runs=[x["scans_per_s"] for x in got_runs], imu_window=cfg["window"])
                if pinned_scans:  # the same stream with the scans in page-locked memory (dliom_host_register before the clock starts)
                    r = subprocess.run([exe, path, poses_path, "pinned"], capture_output=True, text=True, timeout=600)
                    if r.returncode == 0:
                        line["scans_per_s_with_pinned_scans"] = json.loads(r.stdout.strip().splitlines()[-1])["scans_per_s"]
                if compare:
                    ctx = dl.Context(0)
                    _, poses, _, _ = wref_full.run_chain(dl, cfg, T, clouds, imus, state0, True, ctx=ctx)
                    got = np.fromfile(poses_path, dtype=np.float64).reshape(-1, 7)
                    have = np.linalg.norm(got[:, 3:], axis=1) > 0  # scans the adapter returned a result for
                    line["pose_difference_to_python_driven_chain_m"] = float(np.max(np.linalg.norm(got[have, :3] - poses[have, :3], axis=1)))
                    line["results_compared"] = int(have.sum())
                    ctx.close()
                out[name + ("" if scene == "cube" else "_yard")] = line
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=24)  # the cube scene's arc leaves its room after ~30 scans (wref_full.py uses 24 too)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--compare", action="store_true", help="also run the Python-driven chain and print the pose difference (the two feed their IMU samples with different first time steps: centimetres through the matcher, not a parity figure)")
    ap.add_argument("--stages", action="store_true", help="build the adapter with -DDLIOM_ADAPTER_STAGE_TIMES: microseconds per scan between its marks")
    ap.add_argument("--pinned-scans", action="store_true", help="one more run per stream with the scans in page-locked memory")
    a = ap.parse_args()
    import dliom as dl
    from dliom import synth
    import wref_full
    dl.load_library()
    out = measure(dl, synth, a.scans, a.warmup, runs=3, compare=a.compare, pinned_scans=a.pinned_scans, stages=a.stages)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
