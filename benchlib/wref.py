"""The W-ref lines of bench.py: the reference's complete per-scan chain (tools/wref_full.py)."""
import os
import sys
import time

import numpy as np

from benchlib import (ROOT, RTCSM_OPTS, CSM_OPTS, HIT_P, MISS_P, FREE, HIGH_RES_MAX_RANGE, HBM_PEAK_GBS, VALU_PEAK_LANE_OPS,
                      build_scene, insertion_targets)


def wref_line(dl, ctx, cpu):
    """What the reference does with a 64 x 1024 scan, complete (tools/wref_full.py): AddImuData, AddRangeData (voxel
    filters + de-skew), adaptive filters + [RTCSM3D] + Ceres, WindowOptimize, insertion, ComputeHistogram -- with
    trajectory_builder_3d.lua's options and with dlio/config/basic_config_3d.lua's (what D-LIOM ships: RTCSM3D off, 0.3 /
    0.2 / 60 m, gravity factor on); each with the same stream on the CPU oracle and the pose difference between the legs."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import wref_full
    out = {name: wref_full.line(dl, ctx, name, scans=24, warmup=4, cpu_scans=20, cpu=cpu)
           for name in ("trajectory_builder_3d", "basic_config_3d")}
    # round 4: the same chains on a world with a floor (dliom.synth's yard: ragged scans, a 15 000-return floor slice for
    # ComputeHistogram, returns to 80 m) -- the cube has neither floor nor far returns inside the beams
    for name in ("trajectory_builder_3d", "basic_config_3d"):
        out[name + "_yard"] = wref_full.line(dl, ctx, name, scans=24, warmup=4, cpu_scans=20, cpu=cpu, scene="ground")
    # round 6: the same streams through the C++ adapters, timed in C++ (tools/wref_cpp.cc: LocalTrajectoryBuilder3D's
    # AddImuData / AddRangeData with the whole MatchingResult assembled -- what a cartographer process calls; the lines
    # above drive the C ABI from Python, whose per-call overhead is in their figures).  One run per stream here.
    try:
        import wref_cpp
        from dliom import synth
        cpp = wref_cpp.measure(dl, synth, scans=24, warmup=4, runs=1, pinned_scans=True)
        for k, v in cpp.items():
            if k in out:
                out[k]["cpp_adapter"] = {x: v[x] for x in ("harness", "scans_per_s", "p50_ms", "p99_ms", "read_backs_per_scan", "results", "inserted")}
                if "scans_per_s_with_pinned_scans" in v:  # the caller's scan buffers page-locked (dliom_host_register): the upload is one DMA
                    out[k]["cpp_adapter"]["scans_per_s_with_pinned_scans"] = v["scans_per_s_with_pinned_scans"]
    except Exception as e:  # g++ missing on the box, ...: the Python-driven lines stand
        out["cpp_adapter_error"] = ("%s: %s" % (type(e).__name__, e))[:300]
    return out

