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
    return out

