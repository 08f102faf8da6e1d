"""Voxel filter on a 64 x 1024 scan: wall time of VoxelFilter(0.15) (fill, insert, read-back, flag, compact, read-back).
Run under rocprofv3 --kernel-trace --stats for the kernels' durations (profiles/r5_voxel_filter_kernel_stats.csv)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "d-liom_amd")]
import numpy as np  # noqa: E402

import dliom as dl  # noqa: E402
from dliom import synth  # noqa: E402


def main():
    dl.load_library()
    ctx = dl.Context(0)
    with synth.scene("cube"):
        raw, _ = synth.scan(synth.trajectory_pose(0.5), 64, 1024)
    cloud = dl.PointCloud(ctx, raw)
    out = {}
    for size in (0.15, 0.05, 2.0):
        for _ in range(3):
            f = cloud.voxel_filter(size)
            f.close()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            f = cloud.voxel_filter(size)
            f.close()
        ctx.synchronize()
        out["voxel_filter_%g_us" % size] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
    out["reruns"] = ctx.voxel_filter_reruns()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
