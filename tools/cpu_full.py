#!/usr/bin/env python3
"""The CPU oracle on the bench scene WITHOUT sampling: the whole RTCSM3D candidate loop (35 937 candidates x 65 536
points) on one thread, then CeresScanMatcher3D and both insertions -- the number bench.py's `cpu_baseline` extrapolates
from 8 880 candidates.  Needs the GPU box only to build the scene the same way bench.py does; ~40 s of CPU."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "d-liom_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    import dliom as dl
    from oracle import oracle as orc
    from helpers import DEFAULT_CSM, DEFAULT_RTCSM, FREE, build_device_scene, device_grid_to_oracle
    ctx = dl.Context(0)
    ins, g_hi, g_lo, scans = build_device_scene(dl, ctx, 64, 1024, 0.10, 0.45, map_scans=20)
    sc = scans[0]
    og_hi, og_lo = device_grid_to_oracle(orc, g_hi), device_grid_to_oracle(orc, g_lo)
    t = time.perf_counter()
    r1 = orc.rtcsm3d_match(DEFAULT_RTCSM, sc["init"], sc["pts"], og_hi)
    t_rtcsm = time.perf_counter() - t
    t = time.perf_counter()
    r2 = orc.csm3d_match(DEFAULT_CSM, sc["init"][:3], r1["pose"], [(sc["pts"], og_hi), (sc["pts"], og_lo)])
    t_csm = time.perf_counter() - t
    pf = np.asarray(r2["pose"], dtype=np.float32)
    world = orc.transform_points(pf, sc["pts"])
    origin = orc.transform_points(pf, np.zeros((1, 3), np.float32))[0]
    d = (world - origin).astype(np.float32)
    nrm = np.sqrt(d[:, 0] * d[:, 0] + (d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]), dtype=np.float32)
    t = time.perf_counter()
    og_hi.insert_tables(origin, world[nrm <= np.float32(20.0)], ins.hit_table, ins.miss_table, FREE)
    og_lo.insert_tables(origin, world, ins.hit_table, ins.miss_table, FREE)
    t_ins = time.perf_counter() - t
    total = t_rtcsm + t_csm + t_ins
    print(json.dumps({"workload": "bench.py W-dense scene, CPU oracle, 1 thread, nothing sampled",
                      "best_index": int(r1["best_index"]), "points": int(len(sc["pts"])),
                      "seconds": {"rtcsm": t_rtcsm, "ceres": t_csm, "insert": t_ins, "total": total},
                      "scans_per_s": 1.0 / total, "host_cores_available": os.cpu_count(),
                      "ceres_evaluations": int(r2["num_residual_evaluations"])}))


if __name__ == "__main__":
    main()
