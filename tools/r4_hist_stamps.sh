#!/bin/bash
# experiments build (make -C d-liom_amd experiments, locally): where the histogram kernels spend their time
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cat > /tmp/hist_dbg.py <<'PY'
import sys, time, ctypes
sys.path.insert(0, "/root/repo/d-liom_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, dliom as dl
from dliom import synth
from oracle import oracle as orc
ctx = dl.Context(0)
L = dl.load_library()
import os
if os.environ.get("COOP_MIN"):
    L.dliom_exp_set_coop_min(int(os.environ["COOP_MIN"]))
    print("coop min", os.environ["COOP_MIN"])
for scene, size in (("cube", 0.15), ("ground", 0.15), ("ground", 0.0)):
    with synth.scene(scene):
        raw, _ = synth.scan(synth.trajectory_pose(0.5), 64, 1024)
    pts = raw[orc.voxel_filter(size, raw)] if size > 0 else raw
    cloud = dl.PointCloud(ctx, pts)
    for _ in range(3): h = dl.cloud_rotational_histogram(ctx, cloud, 120)
    t = time.perf_counter()
    for _ in range(20): h = dl.cloud_rotational_histogram(ctx, cloud, 120)
    print(scene, size, len(pts), "us per call", (time.perf_counter() - t) / 20 * 1e6)
    buf = (ctypes.c_ulonglong * (64 * 16))()
    L.dliom_exp_rothist_stamps(buf)
    a = np.array(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
    total = a[:, 7] - a[:, 0]
    for b in np.argsort(-total)[:4]:
        s = a[b]
        print("  wg", b, "count/m/E", s[10], s[11], s[12], "phases(cycles):", [int(s[k + 1] - s[k]) for k in range(7)],
              "sort", int(s[13] - s[3]), "ties+items", int(s[14] - s[13]), "replay", int(s[15] - s[14]), "order", int(s[4] - s[15]))
    L.dliom_exp_rothist_big_stamps(buf)
    a = np.array(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
    for b in range(2):
        print("  big prepare wg", b, [int(a[b, k + 1] - a[b, k]) for k in range(4)])
        print("  big slice   wg", b, "count/m", a[4 + b, 10], a[4 + b, 11], [int(a[4 + b, k + 1] - a[4 + b, k]) for k in range(6)])
    so = a[8]
    print("  LDS sort order: ties", int(so[1] - so[0]), "ranks+items", int(so[2] - so[1]), "workgroup partitions", int(so[3] - so[2]), "(%d rounds, %d segments left)" % (int(so[15] & 0xffffffff), int(so[15] >> 32)),
          "wave stage", int(so[4] - so[3]), "tie groups", int(so[5] - so[4]))
    rounds = int(so[15] >> 32)
    print("  big sort order: tie detect + init", int(so[1] - so[0]), "global rounds", [int(so[k + 1] - so[k]) for k in range(1, min(rounds, 8))],
          "handover T", int(so[15] & 0xffffffff), "prep", int(so[11] - so[min(rounds, 9)]), "LDS replay", int(so[12] - so[11]),
          "write back", int(so[13] - so[12]), "tie groups", int(so[14] - so[13]))
    es = (ctypes.c_ulonglong * 16)()
    L.dliom_exp_exact_sum_stamps(es)
    e = [int(v) for v in es]
    print("  exact sums (last call of block 0): k=0", e[1] - e[0], "k=1", e[2] - e[1], "walk", e[8] - e[2], "| last k: loads+prefix", e[10] - e[1],
          "scans", e[11] - e[10], "functions", e[12] - e[11], "runs", e[2] - e[12])
    cloud.close()
PY
DLIOM_LIB=$R/d-liom_amd/ab/libdliom_exp.so timeout 300 python /tmp/hist_dbg.py
