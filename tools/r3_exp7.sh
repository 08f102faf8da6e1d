#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cat > /tmp/hist_dbg.py <<'PY'
import sys, time, ctypes
sys.path.insert(0, "/root/repo/d-liom_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, dliom as dl
from dliom import synth
from oracle import oracle as orc
ctx = dl.Context(0)
raw, _ = synth.scan(synth.trajectory_pose(0.7), 64, 1024)
pts = raw[orc.voxel_filter(0.15, raw)]
cloud = dl.PointCloud(ctx, pts)
for _ in range(3): h = dl.cloud_rotational_histogram(ctx, cloud, 120)
print("hist top buckets:", np.sort(h)[-5:], "sum", h.sum())
L = dl.load_library()
buf = (ctypes.c_ulonglong * (64 * 16))()
L.dliom_exp_rothist_stamps(buf)
a = np.array(buf, dtype=np.uint64).reshape(64, 16)
for b in range(0, 64, 6):
    s = a[b].astype(np.int64)
    print("wg", b, "count/m/E", s[10], s[11], s[12], "phases(cycles):", [int(s[k + 1] - s[k]) for k in range(7)])
buf2 = (ctypes.c_ulonglong * (128 * 8))()
L.dliom_exp_rothist_acc_stamps(buf2)
b = np.array(buf2, dtype=np.uint64).reshape(128, 8).astype(np.int64)
order = np.argsort(-b[:, 5])[:4]
for k in list(order) + [3, 50]:
    print("bucket", k, "queued", b[k, 5], "scan", b[k, 1] - b[k, 0], "gather", b[k, 2] - b[k, 1], "sum", b[k, 3] - b[k, 2])
PY
DLIOM_LIB=$R/d-liom_amd/ab/libdliom_exp.so timeout 120 python /tmp/hist_dbg.py
