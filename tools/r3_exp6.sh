#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/r3_exp6
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "rotational_histogram or local_trajectory_builder_adapter" 2>&1 | tail -15
cat > /tmp/hist_time.py <<'PY'
import sys, time
sys.path.insert(0, "/root/repo/d-liom_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, dliom as dl
from dliom import synth
from oracle import oracle as orc
ctx = dl.Context(0)
raw, _ = synth.scan(synth.trajectory_pose(0.7), 64, 1024)
pts = raw[orc.voxel_filter(0.15, raw)]
cloud = dl.PointCloud(ctx, pts)
rot = np.array([0.999, 0.01, -0.02, 0.03], np.float32); rot /= np.linalg.norm(rot)
for _ in range(5): dl.cloud_rotational_histogram(ctx, cloud, 120, rot)
t = time.perf_counter()
for _ in range(200): dl.cloud_rotational_histogram(ctx, cloud, 120, rot)
print("device histogram: %.1f us per call, %d points" % ((time.perf_counter() - t) / 200 * 1e6, len(pts)))
t = time.perf_counter()
for _ in range(20): dl.rotational_histogram(pts, 120)
print("host histogram: %.1f us per call" % ((time.perf_counter() - t) / 20 * 1e6))
PY
timeout 120 python /tmp/hist_time.py
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python /tmp/hist_time.py > $OUT/trace.log 2>&1
cd $R; python3 - <<PY
import glob
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for l in open(f).read().splitlines()[:8]: print(l[:200])
PY
