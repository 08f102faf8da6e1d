#!/bin/bash
# The complete W-ref chain (tools/wref_full.py) under rocprofv3 kernel trace: which kernels a scan launches, how often.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/wref_trace_r3
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o w -- python $R/tools/wref_full.py --no-cpu --options trajectory_builder_3d --scans 24 --warmup 0 > $OUT/trace.log 2>&1; echo "trace rc=$?"
f=$(find $OUT -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv
rm -f $OUT/trace/*/*kernel_trace.csv $OUT/trace/*/*.db 2>/dev/null
cut -c1-160 $OUT/kernel_stats.csv | head -70
