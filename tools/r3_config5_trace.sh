#!/bin/bash
# config 5 (128x2048, 5 cm) under rocprofv3 kernel trace -> profiles/r3_config5_rocprofv3_kernel_stats.csv
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/config5_trace_r3
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o c5 -- python $R/bench.py --config 5 --steps 3 --warmup 1 --no-pmc --no-cpu-baseline > $OUT/trace.log 2>&1; echo "trace rc=$?"
f=$(find $OUT -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv
rm -f $OUT/trace/*/*kernel_trace.csv $OUT/trace/*/*.db 2>/dev/null
cut -c1-150 $OUT/kernel_stats.csv | head -12
