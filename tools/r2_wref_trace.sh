#!/bin/bash
# W-ref chain under rocprofv3 kernel trace: per-kernel GPU time vs the wall time of tools/wref.py's stages.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/wref_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o w -- python $R/tools/wref.py --stages > $OUT/trace.log 2>&1; echo "trace rc=$?"
grep '^{' $OUT/trace.log > $OUT/stages_under_trace.json
f=$(find $OUT -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv; head -40 $OUT/kernel_stats.csv | cut -c1-200
rm -f $OUT/trace/*/*kernel_trace.csv $OUT/trace/*/*.db 2>/dev/null
