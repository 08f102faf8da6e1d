#!/bin/bash
# round 4: the device histogram's new paths -- tests, timing on both scene families, kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/r4_hist
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "sequential_sums or std_sort_order or rotational_histogram" 2>&1 | tail -40 > $OUT/tests.txt
tail -5 $OUT/tests.txt
timeout 300 python tools/hist_bench.py --check > $OUT/hist_bench.json 2> $OUT/hist_bench.err
cat $OUT/hist_bench.json; tail -3 $OUT/hist_bench.err
cd /tmp; export TMPDIR=/tmp
for s in cube_64x1024 yard_64x1024; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$s -o t -- python $R/tools/hist_bench.py --only $s --reps 100 > $OUT/trace_$s.log 2>&1
done
cd $R; python3 - <<PY
import glob
for f in sorted(glob.glob("$OUT/trace_*/**/*kernel_stats.csv", recursive=True)):
    print(f)
    for l in open(f).read().splitlines()[:14]: print(l[:230])
PY
