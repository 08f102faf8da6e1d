#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/r4_hist2
mkdir -p $OUT
timeout 120 python -m pytest tests/test_gpu_parity.py -q -x --timeout 60 -k "sequential_sums or std_sort_order or rotational_histogram" 2>&1 | tail -15
timeout 300 python tools/hist_bench.py --check > $OUT/hist_bench.json 2> $OUT/hist_bench.err
cat $OUT/hist_bench.json; tail -3 $OUT/hist_bench.err
bash tools/r4_hist_stamps.sh
