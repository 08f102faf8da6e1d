#!/bin/bash
# bits = 8 grid, histogram after the queue row layout, W-ref chain with the IMU batch entry, config-5 bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/r3_exp10
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "bits_8 or rotational_histogram or local_trajectory_builder_adapter or hybrid_grid" 2>&1 | tail -8
bash tools/r3_exp6.sh 2>&1 | grep -v "^\." | tail -14
timeout 300 python tools/wref_full.py > $OUT/wref_full.json 2> $OUT/wref_full.err; echo "wref rc=$?"; tail -c 1500 $OUT/wref_full.json
timeout 900 python bench.py --config 5 > $OUT/config5_bench.json 2> $OUT/config5_bench.err; echo "config5 rc=$?"; tail -c 3000 $OUT/config5_bench.json; tail -5 $OUT/config5_bench.err
