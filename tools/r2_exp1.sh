#!/bin/bash
# Round-2 experiment 1 (GPU box): instruction-rate micro-benchmarks + score-kernel pipe isolation.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2_exp1
mkdir -p $OUT
cd $R
timeout 300 tools/ubench/ubench > $OUT/ubench.json 2> $OUT/ubench.err
echo "ubench rc=$?"
for D in 0 1 2 3; do
  DLIOM_SCORE_DEBUG=$D timeout 300 python tools/kbench.py --reps 10 > $OUT/kbench_debug$D.log 2>&1
  echo "kbench debug=$D rc=$?"
  grep -A1 "^rtcsm" $OUT/kbench_debug$D.log
done
