#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2_exp2
mkdir -p $OUT
cd $R
for M in 3 2; do
  DLIOM_SCORE_MAPPING=$M KBENCH_CHECK=48 timeout 600 python tools/kbench.py --reps 10 --check > $OUT/kbench_map$M.log 2>&1
  echo "kbench mapping=$M rc=$?"; grep -A1 "^rtcsm\|check\|Error\|error" $OUT/kbench_map$M.log | head
done
DLIOM_BOX_MIN_LOG2=0 timeout 1200 python -m pytest tests -m gpu -x -q -k "rtcsm or score or golden or front_end or shard" > $OUT/pytest_box.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest_box.log
