#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2_exp6
mkdir -p $OUT
cd $R
DLIOM_SCORE_MAPPING=3 KBENCH_CHECK=64 timeout 600 python tools/kbench.py --reps 2 --check --map-scans 20 2>&1 | grep -A1 "^rtcsm\|check"
DLIOM_SCORE_MAPPING=3 KBENCH_CHECK=64 timeout 600 python tools/kbench.py --reps 2 --check 2>&1 | grep -A1 "^rtcsm\|check"
for c in 16 32 64; do for cells in 10240 14336 20480; do DLIOM_BOX_CHUNK=$c DLIOM_BOX_CELLS=$cells timeout 300 python tools/kbench.py --reps 10 --map-scans 20 2>&1 | grep "^rtcsm" | cut -c1-50 | sed "s/^/A5 chunk$c cells$cells: /"; done; done
for c in 32 64; do DLIOM_BOX_CHUNK=$c timeout 300 python tools/kbench.py --reps 10 2>&1 | grep "^rtcsm" | cut -c1-50 | sed "s/^/A4 chunk$c: /"; done
for d in 2 3; do DLIOM_BOX_DEBUG=$d timeout 300 python tools/kbench.py --reps 10 --map-scans 20 2>&1 | grep "^rtcsm" | cut -c1-50 | sed "s/^/A5 debug$d: /"; done
DLIOM_BOX_MIN_LOG2=0 timeout 1200 python -m pytest tests -m gpu -x -q -k "rtcsm or score or golden or front_end or shard" > $OUT/pytest_box.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest_box.log
