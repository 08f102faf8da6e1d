#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2_exp4
mkdir -p $OUT
cd $R
run() { local label=$1; shift; env "$@" timeout 300 python tools/kbench.py --reps 10 > $OUT/k_$label.log 2>&1; echo "$label: $(grep '^rtcsm' $OUT/k_$label.log| cut -c1-70)"; }
DLIOM_SCORE_MAPPING=3 KBENCH_CHECK=64 timeout 600 python tools/kbench.py --reps 2 --check 2>&1 | tail -1
for d in 0 2; do run dbg$d DLIOM_BOX_DEBUG=$d; done
for c in 8 24 32; do run chunk$c DLIOM_BOX_CHUNK=$c; done
for c in 3072 4096; do run cells$c DLIOM_BOX_CELLS=$c; done
for w in 3072 8192; do run waves$w DLIOM_BOX_WAVES=$w; done
