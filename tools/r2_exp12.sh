#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { echo "== $*"; env "$@" timeout 100 python tools/kbench.py --reps 10 --map-scans 20 2>&1 | grep -E "^rtcsm" | cut -c1-60; }
run DLIOM_BOX_CHUNK=24
run DLIOM_BOX_CHUNK=40
run DLIOM_BOX_CHUNK=48
run DLIOM_BOX_CHUNK=40 DLIOM_BOX_CELLS=16384
run DLIOM_BOX_CELLS=12288
run DLIOM_BOX_CHUNK=32
