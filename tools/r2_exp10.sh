#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { echo "== $*"; env "$@" timeout 100 python tools/kbench.py --reps 8 --map-scans 20 --check 2>&1 | grep -E "^rtcsm|check" | cut -c1-75 | tr '\n' ' '; echo; }
run DLIOM_BOX_NW=4
run DLIOM_BOX_NW=6 DLIOM_BOX_CELLS=14336
run DLIOM_BOX_NW=7 DLIOM_BOX_CELLS=14336
run DLIOM_BOX_NW=7 DLIOM_BOX_CELLS=18432
run DLIOM_BOX_NW=8 DLIOM_BOX_CELLS=14336
run DLIOM_BOX_NW=8 DLIOM_BOX_CELLS=18432
run DLIOM_BOX_NW=8 DLIOM_BOX_CELLS=24576
run DLIOM_BOX_NW=7 DLIOM_BOX_CELLS=18432 DLIOM_BOX_CHUNK=64
