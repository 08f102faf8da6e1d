#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { echo "== $*"; env "$@" timeout 100 python tools/kbench.py --reps 5 --map-scans 20 2>&1 | grep "^rtcsm" ; }
run DLIOM_BOX_DEBUG=0
run DLIOM_BOX_DEBUG=32
run DLIOM_BOX_DEBUG=6
run DLIOM_BOX_DEBUG=22
run DLIOM_BOX_DEBUG=14
run DLIOM_BOX_DEBUG=30
run DLIOM_BOX_DEBUG=46
