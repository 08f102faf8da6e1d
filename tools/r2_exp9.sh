#!/bin/bash
# Box score kernel: waves per workgroup / box capacity sweep (smaller workgroups = smaller rotation spread = smaller
# boxes, more workgroups per CU, fewer waves coupled by a barrier).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { echo "== $*"; env "$@" timeout 100 python tools/kbench.py --reps 8 --map-scans 20 --check 2>&1 | grep -E "^rtcsm|check" | cut -c1-75 | tr '\n' ' '; echo; }
run DLIOM_BOX_NW=0
run DLIOM_BOX_NW=1 DLIOM_BOX_CELLS=4096
run DLIOM_BOX_NW=1 DLIOM_BOX_CELLS=6144
run DLIOM_BOX_NW=1 DLIOM_BOX_CELLS=8192
run DLIOM_BOX_NW=2 DLIOM_BOX_CELLS=8192
run DLIOM_BOX_NW=2 DLIOM_BOX_CELLS=10240
run DLIOM_BOX_NW=2 DLIOM_BOX_CELLS=14336
run DLIOM_BOX_NW=4 DLIOM_BOX_CELLS=14336
run DLIOM_BOX_NW=1 DLIOM_BOX_CELLS=6144 DLIOM_BOX_CHUNK=64
run DLIOM_BOX_NW=1 DLIOM_BOX_CELLS=4096 DLIOM_BOX_CHUNK=16
