#!/bin/bash
# Box score kernel: where does the time go?  DLIOM_BOX_DEBUG timing variants (wrong sums by design) and the
# launch-shape knobs, on the bench workload (C = 35 937, N = 65 536).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { echo "== $*"; env "$@" timeout 100 python tools/kbench.py --reps 5 --map-scans 20 2>&1 | grep -A1 "^rtcsm" | tr '\n' ' '; echo; }
run DLIOM_BOX_DEBUG=0
run DLIOM_BOX_DEBUG=1
run DLIOM_BOX_DEBUG=2
run DLIOM_BOX_DEBUG=3
run DLIOM_BOX_DEBUG=4
run DLIOM_BOX_DEBUG=6
run DLIOM_BOX_CELLS=10240
run DLIOM_BOX_CELLS=20480
run DLIOM_BOX_CHUNK=16
run DLIOM_BOX_CHUNK=64
run DLIOM_BOX_CHUNK=64 DLIOM_BOX_CELLS=20480
run DLIOM_BOX_WAVES=2048
run DLIOM_BOX_WAVES=4096
