#!/bin/bash
# Config 5 (C = 2 352 637, N = 262 144): the wide-pass / big-box experiments of round 6 (results: DESIGN.md 3.1).
cd "$(dirname "$0")/.."
run() { lib=$1; shift; echo "== lib=$lib $*"; env DLIOM_LIB=d-liom_amd/ab/libdliom_$lib.so "$@" python tools/c5bench.py --reps 3 2>&1 | grep -E "config5|check|rror"; }
run exp
run w3 DLIOM_BOX_CELLS=21000 DLIOM_BOX_CHUNK=64
run w3 DLIOM_BOX_CELLS=21400
run t54w3 DLIOM_BOX_CELLS=21000
run t54w3 DLIOM_BOX_CELLS=21000 DLIOM_BOX_CHUNK=64
run t54w3 DLIOM_BOX_CELLS=21000 DLIOM_BOX_CHUNK=48
run t54w3 DLIOM_BOX_CELLS=18000
run t81w3 DLIOM_BOX_CELLS=21000
run t54w3n6 DLIOM_BOX_CELLS=33000
run t54w3n6 DLIOM_BOX_CELLS=33000 DLIOM_BOX_CHUNK=64
run t54w3n6 DLIOM_BOX_CELLS=21000 DLIOM_BOX_NW=4
