#!/bin/bash
# Config 5 (C = 2 352 637, N = 262 144): the wide-pass / big-box experiments of round 6 (results: DESIGN.md 3.1).
# Libraries: make -C d-liom_amd experiments EXP_NAME=<name> EXP_FLAGS="<flags>" (the names below say which flags).
cd "$(dirname "$0")/.."
run() { lib=$1; shift; echo "== lib=$lib $*"; env DLIOM_LIB=d-liom_amd/ab/libdliom_$lib.so "$@" python tools/c5bench.py --reps 3 2>&1 | grep -E "config5|check|rror"; }
run tw54
run tw49
run tw54 DLIOM_BOX_VARIANT=1
run tw54 DLIOM_BOX_VARIANT=0
run tw49
run tw54
