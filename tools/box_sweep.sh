#!/bin/bash
# A/B sweep of the LDS-box score kernel in ONE process chain on ONE box (round 6): experiments builds
# (make -C d-liom_amd experiments EXP_NAME=exp ; ... EXP_NAME=w3 EXP_FLAGS=-DDLIOM_BOX_WPE=3) x chunk size x box cells.
#   gpurun -- 'bash tools/box_sweep.sh > gpurun_out/r6_box_sweep.txt 2>&1'
cd "$(dirname "$0")/.."
run() {  # lib, then env assignments
  lib=$1; shift
  echo "== lib=$lib $*"
  env DLIOM_LIB=d-liom_amd/ab/libdliom_$lib.so "$@" python tools/kbench.py --map-scans 20 --reps 10 --check 2>&1 | grep -E "rtcsm|C=|check|stats|rror" 
}
for rep in 1 2; do
run exp
run exp DLIOM_BOX_DEBUG=512
run exp DLIOM_BOX_DEBUG=512 DLIOM_BOX_CHUNK=64
run exp DLIOM_BOX_DEBUG=512 DLIOM_BOX_CHUNK=48
run w3
run w3 DLIOM_BOX_CELLS=21000
run w3 DLIOM_BOX_DEBUG=512 DLIOM_BOX_CELLS=21000
run w3 DLIOM_BOX_DEBUG=512 DLIOM_BOX_CELLS=21000 DLIOM_BOX_CHUNK=64
run w3 DLIOM_BOX_DEBUG=512 DLIOM_BOX_CELLS=21000 DLIOM_BOX_CHUNK=48
run w3 DLIOM_BOX_DEBUG=512 DLIOM_BOX_CELLS=14336 DLIOM_BOX_CHUNK=64
done
