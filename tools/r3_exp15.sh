#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export DLIOM_LIB=$R/d-liom_amd/ab/libdliom_w8.so
echo "nw 4: $(DLIOM_BOX_NW=4 timeout 200 python tools/kbench.py --reps 10 --map-scans 20 2>&1 | grep -E '^rtcsm')"
for nw in 7 8; do
for cells in 20480 28672 32768; do
for chunk in 32 48 64; do
  echo "nw $nw cells $cells chunk $chunk: $(DLIOM_BOX_NW=$nw DLIOM_BOX_CELLS=$cells DLIOM_BOX_CHUNK=$chunk timeout 200 python tools/kbench.py --reps 10 --map-scans 20 2>&1 | grep -E '^rtcsm')"
done
done
done
echo "nw 4: $(DLIOM_BOX_NW=4 timeout 200 python tools/kbench.py --reps 10 --map-scans 20 2>&1 | grep -E '^rtcsm')"
