#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export DLIOM_LIB=$R/d-liom_amd/ab/libdliom_exp.so
for rep in 1 2; do
for dbg in 0 512 1024 1536; do
  echo "debug $dbg: $(DLIOM_BOX_DEBUG=$dbg timeout 200 python tools/kbench.py --reps 20 --map-scans 20 2>&1 | grep -E '^rtcsm')"
done
done
for dbg in 256 768 1280 1792; do
  echo "== stamps debug $dbg"; DLIOM_BOX_DEBUG=$dbg timeout 200 python tools/box_stamps.py 2>&1 | grep -E "span|^end |^life|^tickets:"
done
