#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "rtcsm" 2>&1 | tail -3
for rep in 1 2; do
  echo "new : $(timeout 200 python tools/kbench.py --reps 20 --map-scans 20 2>&1 | grep -E '^rtcsm')"
  echo "prev: $(DLIOM_LIB=$R/d-liom_amd/ab/libdliom_prev.so timeout 200 python tools/kbench.py --reps 20 --map-scans 20 2>&1 | grep -E '^rtcsm')"
done
