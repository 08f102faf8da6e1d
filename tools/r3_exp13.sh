#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_gpu_full_size.py -x -q -k "config2_full or every_score_kernel or eight_candidate" 2>&1 | tail -3
timeout 200 python tools/kbench.py --reps 10 --map-scans 20 2>&1 | grep -E "^rtcsm|C="
DLIOM_LIB=$R/d-liom_amd/ab/libdliom_exp.so DLIOM_BOX_DEBUG=256 timeout 200 python tools/box_stamps.py 2>&1 | tail -22
