#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export DLIOM_LIB=$R/d-liom_amd/ab/libdliom_exp.so
DLIOM_BOX_DEBUG=256 timeout 200 python tools/box_stamps.py 2>&1 | tail -30
