#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/r3_exp4
mkdir -p $OUT
EXP=$R/d-liom_amd/ab/libdliom_exp.so
timeout 300 python tools/kbench.py --reps 10 --map-scans 20 --check 2>&1 | grep -E "^rtcsm|pairs|check|Error|error|assert" | cut -c1-160
run() { echo "== $*"; env DLIOM_LIB=$EXP "$@" timeout 120 python tools/kbench.py --reps 10 --map-scans 20 2>&1 | grep -E "^rtcsm|stats" | cut -c1-230; }
run DLIOM_BOX_DEBUG=128
for NW in 4 5 6 7 8; do for WGS in 4 3 2; do run DLIOM_BOX_NW=$NW DLIOM_BOX_WGS=$WGS; done; done
run DLIOM_BOX_NW=7 DLIOM_BOX_WGS=2 DLIOM_BOX_DEBUG=128
run DLIOM_BOX_NW=7 DLIOM_BOX_WGS=2 DLIOM_BOX_DEBUG=4
cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/kbench.py --reps 5 --map-scans 20"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
head -6 $OUT/trace/*/t_kernel_stats.csv 2>/dev/null || find $OUT/trace -name "*kernel_stats.csv" -exec head -6 {} \;
