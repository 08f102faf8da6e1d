#!/bin/bash
# Round 3, call 3: plan kernel + double-buffered LDS-DMA box kernel: parity, timing, counters
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/r3_exp3
mkdir -p $OUT
EXP=$R/d-liom_amd/ab/libdliom_exp.so
timeout 300 python tools/kbench.py --reps 10 --map-scans 20 --check 2>&1 | grep -E "^rtcsm|pairs|check|Error|error|assert" | cut -c1-160
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q 2>&1 | tail -8
run() { echo "== $*"; env DLIOM_LIB=$EXP "$@" timeout 120 python tools/kbench.py --reps 10 --map-scans 20 2>&1 | grep -E "^rtcsm|stats" | cut -c1-220; }
run DLIOM_BOX_DEBUG=128
run X=1
run DLIOM_BOX_DEBUG=4
run DLIOM_BOX_DEBUG=1
run DLIOM_BOX_DEBUG=64
run DLIOM_BOX_DEBUG=5
run DLIOM_BOX_CELLS=6400
run DLIOM_BOX_CELLS=5120
run DLIOM_BOX_CHUNK=24
run DLIOM_BOX_CHUNK=16
run DLIOM_BOX_NW=3
cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/kbench.py --reps 5 --map-scans 20"
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-include-regex "rtcsm_score_box|rtcsm_box_plan" --output-format csv -d $OUT/pmc0 -o p -- $CMD > $OUT/pmc0.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-include-regex "rtcsm_score_box|rtcsm_box_plan" --output-format csv -d $OUT/pmc1 -o p -- $CMD > $OUT/pmc1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("pmc0", "pmc1"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            acc[(row["Kernel_Name"][:40], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(k, len(v), "%.4g" % (sum(v) / len(v)))
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:2500])
PY
