#!/bin/bash
# where the box kernel's non-lookup instructions go: work counters + SQ_INSTS_VALU with parts of the kernel switched off
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/r3_exp11
mkdir -p $OUT
export DLIOM_LIB=$R/d-liom_amd/ab/libdliom_exp.so
DLIOM_BOX_DEBUG=128 timeout 200 python tools/kbench.py --reps 5 --map-scans 20 2>&1 | grep -E "^rtcsm|stats|C="
cd /tmp; export TMPDIR=/tmp
for dbg in 0 1 4 6 7; do
  DLIOM_BOX_DEBUG=$dbg timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-include-regex rtcsm_score_box_kernel --output-format csv -d $OUT/pmc$dbg -o p -- python $R/tools/kbench.py --reps 3 --map-scans 20 > $OUT/pmc$dbg.log 2>&1
  python3 - <<PY
import csv,glob
acc={}
for f in glob.glob("$OUT/pmc$dbg/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)): acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
print("debug $dbg:", {k: "%.4g" % (sum(v)/len(v)) for k,v in acc.items()}, "launches", len(next(iter(acc.values()),[])))
PY
  grep -E "^rtcsm" $OUT/pmc$dbg.log
done
