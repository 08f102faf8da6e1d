#!/bin/bash
# Round 3, call 2: where do the score kernel's instructions and time go (experiments library)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/r3_exp2
mkdir -p $OUT
EXP=$R/d-liom_amd/ab/libdliom_exp.so
timeout 600 python -m pytest tests/test_gpu_full_size.py -x -q -k "config5_benchmarked" 2>&1 | tail -5
run() { echo "== $*"; env DLIOM_LIB=$EXP "$@" timeout 120 python tools/kbench.py --reps 10 --map-scans 20 2>&1 | grep -E "^rtcsm|stats" | cut -c1-200; }
run DLIOM_BOX_PCT_A=0 DLIOM_BOX_PCT_B=0 DLIOM_BOX_DEBUG=128
run DLIOM_BOX_PCT_A=0 DLIOM_BOX_PCT_B=0
run DLIOM_BOX_PCT_A=0 DLIOM_BOX_PCT_B=0 DLIOM_BOX_DEBUG=1
run DLIOM_BOX_PCT_A=0 DLIOM_BOX_PCT_B=0 DLIOM_BOX_DEBUG=2
run DLIOM_BOX_PCT_A=0 DLIOM_BOX_PCT_B=0 DLIOM_BOX_DEBUG=3
run DLIOM_BOX_PCT_A=0 DLIOM_BOX_PCT_B=0 DLIOM_BOX_DEBUG=4
run DLIOM_BOX_PCT_A=0 DLIOM_BOX_PCT_B=0 DLIOM_BOX_DEBUG=32
run DLIOM_BOX_PCT_A=0 DLIOM_BOX_PCT_B=0 DLIOM_BOX_CHUNK=24
run DLIOM_BOX_PCT_A=0 DLIOM_BOX_PCT_B=0 DLIOM_BOX_CHUNK=16
run DLIOM_BOX_PCT_A=0 DLIOM_BOX_PCT_B=0 DLIOM_BOX_CHUNK=40
run DLIOM_BOX_PCT_A=0 DLIOM_BOX_PCT_B=0 DLIOM_BOX_CELLS=12288
run DLIOM_BOX_PCT_A=0 DLIOM_BOX_PCT_B=0 DLIOM_BOX_CELLS=16384
cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/kbench.py --reps 5 --map-scans 20"
for D in 0 1 2 3 4; do
  DLIOM_LIB=$EXP DLIOM_BOX_PCT_A=0 DLIOM_BOX_PCT_B=0 DLIOM_BOX_DEBUG=$D timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-include-regex "rtcsm_score_box" --output-format csv -d $OUT/pmc_d$D -o p -- $CMD > $OUT/pmc_d$D.log 2>&1
  echo "debug $D rc=$?"
  python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/pmc_d$D/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("  ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in sorted(acc.items())))
PY
done
