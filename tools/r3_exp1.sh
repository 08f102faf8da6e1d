#!/bin/bash
# Round 3, call 1: new scheduler (guided tickets + unit hopping) -- parity at full size, then A/B timings.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/r3_exp1
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_sharded.py -x -q 2>&1 | tail -15 > $OUT/tests.log
cat $OUT/tests.log
run() { echo "== $*"; env "$@" timeout 120 python tools/kbench.py --reps 10 --map-scans 20 2>&1 | grep -E "^rtcsm|pairs" | cut -c1-100; }
echo "--- production library"
run X=1
EXP=$R/d-liom_amd/ab/libdliom_exp.so
echo "--- experiments library"
run DLIOM_LIB=$EXP
run DLIOM_LIB=$EXP DLIOM_BOX_DEBUG=64
run DLIOM_LIB=$EXP DLIOM_BOX_DEBUG=64 DLIOM_BOX_PCT_A=100
run DLIOM_LIB=$EXP DLIOM_BOX_PCT_A=100
run DLIOM_LIB=$EXP DLIOM_BOX_PCT_A=0 DLIOM_BOX_PCT_B=0
run DLIOM_LIB=$EXP DLIOM_BOX_PCT_A=25 DLIOM_BOX_PCT_B=50
run DLIOM_LIB=$EXP DLIOM_BOX_PCT_A=75 DLIOM_BOX_PCT_B=50
run DLIOM_LIB=$EXP DLIOM_BOX_NW=3
run DLIOM_LIB=$EXP DLIOM_BOX_NW=3 DLIOM_BOX_CELLS=11264
run DLIOM_LIB=$EXP DLIOM_BOX_CHUNK=16
run DLIOM_LIB=$EXP DLIOM_BOX_CHUNK=48 DLIOM_BOX_CELLS=16384
echo "--- PMC (production library)"
cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/kbench.py --reps 5 --map-scans 20"
i=0
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  timeout 200 rocprofv3 --pmc $P --kernel-include-regex "rtcsm_score_box" --output-format csv -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
  echo "pmc pass $i rc=$?"
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/pmc*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            print(k, len(v), sum(v) / len(v))
PY
