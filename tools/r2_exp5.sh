#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2_exp5
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for D in 0 4 5 7; do
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE"; do
  i=$(echo $P | md5sum | cut -c1-6)
  DLIOM_BOX_DEBUG=$D timeout 300 rocprofv3 --pmc $P --kernel-include-regex "rtcsm_score_box" --output-format csv -d $OUT/pmc${D}_$i -o p -- python $R/tools/kbench.py --reps 3 --map-scans 20 > $OUT/pmc${D}_$i.log 2>&1
done
echo "== debug $D"
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('/root/repo/gpurun_out/r2_exp5/pmc${D}_*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'score_box' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(sum(v)/len(v)/1e6,1) for k,v in sorted(acc.items())})
PY
done
