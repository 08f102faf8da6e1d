#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2_exp3
mkdir -p $OUT
cd $R
run() { # label env...
  local label=$1; shift
  env "$@" timeout 300 python tools/kbench.py --reps 10 > $OUT/k_$label.log 2>&1
  echo "$label: $(grep '^rtcsm' $OUT/k_$label.log)"
}
run base DLIOM_SCORE_MAPPING=3
for c in 2048 3072 6144; do run cells$c DLIOM_BOX_CELLS=$c; done
for c in 8 16 64; do run chunk$c DLIOM_BOX_CHUNK=$c; done
for w in 4096 6144 12288 16384; do run waves$w DLIOM_BOX_WAVES=$w; done
DLIOM_SCORE_MAPPING=3 KBENCH_CHECK=64 timeout 600 python tools/kbench.py --reps 2 --check 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
         "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH"; do
  i=$(echo $P | md5sum | cut -c1-6)
  DLIOM_BOX_CELLS=4096 timeout 300 rocprofv3 --pmc $P --kernel-include-regex "rtcsm_score_box" --output-format csv -d $OUT/pmc_$i -o p -- python $R/tools/kbench.py --reps 3 > $OUT/pmc_$i.log 2>&1
  echo "pmc $i rc=$?"
done
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('/root/repo/gpurun_out/r2_exp3/pmc_*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'score_box' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k, sum(v)/len(v), len(v))
PY
