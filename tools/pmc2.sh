#!/bin/bash
# usage: tools/pmc2.sh <outdir> <kernel-regex> -- <command...>   (SQ/TA/LDS oriented passes)
set -u
OUT=$1; REGEX=$2; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$OUT
cd /tmp; export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"
 "SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
 "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum"
 "SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_BUSY_CU_CYCLES"
)
i=0
for P in "${PASSES[@]}"; do
  timeout 180 rocprofv3 --pmc $P --kernel-include-regex "$REGEX" --output-format csv -d $R/gpurun_out/$OUT/pass$i -o p -- "$@" > $R/gpurun_out/$OUT/pass$i.log 2>&1
  echo "pass $i rc=$?"
  i=$((i+1))
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in sorted(glob.glob('$R/gpurun_out/$OUT/pass*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()):
    print("%-36s n=%d mean=%.5g"%(k,len(v),sum(v)/len(v)))
PY
