#!/bin/bash
# PMC passes over one command (each pass = its own rocprofv3 run; never combined with tracing).
# usage: tools/pmc.sh <outdir-under-gpurun_out> <kernel-regex> -- <command...>
set -u
OUT=$1; REGEX=$2; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$OUT
cd /tmp; export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"
 "SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
 "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
 "TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
 "FETCH_SIZE"
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
)
i=0
for P in "${PASSES[@]}"; do
  timeout 180 rocprofv3 --pmc $P --kernel-include-regex "$REGEX" --output-format csv -d $R/gpurun_out/$OUT/pass$i -o p -- "$@" > $R/gpurun_out/$OUT/pass$i.log 2>&1
  echo "pass $i rc=$?"
  i=$((i+1))
done
find $R/gpurun_out/$OUT -name "*counter_collection.csv" | head
