#!/bin/bash
# Run on the GPU box (via gpurun).  Produces, under gpurun_out/prof_<tag>/:
#   kernel-trace + stats CSVs of `bench.py` (same command as the bench line, fewer steps)
#   PMC passes (FETCH_SIZE / WRITE_SIZE / TCC hit-miss / SQ / LDS) over the score kernel, separate passes,
#   never combined with tracing (gpurun refuses that)
# Copy the summaries you want judged into profiles/ afterwards (tools/collect_profiles.py).
set -u
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-wref --no-pmc"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1
echo "trace rc=$?"
grep '^{' $OUT/trace.log > $OUT/bench_under_trace.json
i=0
for P in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
         "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
  timeout 300 rocprofv3 --pmc $P --kernel-include-regex "rtcsm_score_box" --output-format csv -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
  echo "pmc pass $i rc=$?"
  i=$((i+1))
done
cd $R
timeout 600 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench rc=$?"
timeout 300 python tools/wref_full.py > $OUT/wref_full.json 2> $OUT/wref.err
timeout 200 python tools/wref.py > $OUT/wref.json 2>> $OUT/wref.err
timeout 100 python tools/wref.py --stages > $OUT/wref_stages.json 2>> $OUT/wref.err
timeout 200 python tools/stream.py > $OUT/stream.json 2> $OUT/stream.err
timeout 200 python tools/stream.py --gentle --imu-noise > $OUT/stream_gentle.json 2>> $OUT/stream.err
find $OUT -name "*.csv" | head -20
# config 5 (128x2048, 5 cm), the fallback kernels and the W-ref kernel trace have their own scripts:
#   tools/r2_config5.sh  tools/r2_fallback.sh  tools/r2_wref_trace.sh
