#!/bin/bash
# BASELINE config 5 (128x2048 scan, 5 cm voxels: bits = 4 dense mirror, T = 343, R = 4913) on the GPU box:
# plain bench line, rocprofv3 kernel trace + stats, and PMC passes (L2 hit/miss, HBM traffic, SQ) over the score kernel.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/config5
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --beams 128 --azimuths 2048 --high-resolution 0.05 --steps 3 --warmup 1 --no-cpu-baseline --no-wref"
timeout 300 $CMD > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o c5 -- $CMD > $OUT/trace.log 2>&1; echo "trace rc=$?"
i=0
for P in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  timeout 400 rocprofv3 --pmc $P --kernel-include-regex "rtcsm_score" --output-format csv -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
  echo "pmc pass $i rc=$?"
  i=$((i+1))
done
python - <<PY
import csv,glob,collections,json
acc=collections.defaultdict(list)
for f in glob.glob('$OUT/pmc*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rtcsm_score' in r['Kernel_Name']: acc[r['Kernel_Name'].split('(')[0]+':'+r['Counter_Name']].append(float(r['Counter_Value']))
json.dump({k: {"launches": len(v), "mean": sum(v)/len(v)} for k,v in sorted(acc.items())}, open('$OUT/pmc_summary.json','w'), indent=1)
PY
find $OUT -name "*kernel_stats.csv" | head; rm -rf $OUT/pmc*/*/*.db 2>/dev/null
head -c 1500 $OUT/bench.json
