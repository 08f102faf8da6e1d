#!/bin/bash
# The score-volume fallbacks on the bench workload: dense-mirror kernel (mapping 2) and the leaf-table kernels
# (mapping 1 / 0, what grids beyond bits = 4 use): kernel time under rocprofv3 and L2 / HBM counters.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/fallback
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for M in 2 1 0; do
  CMD="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-wref"
  DLIOM_SCORE_MAPPING=$M timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$M -o t -- $CMD > $OUT/trace$M.log 2>&1
  DLIOM_SCORE_MAPPING=$M timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "rtcsm_score" --output-format csv -d $OUT/pmc$M -o p -- $CMD > $OUT/pmc$M.log 2>&1
  grep '^{' $OUT/trace$M.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mapping $M:', d['value'], 'scans/s', d['kernel_ms_per_scan']['rtcsm_score'], 'ms', d['roofline']['kernel'])"
  f=$(find $OUT/trace$M -name "*kernel_stats.csv" | head -1); grep rtcsm_score $f | cut -c1-80,200-260
done
python - <<PY
import csv,glob,collections,json
out={}
for M in (2,1,0):
    acc=collections.defaultdict(list)
    for f in glob.glob('$OUT/pmc%d/**/*counter_collection.csv'%M,recursive=True):
        for r in csv.DictReader(open(f)):
            if 'rtcsm_score' in r['Kernel_Name']: acc[r['Kernel_Name'].split('(')[0].split('<')[0]+':'+r['Counter_Name']].append(float(r['Counter_Value']))
    out['mapping_%d'%M]={k: sum(v)/len(v) for k,v in sorted(acc.items())}
json.dump(out, open('$OUT/pmc_summary.json','w'), indent=1); print(json.dumps(out, indent=1))
PY
