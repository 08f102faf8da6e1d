#!/bin/bash
# round 5, third GPU job: rotation groups shared among the waves of a short rotation block (A/B), insertion host timing,
# parity at the benchmarked sizes, bench line, W-ref kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r5_run3
mkdir -p $O
for v in 1 0 1 0; do
  echo "== DLIOM_BOX_SPLIT=$v"
  DLIOM_LIB=$R/d-liom_amd/ab/libdliom_exp.so DLIOM_BOX_SPLIT=$v timeout 200 python tools/kbench.py --map-scans 20 --reps 20 2>&1 | grep -E "^rtcsm|C="
done | tee $O/split_ab.txt
DLIOM_LIB=$R/d-liom_amd/ab/libdliom_exp.so DLIOM_TIMING=1 timeout 200 python tools/kbench.py --map-scans 20 --reps 8 2>&1 | grep -E "TIMING insert|^insert|refresh" | tail -14 | tee $O/insert_timing.txt
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3 | tee $O/gputest_subset.txt
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import json
try:
    b = json.loads(open('gpurun_out/r5_run3/bench.json').read().strip().splitlines()[-1])
    print(b['value'], b['ms_per_step'], b['stage_ms_per_scan'], b['kernel_ms_per_scan'], b['roofline']['avg_launch_ms'], b['parity_checked'])
    print({k: (v['scans_per_s'] if isinstance(v, dict) and 'scans_per_s' in v else None) for k, v in b.get('wref', {}).items()})
    r = b['roofline']; print({k: r[k] for k in ('frac', 'frac_useful', 'valu_busy_frac', 'valu_instructions_per_pair', 'traffic')})
except Exception as e:
    print('bench parse failed', e)
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o wref -- python $R/tools/wref_full.py > $R/$O/wref_trace.log 2>&1
echo "trace rc=$?"
cd $R
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/wref_kernel_stats.csv && cut -c1-100 $O/wref_kernel_stats.csv | head -40
rm -rf $O/trace
grep '^{' $O/wref_trace.log | head -3 | cut -c1-600
