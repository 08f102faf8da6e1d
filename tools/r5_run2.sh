#!/bin/bash
# round 5, second GPU job: box plan (A/B inside the experiments build), insertion proof with per-axis extents, GPU tests,
# hipGraph latency, kernel trace of the bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r5_run2
mkdir -p $O
for v in 1 0 1 0; do
  echo "== DLIOM_BOX_USE_PLAN=$v"
  DLIOM_LIB=$R/d-liom_amd/ab/libdliom_exp.so DLIOM_BOX_USE_PLAN=$v timeout 200 python tools/kbench.py --map-scans 20 --reps 20 2>&1 | grep -E "^rtcsm|C="
done | tee $O/plan_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest.txt 2>&1
tail -5 $O/gputest.txt
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import json
try:
    b = json.loads(open('gpurun_out/r5_run2/bench.json').read().strip().splitlines()[-1])
    print(b['value'], b['ms_per_step'], b['stage_ms_per_scan'], b['kernel_ms_per_scan'], b['roofline']['avg_launch_ms'], b['parity_checked'])
    print({k: (v['scans_per_s'] if isinstance(v, dict) and 'scans_per_s' in v else None) for k, v in b.get('wref', {}).items()})
    c = b['cpu_baseline']
    print('cpu', c['value'], c['host_cores_available'], {k: v['value'] for k, v in c['reference_layout'].items() if isinstance(v, dict)}, {k: v['value'] for k, v in c['fair_cpu'].items() if isinstance(v, dict)})
except Exception as e:
    print('bench parse failed', e)
PY
(cd tools/ubench && ./graph_latency) 2>&1 | tee $O/graph_latency.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-wref --no-pmc --no-rccl-check > $R/$O/trace.log 2>&1
echo "trace rc=$?"
cd $R
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/kernel_stats.csv && cut -c1-110 $O/kernel_stats.csv | head -30
rm -rf $O/trace
