#!/bin/bash
# round 5, ninth GPU job: the de-skew records' launch tags, config 5 as a run of its own (CPU legs sampled there)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/prof_r5
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "deskew or range_data or front_end or fuzz" 2>&1 | tail -3
SECONDS=0
timeout 900 python bench.py --config 5 --no-wref > $O/config5_bench.json 2> $O/config5_bench.err; echo "config5 rc=$? in $SECONDS s"
python3 - <<'PY'
import json
b = json.loads(open('gpurun_out/prof_r5/config5_bench.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], b['config'], b['stage_ms_per_scan'])
r = b['roofline']; print({k: r[k] for k in ('frac', 'frac_useful', 'valu_busy_frac', 'valu_instructions_per_pair', 'avg_launch_ms', 'pairs_per_s')})
print(b['parity']); c = b['cpu_baseline']; print(c['value'], c['sample'][:200], c['host_cpu_quota'])
PY
