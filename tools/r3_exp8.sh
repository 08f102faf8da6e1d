#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python tools/wref_full.py > gpurun_out/r3_wref_full.json 2> gpurun_out/r3_wref_full.err; tail -5 gpurun_out/r3_wref_full.err
python3 -c "
import json
d=json.load(open('gpurun_out/r3_wref_full.json'))
for k,v in d.items():
    print(k, round(v['scans_per_s'],1), 'scans/s', {a: round(b,3) for a,b in v['p50_ms'].items()}, 'gfac', v['gravity_factors_added'])
    print('   cpu', round(v['cpu_baseline']['value'],2), {a: round(b,2) for a,b in v['cpu_baseline']['p50_ms'].items()}, 'speedup', round(v['speedup_vs_cpu'],1), v['parity'])
"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "adapter" 2>&1 | tail -4
