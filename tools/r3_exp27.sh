#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/wref_full.py > gpurun_out/r3_exp27/wref_full.json 2> gpurun_out/r3_exp27/wref_full.err; echo "wref rc=$?"
python3 - <<PY
import json
w=json.load(open("gpurun_out/r3_exp27/wref_full.json"))
for k,v in w.items(): print(k, v["scans_per_s"], v["p50_ms"], v["speedup_vs_cpu"], v["parity"]["ok"], v["parity"]["histograms_max_abs_difference"])
PY
timeout 300 python bench.py --no-pmc --no-wref 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print(b['value'], b['ms_per_step'], b['stage_ms_per_scan'], b['kernel_ms_per_scan'], b['parity_checked'])
"
timeout 100 python tools/wref.py --stages 2>/dev/null | tail -1
timeout 100 python tools/wref.py 2>/dev/null | tail -1
python tools/fast_csm_bench.py --full --reps 9 --no-cpu 2>/dev/null | tail -1
python tools/fast_csm_bench.py --reps 9 --no-cpu 2>/dev/null | tail -1
