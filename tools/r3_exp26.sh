#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fast_csm.py -x -q 2>&1 | tail -3
timeout 300 python tools/wref_full.py > gpurun_out/r3_exp26/wref_full.json 2> gpurun_out/r3_exp26/wref_full.err; echo "wref rc=$?"
python3 - <<PY
import json
w=json.load(open("gpurun_out/r3_exp26/wref_full.json"))
for k,v in w.items(): print(k, v["scans_per_s"], v["p50_ms"], v["speedup_vs_cpu"], v["parity"]["ok"], v["parity"]["histograms_max_abs_difference"])
PY
timeout 300 python bench.py --no-pmc --no-wref --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print(b['value'], b['ms_per_step'], b['stage_ms_per_scan'], b['kernel_ms_per_scan'])
"
timeout 100 python tools/wref.py --stages 2>/dev/null | tail -1
