#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests/test_gpu_full_size.py -x -q -k "ragged" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "adapter or front_end" 2>&1 | tail -3
timeout 300 python tools/wref_full.py > gpurun_out/r3_exp25/wref_full.json 2> gpurun_out/r3_exp25/wref_full.err; echo "wref rc=$?"
python3 - <<PY
import json
w=json.load(open("gpurun_out/r3_exp25/wref_full.json"))
for k,v in w.items(): print(k, v["scans_per_s"], v["p50_ms"], v["speedup_vs_cpu"], v["parity"]["ok"], v["parity"]["histograms_max_abs_difference"])
PY
