#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q -k "csm3d or ceres or front_end or adapter" 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --no-pmc --no-wref --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print(b['value'], b['ms_per_step'], b['stage_ms_per_scan'], b['kernel_ms_per_scan'])
"
done
