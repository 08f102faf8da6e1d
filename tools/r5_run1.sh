#!/bin/bash
# round 5, first GPU job: the whole GPU test suite on the changed library (insertion without the verdict wait, pending
# copies / fills carried by the extent pre-pass, the three-barrier chunk scan), then the bench line and host-side timings
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r5_run1
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest.txt 2>&1
tail -5 $O/gputest.txt
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import json
try:
    b = json.loads(open('gpurun_out/r5_run1/bench.json').read().strip().splitlines()[-1])
    print(b['value'], b['ms_per_step'], b['stage_ms_per_scan'], b['kernel_ms_per_scan'], b['roofline']['avg_launch_ms'], b['parity_checked'])
    print({k: (v['scans_per_s'] if isinstance(v, dict) and 'scans_per_s' in v else None) for k, v in b.get('wref', {}).items()})
except Exception as e:
    print('bench parse failed', e)
PY
DLIOM_LIB=$R/d-liom_amd/ab/libdliom_exp.so DLIOM_TIMING=1 timeout 200 python tools/kbench.py --map-scans 20 --reps 5 > $O/kbench_timing.txt 2>&1
tail -30 $O/kbench_timing.txt
