#!/bin/bash
# round 5, fifth GPU job: exact leaf counts through the pinned slot (no read-back in ensure_capacity), bench line with the
# config-5 line and the independent IMU-window leg
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r5_run5
mkdir -p $O
DLIOM_LIB=$R/d-liom_amd/ab/libdliom_exp.so DLIOM_TIMING=1 timeout 200 python tools/kbench.py --map-scans 20 --reps 8 2>&1 | grep -E "TIMING insert|^insert" | tail -10 | tee $O/insert_timing.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q -k "insert or front_end or adapter or config2 or grid" 2>&1 | tail -3 | tee $O/gputest_subset.txt
SECONDS=0
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench.py took $SECONDS s"
python3 - <<'PY'
import json
try:
    b = json.loads(open('gpurun_out/r5_run5/bench.json').read().strip().splitlines()[-1])
    print(b['value'], b['ms_per_step'], b['stage_ms_per_scan'], b['kernel_ms_per_scan'], b['roofline']['avg_launch_ms'], b['parity_checked'])
    for k, v in b.get('wref', {}).items():
        print(k, round(v['scans_per_s']), 'x%.1f' % v['speedup_vs_cpu'], {a: (round(c, 7) if isinstance(c, float) else c) for a, c in v['parity_independent_imu_window'].items() if a in ('scans_compared', 'max_translation_difference_m', 'max_rotation_difference_rad', 'ok', 'seconds')})
    print('config5', json.dumps(b.get('config5'))[:1800])
    c = b['cpu_baseline']
    print('cpu', c['value'], c['host_cores_available'], c['fastest_cpu_variant_measured'])
except Exception as e:
    print('bench parse failed', e)
    print(open('gpurun_out/r5_run5/bench.err').read()[-1500:])
PY
