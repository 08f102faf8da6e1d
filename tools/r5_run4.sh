#!/bin/bash
# round 5, fourth GPU job: hipGraph enqueue cost with big arguments, the whole GPU suite on the changed library (atomic-free
# transform kernel, mirror reuse, histogram poll window), the bench line with the config-5 line and the independent
# IMU-window leg, the moving-sensor mirror stream, histogram timing
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r5_run4
mkdir -p $O
(cd tools/ubench && ./graph_latency) 2>&1 | tee $O/graph_latency.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest.txt 2>&1
tail -5 $O/gputest.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err

python3 - <<'PY'
import json
try:
    b = json.loads(open('gpurun_out/r5_run4/bench.json').read().strip().splitlines()[-1])
    print(b['value'], b['ms_per_step'], b['stage_ms_per_scan'], b['kernel_ms_per_scan'], b['roofline']['avg_launch_ms'], b['parity_checked'])
    for k, v in b.get('wref', {}).items():
        print(k, round(v['scans_per_s']), 'x%.1f' % v['speedup_vs_cpu'], {a: (round(c, 7) if isinstance(c, float) else c) for a, c in v['parity_independent_imu_window'].items() if a in ('scans_compared', 'max_translation_difference_m', 'max_rotation_difference_rad', 'ok', 'seconds')})
    print('config5', json.dumps(b.get('config5'))[:1500])
    c = b['cpu_baseline']
    print('cpu', c['value'], c['host_cores_available'], c['fastest_cpu_variant_measured'])
except Exception as e:
    print('bench parse failed', e)
PY
timeout 300 python tools/mirror_window_stream.py 2> $O/mirror.err | tee $O/mirror_window_stream.json
timeout 300 python tools/hist_bench.py --reps 100 2> $O/hist.err | tee $O/hist_bench.json | cut -c1-1500
