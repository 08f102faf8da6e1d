#!/bin/bash
# round 5, sixth GPU job: the whole GPU suite (de-skew check, pinned leaf counts, config-5 line), the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r5_run6
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest.txt 2>&1
tail -6 $O/gputest.txt
SECONDS=0
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench.py took $SECONDS s"
python3 - <<'PY'
import json
try:
    b = json.loads(open('gpurun_out/r5_run6/bench.json').read().strip().splitlines()[-1])
    print(b['value'], b['ms_per_step'], b['stage_ms_per_scan'], b['kernel_ms_per_scan'], b['roofline']['avg_launch_ms'], b['parity_checked'])
    for k, v in b.get('wref', {}).items():
        p = v['parity_independent_imu_window']
        print(k, round(v['scans_per_s']), 'x%.1f' % v['speedup_vs_cpu'], v['p50_ms'], 'same-inputs %.3g m / closed-loop %.3g m' % (p['same_inputs']['max_translation_difference_m'], p['closed_loop']['max_translation_difference_m']), p['ok'])
    print('config5', json.dumps(b.get('config5'))[:1800])
except Exception as e:
    print('bench parse failed', e)
    print(open('gpurun_out/r5_run6/bench.err').read()[-1500:])
PY
