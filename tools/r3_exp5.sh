#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/r3_exp5
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
(time timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err) 2>&1 | grep real
tail -3 $OUT/bench.err
python3 -c "
import json
b=json.load(open('$OUT/bench.json'))
print({k:b[k] for k in ('value','ms_per_step','n_gpus')}, b['kernel_ms_per_scan'])
r=b['roofline']; print({k:r[k] for k in ('frac','frac_useful','valu_instructions_per_pair','avg_launch_ms','traffic','counter_problems')}); print(r['derived'])
print(b.get('wref',{}).get('scans_per_s'), b['parity']['ok'], b['cpu_baseline']['value'])
"
(time DLIOM_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 > $OUT/bench2.json 2> $OUT/bench2.err) 2>&1 | grep real
tail -3 $OUT/bench2.err
python3 -c "
import json
b=json.load(open('$OUT/bench2.json'))
print({k:b[k] for k in ('value','ms_per_step','n_gpus','scaling')}); print(b.get('sharded')); print(b.get('sharded_config5'))
"
