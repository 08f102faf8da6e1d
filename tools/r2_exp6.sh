#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2_exp6
mkdir -p $OUT
cd $R
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json
d=json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['kernel_ms_per_scan'], d['roofline']['launches'])"
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_all.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest_all.log
timeout 300 python tools/wref.py 2>&1 | tail -2
