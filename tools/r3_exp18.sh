#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_gpu_fast_csm.py -x -q 2>&1 | tail -5
timeout 300 python tools/fuzz_fast_csm.py --cases 80 --seconds 120 2>&1 | tail -4
python tools/fast_csm_bench.py --full --reps 9
python tools/fast_csm_bench.py --reps 9
python tools/fast_csm_bench.py --full --dense --reps 5
python tools/fast_csm_bench.py --dense --reps 5
