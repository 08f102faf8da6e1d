#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r3_exp19/bench.json 2> gpurun_out/r3_exp19/bench.err; echo "bench rc=$?"
