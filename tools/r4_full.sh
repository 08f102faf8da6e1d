#!/bin/bash
# round 4: the whole GPU suite, then the bench line (driver contract) -- outputs under gpurun_out/r4_full
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/r4_full
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -s > $OUT/gputest.log 2>&1
tail -5 $OUT/gputest.log
grep -h "windowed mirror\|fuzz ok\|round-3 fuzz ok" $OUT/gputest.log | head
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json; tail -3 $OUT/bench.err
