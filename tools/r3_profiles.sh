#!/bin/bash
# Round 3: everything profiles/r3_* is made of.  Run on the GPU box (gpurun), then tools/collect_profiles.py r3.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/profile_round.sh r3
OUT=$R/gpurun_out/prof_r3
timeout 600 python bench.py --config 5 > $OUT/config5_bench.json 2> $OUT/config5_bench.err; echo "config5 rc=$?"
DLIOM_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 2> $OUT/bench_gloo2.err | grep '^{' > $OUT/bench_gloo2.json; echo "gloo2 rc=$?"
timeout 200 python tools/fast_csm_bench.py --reps 9 > $OUT/fast_csm.json 2> $OUT/fast_csm.err
timeout 200 python tools/fast_csm_bench.py --full --reps 9 > $OUT/fast_csm_full.json 2>> $OUT/fast_csm.err
timeout 300 python tools/fast_csm_bench.py --dense --reps 5 > $OUT/fast_csm_dense.json 2>> $OUT/fast_csm.err
bash tools/r3_exp6.sh > $OUT/histogram.txt 2>&1
ls -la $OUT | head -40
