#!/bin/bash
# Round 4: everything profiles/r4_* is made of.  Run on the GPU box (gpurun), then tools/collect_profiles.py r4.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/profile_round.sh r4
OUT=$R/gpurun_out/prof_r4
timeout 400 python bench.py --config 5 --no-wref > $OUT/config5_bench.json 2> $OUT/config5_bench.err; echo "config5 rc=$?"
timeout 300 python bench.py --gpus 1 --shard-candidates --steps 10 --no-pmc --no-wref --no-cpu-baseline 2> $OUT/bench_shard1.err | grep '^{' > $OUT/bench_rccl_1rank.json; echo "rccl 1 rank rc=$?"
DLIOM_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --no-wref 2> $OUT/bench_gloo2.err | grep '^{' > $OUT/bench_gloo2.json; echo "gloo2 rc=$?"
timeout 200 python tools/fast_csm_bench.py --reps 9 > $OUT/fast_csm.json 2> $OUT/fast_csm.err
timeout 200 python tools/fast_csm_bench.py --full --reps 9 > $OUT/fast_csm_full.json 2>> $OUT/fast_csm.err
# ComputeHistogram: both scene families, time per call + equality with the oracle, then the kernels under rocprofv3
timeout 300 python tools/hist_bench.py --check > $OUT/hist_bench.json 2> $OUT/hist_bench.err
cd /tmp; export TMPDIR=/tmp
{
  cat $OUT/hist_bench.json
  for s in cube_64x1024 yard_64x1024_level yard_64x1024_raw; do
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hist_trace_$s -o t -- python $R/tools/hist_bench.py --only $s --reps 100 > $OUT/hist_trace_$s.log 2>&1
    echo "== $s (rocprofv3 --kernel-trace --stats, 105 calls)"
    python3 - <<PY
import glob
for f in glob.glob("$OUT/hist_trace_$s/**/*kernel_stats.csv", recursive=True):
    for l in open(f).read().splitlines()[:12]: print(l[:260])
PY
  done
} > $OUT/histogram.txt 2>&1
# the complete W-ref chain on the yard scene under the kernel trace (launches per scan, GPU time per scan)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wref_trace -o w -- python $R/tools/wref_full.py --no-cpu --options trajectory_builder_3d --scans 24 > $OUT/wref_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wref_trace_yard -o w -- python $R/tools/wref_full.py --no-cpu --options trajectory_builder_3d --scans 24 --scene ground > $OUT/wref_trace_yard.log 2>&1
cd $R
if [ -n "$R4_TEST_SUBSET" ]; then  # (a refresh of the numbers after a small change: the tests that cover it)
  timeout 600 python -m pytest tests -q -m gpu --timeout 600 -k "$R4_TEST_SUBSET" > $OUT/gputest_subset.log 2>&1; tail -3 $OUT/gputest_subset.log
else
  timeout 900 python -m pytest tests -q -m gpu --timeout 600 -s > $OUT/gputest.log 2>&1; tail -3 $OUT/gputest.log
fi
ls $OUT | head -60
