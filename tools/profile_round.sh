#!/bin/bash
# Everything profiles/<tag>_* is made of.  Run on the GPU box (gpurun -- 'bash tools/profile_round.sh r5'), then
# `python tools/collect_profiles.py r5` here.  Writes under gpurun_out/prof_<tag>/:
#   trace     kernel trace + stats of bench.py (same command as the bench line, no CPU legs)
#   pmc       PMC passes over the score kernel (separate passes, never combined with tracing: gpurun refuses that)
#   wreftrace the W-ref chains under the kernel trace (launches per scan, GPU time per scan), both scenes
#   bench     the full bench line (what the driver runs)
#   config5   config 5 as its own run
#   gloo2     the N = 2 control run over gloo
#   wref      W-ref chains, stages, streaming config 3
#   loop      loop closure matcher, the moving-sensor mirror stream, hipGraph / launch latency
#   hist      ComputeHistogram on the device against the oracle, per scene
#   voxel     the voxel filter's kernels under the kernel trace
#   tests     the GPU test log
#   wrefcpp   W-ref through the C++ adapters, timed in C++ (tools/wref_cpp.py)
#   pmc5      PMC passes over config 5's score kernel (tools/c5bench.py: the wide instantiation)
#   imucost   WindowOptimize's host cost per scan by mode (tools/imu_window_cost.py)
# A second argument selects parts (quoted, space separated); without it everything runs and the directory starts empty.
set -u
TAG=${1:-r5}
PARTS=${2:-all}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
if [ "$PARTS" = all ]; then rm -rf $OUT; fi
mkdir -p $OUT
want() { [ "$PARTS" = all ] || [[ " $PARTS " == *" $1 "* ]]; }
cd /tmp; export TMPDIR=/tmp
if want trace; then
  rm -rf $OUT/trace
  CMD="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-wref --no-pmc --no-config5 --no-rccl-check"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1
  echo "trace rc=$?"
  grep '^{' $OUT/trace.log > $OUT/bench_under_trace.json
fi
if want pmc; then
  i=0
  for P in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
    timeout 300 rocprofv3 --pmc $P --kernel-include-regex "rtcsm_score_box" --output-format csv -d $OUT/pmc$i -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-wref --no-pmc --no-config5 --no-rccl-check > $OUT/pmc$i.log 2>&1
    echo "pmc pass $i rc=$?"
    i=$((i+1))
  done
fi
if want wreftrace; then
  rm -rf $OUT/wref_trace $OUT/wref_trace_yard
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wref_trace -o w -- python $R/tools/wref_full.py --no-cpu --options trajectory_builder_3d --scans 24 > $OUT/wref_trace.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wref_trace_yard -o w -- python $R/tools/wref_full.py --no-cpu --options trajectory_builder_3d --scans 24 --scene ground > $OUT/wref_trace_yard.log 2>&1
fi
if want voxel; then
  rm -rf $OUT/voxel_trace
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/voxel_trace -o v -- python $R/tools/vf_bench.py > $OUT/voxel_filter.json 2> $OUT/voxel_filter.err
fi
cd $R
if want bench; then
  SECONDS=0
  timeout 600 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
  echo "bench rc=$? in $SECONDS s"
fi
if want config5; then timeout 500 python bench.py --config 5 --no-wref > $OUT/config5_bench.json 2> $OUT/config5_bench.err; echo "config5 rc=$?"; fi
if want gloo2; then DLIOM_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --no-wref 2> $OUT/bench_gloo2.err | grep '^{' > $OUT/bench_gloo2.json; echo "gloo2 rc=$?"; fi
if want wref; then
  timeout 300 python tools/wref_full.py > $OUT/wref_full.json 2> $OUT/wref.err
  timeout 200 python tools/wref.py > $OUT/wref.json 2>> $OUT/wref.err
  timeout 100 python tools/wref.py --stages > $OUT/wref_stages.json 2>> $OUT/wref.err
  timeout 200 python tools/stream.py > $OUT/stream.json 2> $OUT/stream.err
  timeout 200 python tools/stream.py --gentle --imu-noise > $OUT/stream_gentle.json 2>> $OUT/stream.err
fi
if want loop; then
  timeout 200 python tools/fast_csm_bench.py --reps 9 > $OUT/fast_csm.json 2> $OUT/fast_csm.err
  timeout 200 python tools/fast_csm_bench.py --full --reps 9 > $OUT/fast_csm_full.json 2>> $OUT/fast_csm.err
  timeout 300 python tools/mirror_window_stream.py > $OUT/mirror_window_stream.json 2> $OUT/mirror.err
  (cd tools/ubench && ./graph_latency) > $OUT/graph_latency.txt 2>&1
fi
if want wrefcpp; then timeout 900 python tools/wref_cpp.py > $OUT/wref_cpp.json 2> $OUT/wref_cpp.err; echo "wrefcpp rc=$?"; fi
if want imucost; then timeout 300 python tools/imu_window_cost.py > $OUT/imu_window_cost.json 2> $OUT/imu_window_cost.err; fi
if want pmc5; then
  cd /tmp
  i=0
  for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
    timeout 300 rocprofv3 --pmc $P --kernel-include-regex "rtcsm_score_box" --output-format csv -d $OUT/pmc5_$i -o p -- python $R/tools/c5bench.py --reps 2 --check 0 > $OUT/pmc5_$i.log 2>&1
    echo "pmc5 pass $i rc=$?"
    i=$((i+1))
  done
  cd $R
fi
if want hist; then timeout 300 python tools/hist_bench.py --check > $OUT/hist_bench.json 2> $OUT/hist_bench.err; fi
if want tests; then timeout 900 python -m pytest tests -q -m gpu > $OUT/gputest.log 2>&1; tail -3 $OUT/gputest.log; fi
ls $OUT | head -60
