#!/usr/bin/env python3
"""bench.py -- scans/sec of the scan-to-submap front end on MI355X.

One "step" = one pass of the hot path over one synthetic scan (BASELINE.json config 2, the
north-star workload "W-dense"): a 64-beam x 1024-azimuth cloud (all returns, nothing filtered)
is matched against a 10 cm HybridGrid submap by RealTimeCorrelativeScanMatcher3D, refined by
CeresScanMatcher3D against the high (0.10 m) and low (0.45 m) resolution grids, and inserted into
both grids -- LocalTrajectoryBuilder3D::AddAccumulatedRangeData minus the GTSAM window
(local_trajectory_builder_3d.cc:493-572).  Clouds and grids are resident in HBM when the timed
region starts.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, every rank runs its own independent scan stream against its own
submap (weak scaling, no data-path collective; SURVEY.md 8e "replicas").  `python bench.py --gpus N`
without a launcher starts the N ranks itself (re-executes under torch.distributed.run); the printed
`n_gpus` is the communicator's size.  Rank 0 prints ONE JSON line; for N > 1 it also carries `sharded`
(config 4: the same stream with the search window sharded over the ranks) and `sharded_config5` (the
128 x 2048 @ 5 cm search, 98 % score volume, sharded the same way).  The CPU oracle is used ONLY for the
`cpu_baseline` / `parity` legs (rank 0, N == 1), after the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "d-liom_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from benchlib import (RTCSM_OPTS, CSM_OPTS, SECOND_SUBMAP, build_scene, insertion_targets)  # noqa: E402
from benchlib.config5 import config5_line, config5_sharded_line  # noqa: E402
from benchlib.roofline import roofline_block  # noqa: E402
from benchlib.wref import wref_line  # noqa: E402
from benchlib import multigpu  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--beams", type=int, default=64)
    ap.add_argument("--azimuths", type=int, default=1024)
    ap.add_argument("--high-resolution", type=float, default=0.10)
    ap.add_argument("--low-resolution", type=float, default=0.45)
    ap.add_argument("--map-scans", type=int, default=20, help="scans inserted at ground truth before matching")
    ap.add_argument("--distinct-scans", type=int, default=4)
    ap.add_argument("--shard-candidates", action="store_true",
                    help="config 4: ONE scan stream; the RTCSM search window is sharded over the ranks "
                         "(two 8-byte RCCL max all-reduces per scan), Ceres + insertion replicated")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skips the CPU oracle leg (cpu_baseline AND the post-timing parity check)")
    ap.add_argument("--no-wref", action="store_true", help="skips the W-ref (reference-faithful filter chain) line")
    ap.add_argument("--dump-steps", action="store_true", help="per-step stage times on stderr (debugging)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 5),
                    help="5 = BASELINE config 5 (128 x 2048 returns, 5 cm voxels, 3 map scans); "
                         "2 = the headline config")
    ap.add_argument("--no-pmc", action="store_true", help="skips the rocprofv3 --pmc child runs (roofline counters)")
    ap.add_argument("--no-config5", action="store_true",
                    help="skips the config-5 line (N = 1: two submaps, four grids, three steps; N > 1: the sharded one)")
    ap.add_argument("--no-rccl-check", action="store_true",
                    help="N = 1: skips the one-rank run of the library's RCCL entry point")
    # internal: scene + 3 matches, no timing
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.config == 5:
        a.beams, a.azimuths, a.high_resolution = 128, 2048, 0.05
        a.map_scans = min(a.map_scans, 3)
        a.distinct_scans = min(a.distinct_scans, 2)
        if a.steps == 20 and a.warmup == 3:  # the defaults: a config-5 step takes ~0.4 s
            a.steps, a.warmup = 5, 1
    return a


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: become N ranks (one per GPU) under torch.distributed.run."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)



def pmc_child(args):
    """Body of the rocprofv3 --pmc child runs (tools/pmc_live.py): the bench scene, three RTCSM3D matches."""
    import dliom as dl
    from dliom import synth
    dl.load_library()
    ctx = dl.Context(0)
    ins, g_hi, g_lo, scans = build_scene(args, dl, synth, ctx)
    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, RTCSM_OPTS)
    for i in range(3):
        rt.Match(scans[i % len(scans)]["init"], scans[i % len(scans)]["cloud"], g_hi)
    ctx.synchronize()


def main():
    args = parse()
    if args.pmc_child:
        return pmc_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)  # does not return
    # ONE JSON line on stdout is the contract; RCCL prints a version banner on stdout when its first communicator comes
    # up (the one-rank check at N = 1, torch's and the library's at N > 1): from here on file descriptor 1 IS stderr, and
    # the JSON line goes to the real stdout kept aside
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU)" % (args.gpus, world))
    import torch
    dist = None
    # DLIOM_BENCH_BACKEND=gloo: control-flow check of the N > 1 path on a one-GPU box (every rank on GPU 0, host
    # collectives); the driver's runs use the default, nccl (= RCCL), one rank per GPU
    backend = os.environ.get("DLIOM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libdliom has no CPU fallback)")
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            if torch.cuda.device_count() < world:
                raise SystemExit("bench.py: --gpus %d needs %d GPUs, %d visible (DLIOM_BENCH_BACKEND=gloo runs the "
                                 "N-rank control flow on one GPU)" % (world, world, torch.cuda.device_count()))
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        world = dist.get_world_size()
    torch.cuda.set_device(local_rank)
    coll_device = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")

    import dliom as dl
    from dliom import synth
    dl.load_library()
    ctx = dl.Context(local_rank)

    # ---------------------------------------------------------------- scene (per rank: own time offset)
    # the sharded lines go through the library's own RCCL entry point (dliom_rtcsm3d_match_sharded_rccl, an ncclComm_t of
    # this process) when the backend is nccl -- also on ONE rank (--gpus 1 --shard-candidates: a one-rank communicator,
    # same winner as the unsharded line); DLIOM_BENCH_BACKEND=gloo keeps the callback entry point over host collectives
    rccl_comm = None
    if backend == "nccl" and (world > 1 or args.shard_candidates):
        from dliom import sharded as _sh
        rccl_comm = _sh.RcclCommunicator(rank, world, dist)
        assert rccl_comm.ranks_seen == world, (rccl_comm.ranks_seen, world)
    sharded_mode = args.shard_candidates and (world > 1 or rccl_comm is not None)
    # replicas: every rank runs the SAME scan stream on its own submap copy, so that the per-GPU work
    # (N, C, evaluations) is identical and the max-over-ranks time measures scaling, not scene
    # differences; sharded: all ranks share one stream by construction
    ins, g_hi, g_lo, scans = build_scene(args, dl, synth, ctx)
    if world > 1:
        args.no_pmc = True  # the counters belong to the 1-GPU line

    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, RTCSM_OPTS)
    cs = dl.CeresScanMatcher3D(ctx, CSM_OPTS)
    shard = dl.RtcsmShard(ctx, RTCSM_OPTS, rank, world) if (world > 1 or rccl_comm is not None) else None
    from dliom import sharded
    dev = coll_device

    coll_clock = multigpu.CollectiveClock(dl, ctx, uses_rccl=rccl_comm is not None)

    def sharded_rtcsm(shard_, sc_, grid_):
        if rccl_comm is not None:  # ncclAllReduce(max, uint64, count 1) on the context's stream, inside the library
            return shard_.match_rccl(sc_["init"], sc_["cloud"], grid_, rccl_comm.handle)
        return shard_.match(sc_["init"], sc_["cloud"], grid_,
                            coll_clock.wrap(lambda v: sharded.all_reduce_max_int(v, dist, dev)))
    stage = {"rtcsm": 0.0, "ceres": 0.0, "insert": 0.0}
    evals = []
    use_shards = [sharded_mode]  # flipped for the second (config 4) line of an N > 1 replica run

    def step(i, timed):
        sc = scans[i % len(scans)]
        a = time.perf_counter()
        if use_shards[0]:
            # dliom_rtcsm3d_match_sharded: own rotations, own exact winner, ONE 8-byte max all-reduce (RCCL)
            _, p1 = sharded_rtcsm(shard, sc, g_hi)
        else:
            _, p1 = rt.Match(sc["init"], sc["cloud"], g_hi)
        b = time.perf_counter()
        p2, summ = cs.Match(sc["init"][:3], p1, [(sc["cloud"], g_hi), (sc["cloud"], g_lo)])
        c = time.perf_counter()
        pf = p2.astype(np.float32)
        # Submap3D::InsertRangeData: high-resolution grid (range filtered) + low-resolution grid, fused
        dl.insert_cloud_multi(ins, sc["cloud"], insertion_targets(g_hi, g_lo, pf))
        d = time.perf_counter()
        if timed and args.dump_steps:
            sys.stderr.write("step %d scan %d: rtcsm %.3f ceres %.3f insert %.3f ms, E %d\n" % (
                i, i % len(scans), 1e3 * (b - a), 1e3 * (c - b), 1e3 * (d - c), summ["num_residual_evaluations"]))
        if timed:
            stage["rtcsm"] += b - a
            stage["ceres"] += c - b
            stage["insert"] += d - c
            evals.append(summ["num_residual_evaluations"])
        return p2

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, False)
    # HIP events around the dominant kernel only inside the timed region (roofline.achieved); the
    # per-kernel breakdown comes from a few extra, untimed steps with every kernel timed
    ctx.set_profiling(2)
    ctx.reset_profiling()
    # The harness, not the product: with torch imported a full (generation 2) collection of CPython's cyclic garbage
    # collector takes ~35 ms -- 24 steps of this benchmark -- and when it fires depends on the interpreter's allocation
    # count.  Collect now, keep the collector off for the timed steps.
    import gc
    gc.collect()
    gc.disable()
    fence()
    lat = []
    t_begin = time.perf_counter()
    for i in range(args.steps):
        s0 = time.perf_counter()
        step(args.warmup + i, True)
        lat.append(time.perf_counter() - s0)
    fence()
    elapsed = time.perf_counter() - t_begin
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    gc.enable()
    score_ms, score_n = ctx.kernel_time(dl.KERNEL_RTCSM_SCORE)
    ctx.set_profiling(0)
    # N > 1 replica run: the same scans once more with the search window SHARDED over the ranks (BASELINE config 4).
    # Replicas hold identical submaps, so every rank can take its share of the rotations of ONE stream: strong scaling.
    sharded_line = None
    if world > 1 and not sharded_mode:
        use_shards[0] = True
        for i in range(args.warmup):
            step(args.warmup + args.steps + i, False)
        fence()
        # HIP events around the score kernel and the library's all-reduce (DLIOM_KERNEL_ALLREDUCE)
        ctx.set_profiling(2 | 64)
        ctx.reset_profiling()
        coll_clock.reset()
        t_s = time.perf_counter()
        for i in range(args.steps):
            step(2 * args.warmup + args.steps + i, False)
        fence()
        el = time.perf_counter() - t_s
        shard_score_ms, shard_score_n = ctx.kernel_time(dl.KERNEL_RTCSM_SCORE)
        allreduce_ms = coll_clock.ms_per_match(args.steps)
        ctx.set_profiling(0)
        # the part of a step that does NOT shrink with the number of ranks: the step minus this rank's score kernel,
        # the largest over the ranks
        tt = torch.tensor([el, el - 1e-3 * shard_score_ms, allreduce_ms if allreduce_ms is not None else -1.0],
                          dtype=torch.float64, device=coll_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        # every rank's view of the communicator, every rank's sharded winner against rank 0's UNSHARDED match, the
        # replicas' grids after all of the above (benchlib/multigpu.py); collective, untimed
        mg_checks = multigpu.checks(dl, ctx, dist, world, rank, rt, shard, scans, g_hi, g_lo, sharded_rtcsm,
                                    rccl_comm.ranks_seen if rccl_comm is not None else dist.get_world_size())
        use_shards[0] = False
        sharded_line = {"workload": "config4: ONE scan stream, RTCSM3D search window sharded over %d ranks (one 8-byte RCCL "
                                    "max all-reduce per scan), Ceres + insertion replicated" % world,
                        "collective": ("dliom_rtcsm3d_match_sharded_rccl (ncclAllReduce inside the library)" if rccl_comm is not None
                                       else "dliom_rtcsm3d_match_sharded + torch.distributed callback (%s)" % backend),
                        "ranks_seen": rccl_comm.ranks_seen if rccl_comm is not None else world,
                        "allreduce_ms_per_match_this_rank": allreduce_ms,
                        "allreduce_ms_per_match_max_over_ranks": (float(tt[2].item()) if float(tt[2].item()) >= 0
                                                                  else None),
                        "allreduce_timed_by": ("HIP events inside the library around copy + ncclAllReduce + copy"
                                               if rccl_comm is not None else
                                               "host clock around the torch.distributed callback"),
                        "checks": mg_checks,
                        "value": args.steps / float(tt[0].item()), "unit": "scans/s", "scaling": "strong",
                        "ms_per_step": 1e3 * float(tt[0].item()) / args.steps,
                        "score_kernel_ms_per_step_this_rank": shard_score_ms / max(1, args.steps),
                        "serial_remainder_ms_per_step": 1e3 * float(tt[1].item()) / args.steps,
                        "note": "Amdahl: only the score volume (~60 % of a 1-GPU step) shards; Ceres, insertion, the "
                                "bounds / rescoring kernels and two host synchronisations per scan stay serial: "
                                "config 2 strong-scales <= 1 / (0.4 + 0.6 / N) = 2.1x at N = 8; read the 8-GPU "
                                "number off sharded_config5"}

    # ... and the search where sharding pays: config 5 (128 x 2048 returns, 5 cm voxels, ~3e6 candidates), whose step
    # is 98 % score volume -- every rank builds the same small submap and takes its share of the rotations
    config5_sharded = None
    if world > 1 and not sharded_mode and args.config == 2 and not args.no_config5:
        config5_sharded = config5_sharded_line(args, dl, synth, ctx, rank, world, dist, coll_device, torch, sharded,
                                               rccl_comm)

    extra = max(1, min(5, args.steps))
    ctx.set_profiling(1)
    ctx.reset_profiling()
    for i in range(extra):
        step(args.warmup + args.steps + i, False)
    fence()
    breakdown = {name: ctx.kernel_time(kid)[0] / extra for name, kid in (
        ("rtcsm_score", dl.KERNEL_RTCSM_SCORE), ("rtcsm_select", dl.KERNEL_RTCSM_SELECT),
        ("rtcsm_rescore", dl.KERNEL_RTCSM_RESCORE), ("csm_eval", dl.KERNEL_CSM_EVAL),
        ("insert", dl.KERNEL_INSERT))}
    ctx.set_profiling(0)
    st = rt.last_stats()
    n_pts = int(st.num_points)
    C = int(st.window.num_candidates)

    # N = 1: the library's RCCL entry point on a one-rank communicator (ncclAllReduce really runs), so that every driver
    # line shows it alive: same winner as the unsharded match, and what the collective costs when there is nobody to wait for
    rccl_one_rank = None
    if world == 1 and backend == "nccl" and not args.no_rccl_check and not sharded_mode:
        try:
            comm1 = sharded.RcclCommunicator(0, 1)
            sh1 = dl.RtcsmShard(ctx, RTCSM_OPTS, 0, 1)
            sc = scans[0]
            s_ref, p_ref = rt.Match(sc["init"], sc["cloud"], g_hi)
            s_got, p_got = sh1.match_rccl(sc["init"], sc["cloud"], g_hi, comm1.handle)
            ctx.synchronize()
            t_r = time.perf_counter()
            for _ in range(5):
                sh1.match_rccl(sc["init"], sc["cloud"], g_hi, comm1.handle)
            ctx.synchronize()
            ms_rccl = 1e3 * (time.perf_counter() - t_r) / 5
            t_r = time.perf_counter()
            for _ in range(5):
                rt.Match(sc["init"], sc["cloud"], g_hi)
            ctx.synchronize()
            rccl_one_rank = {"entry_point": "dliom_rtcsm3d_match_sharded_rccl", "ranks_seen": comm1.ranks_seen,
                             "same_winner": bool(np.array_equal(p_ref, p_got) and
                                                 np.float32(s_ref) == np.float32(s_got)),
                             "ms_per_match": ms_rccl, "unsharded_ms_per_match": 1e3 * (time.perf_counter() - t_r) / 5}
            comm1.close()
        except Exception as e:  # RCCL's bootstrap is the environment's, not the product's
            rccl_one_rank = {"error": ("%s: %s" % (type(e).__name__, e))[:300]}

    out = None
    if rank == 0:
        total_scans = args.steps if sharded_mode else args.steps * world
        value = total_scans / elapsed
        # 12 B point + 2 B voxel per candidate-point pair (a shard scores C/world of them)
        alg_bytes = 14.0 * C * n_pts / (world if sharded_mode else 1)
        k_ms = score_ms / max(score_n, 1)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        pairs = float(C) * n_pts / (world if sharded_mode else 1)
        out = {
            "metric": "scans/sec (%d-beam x %d pts -> %g cm 3D submap)" % (args.beams, args.azimuths,
                                                                           100 * args.high_resolution),
            "value": value,
            "unit": "scans/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "p50_latency_ms": 1e3 * float(np.median(lat)),
            "higher_is_better": True,
            "scaling": "strong" if sharded_mode else "weak",
            "vs_baseline": None,
            "dtype": "f32 transforms + u16 voxels/u64 sums (rtcsm), f64 (ceres)",
            "data": "synthetic",
            "config": {
                "workload": "%s W-dense: %dx%d scan, all returns matched (adaptive voxel filters "
                            "neutralised), RTCSM3D + CeresScanMatcher3D(hi+lo) + insertion(hi+lo)" %
                            ({(64, 1024, 0.1): "config2", (128, 2048, 0.05): "config5"}.get(
                                (args.beams, args.azimuths, args.high_resolution), "custom"),
                             args.beams, args.azimuths),
                "high_resolution": args.high_resolution, "low_resolution": args.low_resolution,
                "N_hi": n_pts, "N_lo": n_pts, "C": C,
                "linear_window": int(st.window.linear_window_size),
                "angular_window": int(st.window.angular_window_size),
                "max_scan_range": float(st.window.max_scan_range),
                "E_mean": float(np.mean(evals)) if evals else 0.0,
                "rescored_candidates_last": int(st.num_rescored), "box_kernel_variant": int(st.box_kernel_variant),
                "map_scans": args.map_scans, "grids_inserted_into": 2 + len(SECOND_SUBMAP),
                "parallelism": ("candidate shards x%d (RCCL max all-reduce)" if sharded_mode
                                else "replicas x%d") % world,
            },
            "stage_ms_per_scan": {k: 1e3 * v / args.steps for k, v in stage.items()},
            # the part of a step that does not shard (config 4's Amdahl ceiling): the step minus this rank's score kernel
            "serial_remainder_ms_per_step": 1e3 * elapsed / args.steps - k_ms,
            "kernel_ms_per_scan": breakdown,
            "roofline": roofline_block(args, pairs, k_ms, int(score_n), alg_bytes, int(st.score_kernel)),
        }
        if rccl_one_rank is not None:
            out["sharded_rccl_one_rank"] = rccl_one_rank
        if sharded_line is not None:
            out["sharded"] = sharded_line
        if config5_sharded is not None:
            out["sharded_config5"] = config5_sharded
        if world == 1 and not args.no_wref and args.config == 2:
            out["wref"] = wref_line(dl, ctx, cpu=not args.no_cpu_baseline)
        if world == 1 and args.config == 2 and not args.no_config5:
            try:  # BASELINE config 5 in the driver's record: two active submaps, four grids, C = 2 352 637, three steps
                out["config5"] = config5_line(dl, synth, ctx, steps=3, with_oracle=not args.no_cpu_baseline)
            except Exception as e:
                out["config5"] = {"error": ("%s: %s" % (type(e).__name__, e))[:300]}
        if world == 1 and not args.no_cpu_baseline:
            from benchlib.cpu_legs import cpu_baseline, parity_check  # the only importer of oracle/
            # the oracle leg: checker first (one more step, compared end to end), then the CPU baseline
            parity = parity_check(dl, ctx, scans[1 % len(scans)], g_hi, g_lo, ins, rt, cs)
            out["parity_checked"] = bool(parity["ok"])
            out["parity"] = parity
            out["cpu_baseline"] = cpu_baseline(args, dl, scans[0], g_hi, g_lo, ins, C, n_pts)
    if out is not None:
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    for sc in scans:
        sc["cloud"].close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

