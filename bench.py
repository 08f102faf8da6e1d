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

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec); 6.29 TB/s measured copy
# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, v_fma_f32 (wave64) = 2 cycles, 2.4 GHz
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9

RTCSM_OPTS = dict(linear_search_window=0.15, angular_search_window=float(np.deg2rad(1.0)),
                  translation_delta_cost_weight=1e-1, rotation_delta_cost_weight=1e-1)
CSM_OPTS = dict(occupied_space_weight=[1.0, 6.0], translation_weight=5.0, rotation_weight=4e2,
                only_optimize_yaw=False, use_nonmonotonic_steps=False, max_num_iterations=12)
HIT_P, MISS_P, FREE = 0.55, 0.49, 2
HIGH_RES_MAX_RANGE = 20.0


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
                    help="5 = BASELINE config 5 (128 x 2048 returns, 5 cm voxels, 3 map scans); 2 = the headline config")
    ap.add_argument("--no-pmc", action="store_true", help="skips the rocprofv3 --pmc child runs (roofline counters)")
    ap.add_argument("--no-config5", action="store_true", help="skips the config-5 line (N = 1: two submaps, four grids, three steps; N > 1: the sharded one)")
    ap.add_argument("--no-rccl-check", action="store_true", help="N = 1: skips the one-rank run of the library's RCCL entry point")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # internal: scene + 3 matches, no timing
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


SECOND_SUBMAP = []  # --config 5: the (hi, lo) grids of the second active submap, inserted into beside the matched one


def insertion_targets(g_hi, g_lo, pf):
    t = [(g_hi, [pf], HIGH_RES_MAX_RANGE), (g_lo, [pf], 0.0)]
    if SECOND_SUBMAP:
        t += [(SECOND_SUBMAP[0], [pf], HIGH_RES_MAX_RANGE), (SECOND_SUBMAP[1], [pf], 0.0)]
    return t


def build_scene(args, dl, synth, ctx):
    """Submap (map_scans scans inserted at ground truth) and the scans to match; the same on every rank."""
    ins = dl.RangeDataInserter3D(HIT_P, MISS_P, FREE, ctx=ctx)
    g_hi = dl.HybridGrid(ctx, args.high_resolution)
    g_lo = dl.HybridGrid(ctx, args.low_resolution)
    second = []
    if args.config == 5:
        # BASELINE config 5, "multi-submap insertion": TWO active submaps (submap_3d.cc:303-314) -- the newer one holds the
        # later half of the map scans -- and beams of +-35 degrees, so that returns reach the cube's corners (26 m) and the
        # search window is the one BASELINE.md section 3 states: C = 343 x 19^3 = 2 352 637
        synth.ELEVATION["cube"] = (-35.0, 35.0)
        second = [dl.HybridGrid(ctx, args.high_resolution), dl.HybridGrid(ctx, args.low_resolution)]
    SECOND_SUBMAP[:] = second
    centers = synth.bubbles()
    for s in range(args.map_scans):
        pose = synth.trajectory_pose(0.1 * s)
        pts, _ = synth.scan(pose, args.beams, args.azimuths, centers=centers)
        cloud = dl.PointCloud(ctx, pts)
        pf = pose.astype(np.float32)
        ins.InsertCloud(g_hi, cloud, poses=[pf], max_range=HIGH_RES_MAX_RANGE)
        ins.InsertCloud(g_lo, cloud, poses=[pf])
        if second and s >= args.map_scans // 2:
            ins.InsertCloud(second[0], cloud, poses=[pf], max_range=HIGH_RES_MAX_RANGE)
            ins.InsertCloud(second[1], cloud, poses=[pf])
        cloud.close()
    scans = []
    for k in range(args.distinct_scans):
        truth = synth.trajectory_pose(0.1 * (args.map_scans + k))
        pts, _ = synth.scan(truth, args.beams, args.azimuths, centers=centers)
        init = synth.perturb_pose(truth, 0.1, 0.5, seed=13 + k)
        scans.append(dict(truth=truth, pts=pts, init=init, cloud=dl.PointCloud(ctx, pts)))
    return ins, g_hi, g_lo, scans


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

    def sharded_rtcsm(shard_, sc_, grid_):
        if rccl_comm is not None:  # ncclAllReduce(max, uint64, count 1) on the context's stream, inside the library
            return shard_.match_rccl(sc_["init"], sc_["cloud"], grid_, rccl_comm.handle)
        return sharded.sharded_match(shard_, sc_["init"], sc_["cloud"], grid_, dist=dist, device=dev)
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
        ctx.set_profiling(2)  # HIP events around the score kernel only, as in the replica run above
        ctx.reset_profiling()
        t_s = time.perf_counter()
        for i in range(args.steps):
            step(2 * args.warmup + args.steps + i, False)
        fence()
        el = time.perf_counter() - t_s
        shard_score_ms, shard_score_n = ctx.kernel_time(dl.KERNEL_RTCSM_SCORE)
        ctx.set_profiling(0)
        # the part of a step that does NOT shrink with the number of ranks: the step minus this rank's score kernel,
        # the largest over the ranks
        tt = torch.tensor([el, el - 1e-3 * shard_score_ms], dtype=torch.float64, device=coll_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        use_shards[0] = False
        sharded_line = {"workload": "config4: ONE scan stream, RTCSM3D search window sharded over %d ranks (one 8-byte RCCL "
                                    "max all-reduce per scan), Ceres + insertion replicated" % world,
                        "collective": ("dliom_rtcsm3d_match_sharded_rccl (ncclAllReduce inside the library)" if rccl_comm is not None
                                       else "dliom_rtcsm3d_match_sharded + torch.distributed callback (%s)" % backend),
                        "ranks_seen": rccl_comm.ranks_seen if rccl_comm is not None else world,
                        "value": args.steps / float(tt[0].item()), "unit": "scans/s", "scaling": "strong",
                        "ms_per_step": 1e3 * float(tt[0].item()) / args.steps,
                        "score_kernel_ms_per_step_this_rank": shard_score_ms / max(1, args.steps),
                        "serial_remainder_ms_per_step": 1e3 * float(tt[1].item()) / args.steps,
                        "note": "Amdahl: only the score volume (~60 % of a 1-GPU step) shards; Ceres, insertion, the "
                                "bounds / rescoring kernels and two host synchronisations per scan stay serial"}

    # ... and the search where sharding pays: config 5 (128 x 2048 returns, 5 cm voxels, ~3e6 candidates), whose step
    # is 98 % score volume -- every rank builds the same small submap and takes its share of the rotations
    config5_sharded = None
    if world > 1 and not sharded_mode and args.config == 2 and not args.no_config5:
        config5_sharded = config5_sharded_line(args, dl, synth, ctx, rank, world, dist, coll_device, torch, sharded, rccl_comm)

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
                             "same_winner": bool(np.array_equal(p_ref, p_got) and np.float32(s_ref) == np.float32(s_got)),
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
            "metric": "scans/sec (%d-beam x %d pts -> %g cm 3D submap)" % (args.beams, args.azimuths, 100 * args.high_resolution),
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
                "rescored_candidates_last": int(st.num_rescored),
                "map_scans": args.map_scans, "grids_inserted_into": 2 + len(SECOND_SUBMAP),
                "parallelism": ("candidate shards x%d (RCCL max all-reduce)" if sharded_mode else "replicas x%d") % world,
            },
            "stage_ms_per_scan": {k: 1e3 * v / args.steps for k, v in stage.items()},
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


def config5_sharded_line(args, dl, synth, ctx, rank, world, dist, dev, torch, sharded, rccl_comm=None):
    """BASELINE config 5 with the search window sharded over the ranks (config 4's protocol): one 128 x 2048 scan
    stream, every rank scores its own rotations of the ~3e6-candidate window, one 8-byte max all-reduce per scan,
    Ceres + insertion replicated."""
    import copy
    a5 = copy.copy(args)
    a5.beams, a5.azimuths, a5.high_resolution, a5.map_scans, a5.distinct_scans = 128, 2048, 0.05, 3, 1
    ins, g_hi, g_lo, scans = build_scene(a5, dl, synth, ctx)
    shard = dl.RtcsmShard(ctx, RTCSM_OPTS, rank, world)
    cs = dl.CeresScanMatcher3D(ctx, CSM_OPTS)
    sc = scans[0]

    def one():
        if rccl_comm is not None:
            _, p1 = shard.match_rccl(sc["init"], sc["cloud"], g_hi, rccl_comm.handle)
        else:
            _, p1 = sharded.sharded_match(shard, sc["init"], sc["cloud"], g_hi, dist=dist, device=dev)
        p2, _ = cs.Match(sc["init"][:3], p1, [(sc["cloud"], g_hi), (sc["cloud"], g_lo)])
        pf = p2.astype(np.float32)
        dl.insert_cloud_multi(ins, sc["cloud"], insertion_targets(g_hi, g_lo, pf))

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    steps = 3
    one()
    fence()
    ctx.set_profiling(2)
    ctx.reset_profiling()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    fence()
    el = time.perf_counter() - t0
    score_ms, _ = ctx.kernel_time(dl.KERNEL_RTCSM_SCORE)
    ctx.set_profiling(0)
    tt = torch.tensor([el, el - 1e-3 * score_ms], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    st = dl.RealTimeCorrelativeScanMatcher3D(ctx, RTCSM_OPTS).last_stats()
    sc["cloud"].close()
    g_hi.close()
    g_lo.close()
    return {"workload": "config5 W-dense: ONE 128x2048 scan stream @ 5 cm, RTCSM3D window sharded over %d ranks (one 8-byte "
                        "RCCL max all-reduce per scan), Ceres + insertion replicated" % world,
            "collective": "dliom_rtcsm3d_match_sharded_rccl" if rccl_comm is not None else "callback",
            "value": steps / float(tt[0].item()), "unit": "scans/s", "scaling": "strong", "steps": steps,
            "ms_per_step": 1e3 * float(tt[0].item()) / steps, "score_kernel_ms_per_step_this_rank": score_ms / steps,
            "serial_remainder_ms_per_step": 1e3 * float(tt[1].item()) / steps,
            "C": int(st.window.num_candidates), "N": int(st.num_points)}


def config5_line(dl, synth, ctx, steps=3, with_oracle=True):
    """BASELINE config 5 as BASELINE.json words it -- "128-beam x 2048 dense cloud, 5 cm voxels, multi-submap insertion +
    scan match" -- short enough for the default N = 1 line, so that the DRIVER times it: two active submaps (four grids:
    hi + lo of each, submap_3d.cc:303-314), the scan matched against the older submap's 5 cm grid (RTCSM3D over
    C = 343 x 19^3 candidates + CeresScanMatcher3D hi + lo) and inserted into all four grids by the fused insertion.
    The sensor's beams span +-35 degrees here so that returns reach the 30 m cube's corners (26 m): the window
    BASELINE.md section 3 states, C = 2 352 637 (with the +-15 degrees of config 2 the farthest return is 23.5 m away and
    C = 1 685 159).  Winner checked against the oracle on sampled candidates (the full loop is 6e11 lookups)."""
    keep = synth.ELEVATION["cube"]
    synth.ELEVATION["cube"] = (-35.0, 35.0)
    try:
        beams, az, res_hi, res_lo, map_scans = 128, 2048, 0.05, 0.45, 3
        ins = dl.RangeDataInserter3D(HIT_P, MISS_P, FREE, ctx=ctx)
        grids = [dl.HybridGrid(ctx, r) for r in (res_hi, res_lo, res_hi, res_lo)]  # submap A (hi, lo), submap B (hi, lo)
        centers = synth.bubbles()
        for s in range(map_scans):
            pose = synth.trajectory_pose(0.1 * s)
            pts, _ = synth.scan(pose, beams, az, centers=centers)
            cloud = dl.PointCloud(ctx, pts)
            pf = pose.astype(np.float32)
            targets = [(grids[0], [pf], HIGH_RES_MAX_RANGE), (grids[1], [pf], 0.0)]
            if s >= map_scans // 2:  # the newer submap holds the later half of the scans (ActiveSubmaps3D)
                targets += [(grids[2], [pf], HIGH_RES_MAX_RANGE), (grids[3], [pf], 0.0)]
            dl.insert_cloud_multi(ins, cloud, targets)
            cloud.close()
        truth = synth.trajectory_pose(0.1 * map_scans)
        pts, _ = synth.scan(truth, beams, az, centers=centers)
        sc = dict(truth=truth, pts=pts, init=synth.perturb_pose(truth, 0.1, 0.5, seed=13), cloud=dl.PointCloud(ctx, pts))
    finally:
        synth.ELEVATION["cube"] = keep
    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, RTCSM_OPTS)
    cs = dl.CeresScanMatcher3D(ctx, CSM_OPTS)
    stage = {"rtcsm": 0.0, "ceres": 0.0, "insert": 0.0}

    def one(timed):
        a = time.perf_counter()
        _, p1 = rt.Match(sc["init"], sc["cloud"], grids[0])
        b = time.perf_counter()
        p2, _ = cs.Match(sc["init"][:3], p1, [(sc["cloud"], grids[0]), (sc["cloud"], grids[1])])
        c = time.perf_counter()
        pf = p2.astype(np.float32)
        dl.insert_cloud_multi(ins, sc["cloud"], [(grids[0], [pf], HIGH_RES_MAX_RANGE), (grids[1], [pf], 0.0),
                                                 (grids[2], [pf], HIGH_RES_MAX_RANGE), (grids[3], [pf], 0.0)])
        ctx.synchronize()
        d = time.perf_counter()
        if timed:
            stage["rtcsm"] += b - a
            stage["ceres"] += c - b
            stage["insert"] += d - c

    one(False)
    ctx.set_profiling(2)
    ctx.reset_profiling()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one(True)
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    k_ms, k_n = ctx.kernel_time(dl.KERNEL_RTCSM_SCORE)
    ctx.set_profiling(0)
    st = rt.last_stats()
    C, n = int(st.window.num_candidates), int(st.num_points)
    rebuilds, mirror_bytes, windowed = grids[0].mirror_stats()
    out = {"workload": "config5 W-dense: 128x2048 scan (beams +-35 deg: returns to the cube's corners), 5 cm voxels, RTCSM3D + "
                       "CeresScanMatcher3D(hi+lo) against the older of TWO active submaps, fused insertion into all four grids",
           "value": steps / elapsed, "unit": "scans/s", "steps": steps, "ms_per_step": 1e3 * elapsed / steps,
           "stage_ms_per_scan": {k: 1e3 * v / steps for k, v in stage.items()},
           "C": C, "N": n, "angular_window": int(st.window.angular_window_size), "linear_window": int(st.window.linear_window_size),
           "max_scan_range": float(st.window.max_scan_range), "grids_inserted_into": 4, "hi_grid_bits": int(grids[0].bits),
           "score_kernel": int(st.score_kernel), "score_kernel_ms": k_ms / max(k_n, 1),
           "pairs_per_s": float(C) * n / (k_ms / max(k_n, 1) * 1e-3) if k_ms > 0 else None,
           "frac_useful": (float(C) * n / (k_ms / max(k_n, 1) * 1e-3)) / USEFUL_PAIRS_PER_S if k_ms > 0 else None,
           "mirror": {"bytes": mirror_bytes, "windowed": windowed, "rebuilds": rebuilds}, "box_kernel_flags": int(rt.box_error())}
    if with_oracle:
        from oracle import oracle as orc
        origins, values = grids[0].download_blocks()
        og = orc.HybridGrid(res_hi)
        leaf, cell = np.nonzero(values)
        xyz = np.stack([origins[leaf, 0] + (cell & 7), origins[leaf, 1] + ((cell >> 3) & 7), origins[leaf, 2] + (cell >> 6)], axis=1).astype(np.int32)
        og.set_values(xyz, values[leaf, cell])
        score, p1 = rt.Match(sc["init"], sc["cloud"], grids[0])
        st = rt.last_stats()
        threads = min(32, os.cpu_count() or 1)
        ref, sampled = sampled_oracle_match(orc, rt, sc, grids[0], og, st, threads, sample=1000, top_n=256)
        out["parity"] = {"ok": bool(sampled["ok"] and int(st.best_index) == ref["best_index"] and
                                    np.float32(score).tobytes() == np.float32(ref["score"]).tobytes() and np.array_equal(p1, ref["pose"])),
                         "how": sampled["how"], "oracle_threads": threads}
    sc["cloud"].close()
    for g in grids:
        g.close()
    return out


def score_kernel_counters(args, kernel):
    """Hardware counters of the score kernel measured NOW: child runs of this script (--pmc-child: the same scene,
    three matches) under `rocprofv3 --pmc`, one per counter group (tools/pmc_live.py).  Nothing is read from a
    committed file; what cannot be measured is absent and printed as null."""
    if args.no_pmc:
        return {}, ["skipped (--no-pmc or N > 1)"]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_live
    child = [os.path.abspath(__file__), "--pmc-child", "--config", str(args.config), "--beams", str(args.beams),
             "--azimuths", str(args.azimuths), "--high-resolution", str(args.high_resolution),
             "--low-resolution", str(args.low_resolution), "--map-scans", str(args.map_scans),
             "--distinct-scans", str(args.distinct_scans)]
    return pmc_live.measure(child, kernel, timeout=240 if args.config == 5 else 150)


# cheapest known instruction sequence per lookup on gfx950: 1.5 v_pk_add_f32 + 3 v_mad_u32_u16 +
# 0.5 v_add3_u32 = 5 VALU per wave-lookup at one issue per 4 cycles and SIMD -> 256 x 4 x 2.4e9 / (5 x 4) x 64 lanes
USEFUL_PAIRS_PER_S = 256 * 4 * 2.4e9 / (5.0 * 4.0) * 64.0


def roofline_block(args, pairs, k_ms, launches, alg_bytes, score_kernel):
    """What bounds the dominant kernel: the vector ALU's instruction issue -- not HBM (the
    kernel moves ~1 % of its algorithmic bytes) and not MFMA (no GEMM in it).  achieved = VALU lane-operations
    per second (SQ_INSTS_VALU x 64 / launch time, both measured in this run); peak = 256 CU x 4 SIMD-32 x 2.4 GHz."""
    t = k_ms * 1e-3
    kernel = {3: "rtcsm_score_box_kernel", 2: "rtcsm_score_dense_kernel", 1: "rtcsm_score_rot_kernel",
              0: "rtcsm_score_kernel"}.get(score_kernel, "?")  # the kernel that ran (dliom_rtcsm_stats.score_kernel)
    counters, problems = score_kernel_counters(args, kernel)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_live
    d = pmc_live.derive(counters, pairs, t) if counters else {}
    valu_per_pair = d.get("valu_instructions_per_pair")
    traffic = d.get("traffic_bytes")
    achieved = (valu_per_pair * pairs / t) if (valu_per_pair and t > 0) else None
    return {
        "kernel": kernel,
        "bound": "valu",
        "achieved": achieved / 1e12 if achieved else None,
        "peak": VALU_PEAK_LANE_OPS / 1e12,
        "unit": "Tlane-op/s",
        "frac": achieved / VALU_PEAK_LANE_OPS if achieved else None,
        # the same launch time against the cheapest instruction sequence known for a lookup: how much of the
        # kernel's issue slots do useful lookups (the headroom; `frac` counts every instruction the kernel issues)
        "frac_useful": (pairs / t) / USEFUL_PAIRS_PER_S if t > 0 else None,
        # the hardware's own counter of the bounding pipe (SQ_ACTIVE_INST_VALU x 4 cycles over the launch's SIMD-cycles):
        # `frac` prices every instruction at the 2-cycle rate of plain fp32 adds, this kernel's are mostly 4-cycle ones
        # (v_mad_u32_u16, v_pk_add_f32, v_add3_u32), so frac <= 0.5 x valu_busy_frac-ish by construction
        "valu_busy_frac": d.get("valu_busy_frac"),
        "valu_instructions_per_pair": valu_per_pair,
        "pairs_per_s": pairs / t if t > 0 else 0.0,
        "avg_launch_ms": k_ms,
        "launches": launches,
        "traffic": traffic,
        "counters_source": "rocprofv3 --pmc child runs of this bench.py invocation (tools/pmc_live.py), %d passes; "
                           "FETCH_SIZE x 2 [gfx950 correction] + WRITE_SIZE" % len(pmc_live.PASSES) if counters else None,
        "counters": {k: v["mean"] for k, v in counters.items()} or None,
        "derived": d or None,
        "counter_problems": problems or None,
        "hbm_side_note": {
            "algorithmic_bytes_per_launch": alg_bytes,  # 14 B per (candidate, point) pair
            "algorithmic_rate_GBs": alg_bytes / t / 1e9 if t > 0 else 0.0,
            "measured_hbm_GBs": (traffic / t / 1e9) if (traffic and t > 0) else None,
            "measured_frac_of_hbm_peak": (traffic / t / 1e9 / HBM_PEAK_GBS) if (traffic and t > 0) else None,
            "note": "points are reused from registers across 27 translations and the mirror sub-boxes from LDS: the "
                    "algorithmic rate is not an HBM rate and is not the roofline",
        },
    }


def wref_line(dl, ctx, cpu):
    """What the reference does with a 64 x 1024 scan, complete (tools/wref_full.py): AddImuData, AddRangeData (voxel
    filters + de-skew), adaptive filters + [RTCSM3D] + Ceres, WindowOptimize, insertion, ComputeHistogram -- with
    trajectory_builder_3d.lua's options and with dlio/config/basic_config_3d.lua's (what D-LIOM ships: RTCSM3D off, 0.3 /
    0.2 / 60 m, gravity factor on); each with the same stream on the CPU oracle and the pose difference between the legs."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import wref_full
    out = {name: wref_full.line(dl, ctx, name, scans=24, warmup=4, cpu_scans=20, cpu=cpu)
           for name in ("trajectory_builder_3d", "basic_config_3d")}
    # round 4: the same chains on a world with a floor (dliom.synth's yard: ragged scans, a 15 000-return floor slice for
    # ComputeHistogram, returns to 80 m) -- the cube has neither floor nor far returns inside the beams
    for name in ("trajectory_builder_3d", "basic_config_3d"):
        out[name + "_yard"] = wref_full.line(dl, ctx, name, scans=24, warmup=4, cpu_scans=20, cpu=cpu, scene="ground")
    return out


def parity_check(dl, ctx, sc, g_hi, g_lo, ins, rt, cs):
    """One more step, compared end to end with the CPU oracle on the grids as they are after the timed region:
    RTCSM3D winner (index, score bits, pose) against the reference's full candidate loop, CeresScanMatcher3D
    pose, both grids after the insertion."""
    from oracle import oracle as orc
    threads = min(8, os.cpu_count() or 1)

    def to_oracle(dg):
        og = orc.HybridGrid(dg.resolution)
        origins, values = dg.download_blocks()
        leaf, cell = np.nonzero(values)
        if len(leaf):
            xyz = np.stack([origins[leaf, 0] + (cell & 7), origins[leaf, 1] + ((cell >> 3) & 7),
                            origins[leaf, 2] + (cell >> 6)], axis=1).astype(np.int32)
            og.set_values(xyz, values[leaf, cell])
        return og

    def cells(keys_from):
        xyz, v = keys_from
        xyz = np.asarray(xyz, dtype=np.int64)
        key = ((xyz[:, 0] + (1 << 20)) << 42) | ((xyz[:, 1] + (1 << 20)) << 21) | (xyz[:, 2] + (1 << 20))
        order = np.argsort(key)
        return key[order], np.asarray(v)[order]

    def device_cells(dg):
        origins, values = dg.download_blocks()
        leaf, cell = np.nonzero(values)
        xyz = np.stack([origins[leaf, 0] + (cell & 7), origins[leaf, 1] + ((cell >> 3) & 7), origins[leaf, 2] + (cell >> 6)], axis=1)
        return cells((xyz, values[leaf, cell]))

    t0 = time.perf_counter()
    og_hi, og_lo = to_oracle(g_hi), to_oracle(g_lo)
    score, p1 = rt.Match(sc["init"], sc["cloud"], g_hi)
    st = rt.last_stats()
    sampled = None
    if float(st.window.num_candidates) * float(st.num_points) <= 2e10:
        ref = orc.rtcsm3d_match_parallel(RTCSM_OPTS, sc["init"], sc["pts"], og_hi, threads=threads)
    else:
        ref, sampled = sampled_oracle_match(orc, rt, sc, g_hi, og_hi, st, threads)
    rtcsm_ok = (int(st.best_index) == ref["best_index"] and np.float32(score).tobytes() == np.float32(ref["score"]).tobytes()
                and np.array_equal(p1, ref["pose"]) and (sampled is None or sampled["ok"]))
    p2, summ = cs.Match(sc["init"][:3], p1, [(sc["cloud"], g_hi), (sc["cloud"], g_lo)])
    r2 = orc.csm3d_match(CSM_OPTS, sc["init"][:3], ref["pose"], [(sc["pts"], og_hi), (sc["pts"], og_lo)])
    dt = float(np.linalg.norm(np.asarray(p2[:3]) - np.asarray(r2["pose"][:3])))
    dq = float(2.0 * np.arccos(min(1.0, abs(float(np.dot(p2[3:], r2["pose"][3:]))))))
    ceres_ok = dt <= 1e-6 and dq <= 1e-6
    pf = np.asarray(p2, dtype=np.float32)
    dl.insert_cloud_multi(ins, sc["cloud"], insertion_targets(g_hi, g_lo, pf))
    world_pts = orc.transform_points(pf, sc["pts"])
    origin = orc.transform_points(pf, np.zeros((1, 3), np.float32))[0]
    d = (world_pts - origin).astype(np.float32)
    nrm = np.sqrt(d[:, 0] * d[:, 0] + (d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]), dtype=np.float32)
    og_hi.insert_tables(origin, world_pts[nrm <= np.float32(HIGH_RES_MAX_RANGE)], ins.hit_table, ins.miss_table, FREE)
    og_lo.insert_tables(origin, world_pts, ins.hit_table, ins.miss_table, FREE)
    grids_ok = True
    for dg, og in ((g_hi, og_hi), (g_lo, og_lo)):
        dk, dv = device_cells(dg)
        ok_, ov = cells(og.export_cells())
        grids_ok = grids_ok and np.array_equal(dk, ok_) and np.array_equal(dv, ov)
    return {"ok": bool(rtcsm_ok and ceres_ok and grids_ok), "rtcsm_winner_bit_equal": bool(rtcsm_ok),
            "rtcsm_best_index": int(st.best_index), "candidates": int(ref["num_candidates"]),
            "ceres_translation_error_m": dt, "ceres_rotation_error_rad": dq, "ceres_tolerance": 1e-6,
            "grids_bit_equal_after_insertion": bool(grids_ok), "box_kernel_flags": int(rt.box_error()),
            "oracle_threads": threads, "seconds": time.perf_counter() - t0,
            "rtcsm_oracle": "full candidate loop" if sampled is None else sampled["how"]}


def sampled_oracle_match(orc, rt, sc, g_hi, og_hi, st, threads, sample=4000, top_n=512):
    """Config 5: the oracle's full candidate loop (4.4e11 lookups) takes hours, so the device's integer score volume is
    checked on `sample` random candidates, and the winner is the first maximum (generation order, strict >) of the
    oracle's exact ScoreCandidate over the `top_n` candidates that rank highest by the real-valued score of that volume
    (tests/test_gpu_full_size.py::test_config5_benchmarked_window_sampled)."""
    C, n = int(st.window.num_candidates), int(st.num_points)
    sums = rt.score_volume(sc["init"], sc["pts"], g_hi)
    idx = np.random.RandomState(11).randint(0, C, size=sample)
    want, _ = orc.rtcsm3d_at(RTCSM_OPTS, sc["init"], sc["pts"], og_hi, idx, threads=threads)
    volume_ok = len(sums) == C and np.array_equal(sums[idx].astype(np.uint64), want)
    tr, ca = orc.rtcsm3d_candidates(RTCSM_OPTS, g_hi.resolution, sc["pts"], sc["init"])
    t_norm = np.linalg.norm(tr[:, :3].astype(np.float64), axis=1)
    angle = 2.0 * np.arctan2(np.linalg.norm(tr[:, 4:7].astype(np.float64), axis=1), np.abs(tr[:, 3].astype(np.float64)))
    arg = t_norm * RTCSM_OPTS["translation_delta_cost_weight"] + angle * RTCSM_OPTS["rotation_delta_cost_weight"]
    k_scale = (0.9 - 0.1) / 32766.0
    real = (sums.astype(np.float64) * k_scale + (0.1 - k_scale) * n) / n * np.exp(-arg * arg)
    order = np.argsort(-real, kind="stable")
    top = np.sort(order[:top_n])
    _, exact = orc.rtcsm3d_at(RTCSM_OPTS, sc["init"], sc["pts"], og_hi, top, threads=threads)
    best = int(top[int(np.argmax(exact))])
    cut_ok = bool(real[order[top_n - 1]] < real[order[0]] * (1.0 - 1e-4))
    ref = {"best_index": best, "score": float(exact.max()), "pose": ca[best].astype(np.float64), "num_candidates": C}
    return ref, {"ok": bool(volume_ok and cut_ok),
                 "how": "%d random candidates' integer sums + exact ScoreCandidate of the %d best-ranked candidates "
                        "(of %d; the full loop is %.1e lookups)" % (sample, top_n, C, float(C) * n)}


def host_cpu_quota():
    """CPUs the cgroup grants this container (cpu.max: quota / period), the scheduler affinity's size, or None."""
    out = {}
    try:
        q, p_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        out["cgroup_cpu_max"] = None if q == "max" else float(q) / float(p_)
    except Exception:
        pass
    try:
        out["sched_affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    return out or None


def cpu_baseline(args, dl, sc, g_hi, g_lo, ins, C, n_pts):
    """Times the CPU oracle (reference-layout pointer-tree HybridGrid, per-candidate
    TransformPointCloud allocation, Jet autodiff + dense QR) on the same scan and the same grids.
    `value` is the FULL RTCSM3D candidate loop on one thread (how the reference runs this path; nothing sampled),
    Ceres and insertion timed in full; beside it the same full loop on 8 threads and on every core of the box, for the
    reference's layout and for the fair-CPU variant."""
    from oracle import oracle as orc

    def to_oracle(dg):
        og = orc.HybridGrid(dg.resolution)
        origins, values = dg.download_blocks()
        for o, v in zip(origins, values):
            nz = np.nonzero(v)[0]
            if len(nz) == 0:
                continue
            xyz = np.stack([o[0] + (nz & 7), o[1] + ((nz >> 3) & 7), o[2] + (nz >> 6)], axis=1)
            og.set_values(xyz, v[nz])
        return og

    og_hi, og_lo = to_oracle(g_hi), to_oracle(g_lo)
    pts, init = sc["pts"], sc["init"]
    from concurrent.futures import ThreadPoolExecutor
    cores = os.cpu_count() or 1
    quota = (host_cpu_quota() or {}).get("cgroup_cpu_max")
    # "all cores" = what this container may use: os.cpu_count() reports the host's 256, the cgroup grants 16 on the GPU
    # boxes of this pool -- more threads than that only add switching
    usable = max(1, min(cores, int(np.ceil(quota)))) if quota else cores
    flat = orc.FlatGridIndex(og_hi)  # built once, outside the timing, as a CPU implementation would keep it beside the tree

    def ref_range(first, cnt):
        return orc.rtcsm3d_match_range(RTCSM_OPTS, init, pts, og_hi, first, cnt)

    def fair_range(first, cnt):
        return orc.rtcsm3d_match_range_fair(RTCSM_OPTS, init, pts, flat, first, cnt)

    # config 2: the WHOLE loop (2.4e9 lookups, ~30 s on one thread).  Config 5's loop is 6e11 lookups -- hours -- so there,
    # and only there, an evenly spread subset of the candidates is timed and scaled (labelled as such)
    budget_lookups = 3.0e9
    sampled = float(C) * n_pts > budget_lookups
    M = C if not sampled else max(256, int(budget_lookups / n_pts))
    scale_up = float(C) / M

    def loop_parts(threads):
        if not sampled:
            return orc._ranges(C, max(1, threads) * 4)
        chunks = 512  # the same subset for every thread count (>= two ranges per thread on a 256-core host)
        per = max(1, M // chunks)
        return [((C // chunks) * k, min(per, C - (C // chunks) * k)) for k in range(chunks)]

    def full_loop(fn, threads):
        """The WHOLE candidate loop (all C candidates, nothing sampled; config 5: see above) cut into contiguous ranges over
        `threads` host threads (ctypes releases the GIL), combined in generation order with the reference's strict `>`."""
        parts = loop_parts(threads)
        if threads <= 1:
            t0 = time.perf_counter()
            res = [fn(f, c) for f, c in parts]
            wall = time.perf_counter() - t0
        else:
            with ThreadPoolExecutor(threads) as pool:
                # untimed: the pool's threads exist and each has its malloc arena (the reference layout allocates per candidate;
                # the first parallel pass over fresh threads measured 4x slower than the second on an 8-core host)
                list(pool.map(lambda k: fn((k * 4) % max(1, C - 4), min(4, C - (k * 4) % max(1, C - 4))), range(threads * 2)))
                t0 = time.perf_counter()
                res = list(pool.map(lambda fc: fn(fc[0], fc[1]), parts))
                wall = time.perf_counter() - t0
        best, best_c = np.float32(-1.0), -1
        for sc_, c_ in res:
            if np.float32(sc_) > best:
                best, best_c = np.float32(sc_), c_
        done_c = sum(c for _, c in parts)
        return wall * (float(C) / done_c), (float(best), int(best_c))

    threads8 = min(8, cores)
    # reference layout (pointer-tree HybridGrid, a transformed copy of the cloud per candidate): the full loop on ONE
    # thread -- how the reference runs this path, and what `value` is -- then on 8 threads and on every core of the box
    t_ref_1, win_1 = full_loop(ref_range, 1)
    t_ref_8, win_8 = full_loop(ref_range, threads8)
    t_ref_all, win_all = full_loop(ref_range, usable)
    # BASELINE.md section 2, variant (ii) "fair-CPU": the same arithmetic on a flat leaf table, no allocation per
    # candidate.  8 threads and all cores: the full loop; one thread: an evenly spread eighth of it, scaled (the full loop
    # of the reference layout above is the unsampled one-thread figure; this one is bounded to keep the bench short)
    t_fair_8, fwin_8 = full_loop(fair_range, threads8)
    t_fair_all, fwin_all = full_loop(fair_range, usable)
    chunks, done = 16, 0
    per_chunk = max(1, (M if sampled else C) // (8 * chunks))
    t = time.perf_counter()
    for k in range(chunks):
        first = (C // chunks) * k
        cnt = min(per_chunk, C - first)
        fair_range(first, cnt)
        done += cnt
    t_fair_1 = (time.perf_counter() - t) / done * C
    same_winner = win_1 == win_8 == win_all == fwin_8 == fwin_all
    how = "full loop" if not sampled else "%d of %d candidates in evenly spread chunks, scaled (the full loop is %.1e lookups)" % (M, C, float(C) * n_pts)
    t = time.perf_counter()
    r = orc.csm3d_match(CSM_OPTS, init[:3], init, [(pts, og_hi), (pts, og_lo)])
    t_csm = time.perf_counter() - t
    from dliom import synth
    world_pts = synth.transform_points(sc["truth"], pts)
    origin = sc["truth"][:3].astype(np.float32)
    near = world_pts[np.linalg.norm((world_pts - origin).astype(np.float64), axis=1) <= HIGH_RES_MAX_RANGE]
    t = time.perf_counter()
    og_hi.insert_tables(origin, near, ins.hit_table, ins.miss_table, FREE)
    og_lo.insert_tables(origin, world_pts, ins.hit_table, ins.miss_table, FREE)
    t_ins = time.perf_counter() - t
    rest = t_csm + t_ins

    def entry(t_rtcsm, threads, **extra):
        d = {"seconds_per_scan": t_rtcsm + rest, "value": 1.0 / (t_rtcsm + rest), "rtcsm_seconds": t_rtcsm, "cores": threads}
        d.update(extra)
        return d

    fastest = min((t_ref_8, threads8, "reference layout"), (t_ref_all, usable, "reference layout"),
                  (t_fair_8, threads8, "fair-CPU"), (t_fair_all, usable, "fair-CPU"))
    per_scan = t_ref_1 + rest
    return {
        "value": 1.0 / per_scan, "unit": "scans/s", "cores": 1, "kind": "port",
        "host_cores_available": cores,
        "host_cpu_quota": host_cpu_quota(),  # what the container may actually use (cgroup), when it says: "all cores" above is os.cpu_count()
        "seconds_per_scan": per_scan,
        "stage_seconds": {"rtcsm": t_ref_1, "ceres": t_csm, "insert": t_ins},
        "all_variants_same_winner": bool(same_winner),
        "reference_layout": {"what": "pointer-tree HybridGrid, per-candidate TransformPointCloud copy: the reference's code shape",
                             "1_thread": entry(t_ref_1, 1, sample=how),
                             "%d_threads" % threads8: entry(t_ref_8, threads8, sample=how),
                             "all_cores": entry(t_ref_all, usable, sample=how)},
        "fair_cpu": {"what": "BASELINE.md section 2 (ii): flat leaf table, no per-candidate allocation, same arithmetic, same scores",
                     "1_thread": entry(t_fair_1, 1, sample="%d of %d candidates in %d evenly spread chunks, scaled" % (done, C, chunks)),
                     "%d_threads" % threads8: entry(t_fair_8, threads8, sample=how),
                     "all_cores": entry(t_fair_all, usable, sample=how)},
        "fastest_cpu_variant_measured": {"value": 1.0 / (fastest[0] + rest), "unit": "scans/s", "cores": fastest[1], "layout": fastest[2],
                                         "what": "the fastest of {reference layout, fair-CPU} x {%d threads, all %d usable cores (cgroup "
                                                 "quota; os.cpu_count() = %d)} for the candidate loop; CeresScanMatcher3D and insertion "
                                                 "on one thread, as the reference runs them (they bound this figure: %.3f s of %.3f s)" %
                                                 (threads8, usable, cores, rest, fastest[0] + rest)},
        "sample": ("full loop: the oracle (C++ restatement of the reference, g++ -O3) runs ALL %d candidates x %d points of the "
                   "same scan on the same grids on one thread (%.1f s), nothing sampled or scaled; CeresScanMatcher3D (%d "
                   "evaluations) and both insertions timed in full" % (C, n_pts, t_ref_1, r["num_residual_evaluations"]))
                  if not sampled else
                  ("the oracle on one thread over %s; CeresScanMatcher3D (%d evaluations) and both insertions timed in full"
                   % (how, r["num_residual_evaluations"])),
    }


if __name__ == "__main__":
    main()
