"""BASELINE config 5 (128 x 2048 returns, 5 cm voxels, two active submaps) lines of bench.py."""
import os
import sys
import time

import numpy as np

from benchlib import (ROOT, RTCSM_OPTS, CSM_OPTS, HIT_P, MISS_P, FREE, HIGH_RES_MAX_RANGE, HBM_PEAK_GBS, VALU_PEAK_LANE_OPS,
                      build_scene, insertion_targets)
from benchlib.roofline import USEFUL_PAIRS_PER_S


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



def config5_scene(dl, synth, ctx):
    """BASELINE config 5's scene: two active submaps (four grids: hi + lo of each, submap_3d.cc:303-314) built from three
    128 x 2048 scans at ground truth, the newer submap holding the later half, and the fourth scan to match.  The sensor's
    beams span +-35 degrees here so that returns reach the 30 m cube's corners (26 m): the window BASELINE.md section 3
    states, C = 343 x 19^3 = 2 352 637 (with the +-15 degrees of config 2 the farthest return is 23.5 m away and
    C = 1 685 159).  Returns (inserter, [A.hi, A.lo, B.hi, B.lo], scan dict, (res_hi, res_lo))."""
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
    return ins, grids, sc, (res_hi, res_lo)


def device_grid_to_oracle(orc, dg, resolution):
    """The oracle's HybridGrid holding exactly the device grid's cells (download of the leaf pool)."""
    origins, values = dg.download_blocks()
    og = orc.HybridGrid(resolution)
    leaf, cell = np.nonzero(values)
    xyz = np.stack([origins[leaf, 0] + (cell & 7), origins[leaf, 1] + ((cell >> 3) & 7), origins[leaf, 2] + (cell >> 6)],
                   axis=1).astype(np.int32)
    og.set_values(xyz, values[leaf, cell])
    return og


def config5_line(dl, synth, ctx, steps=3, with_oracle=True):
    """BASELINE config 5 as BASELINE.json words it -- "128-beam x 2048 dense cloud, 5 cm voxels, multi-submap insertion +
    scan match" -- short enough for the default N = 1 line, so that the DRIVER times it: the scan matched against the older
    submap's 5 cm grid (RTCSM3D over C = 343 x 19^3 candidates + CeresScanMatcher3D hi + lo) and inserted into all four
    grids by the fused insertion (config5_scene).  Parity in this line: sampled candidates (the full loop is 6e11 lookups,
    minutes on every core of the box); the SAME scene's full loop -- every candidate's integer sum and reference score, the
    winner -- is tools/config5_full_parity.py, its record profiles/r6_config5_full_parity.json."""
    ins, grids, sc, (res_hi, res_lo) = config5_scene(dl, synth, ctx)
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
           "score_kernel": int(st.score_kernel), "box_kernel_variant": int(st.box_kernel_variant),
           "score_kernel_ms": k_ms / max(k_n, 1),
           "pairs_per_s": float(C) * n / (k_ms / max(k_n, 1) * 1e-3) if k_ms > 0 else None,
           "frac_useful": (float(C) * n / (k_ms / max(k_n, 1) * 1e-3)) / USEFUL_PAIRS_PER_S if k_ms > 0 else None,
           "mirror": {"bytes": mirror_bytes, "windowed": windowed, "rebuilds": rebuilds}, "box_kernel_flags": int(rt.box_error())}
    if with_oracle:
        from oracle import oracle as orc
        from benchlib.cpu_legs import sampled_oracle_match
        og = device_grid_to_oracle(orc, grids[0], res_hi)
        score, p1 = rt.Match(sc["init"], sc["cloud"], grids[0])
        st = rt.last_stats()
        threads = min(32, os.cpu_count() or 1)
        ref, sampled = sampled_oracle_match(orc, rt, sc, grids[0], og, st, threads, sample=1000, top_n=256)
        out["parity"] = {"ok": bool(sampled["ok"] and int(st.best_index) == ref["best_index"] and
                                    np.float32(score).tobytes() == np.float32(ref["score"]).tobytes() and np.array_equal(p1, ref["pose"])),
                         "how": sampled["how"], "oracle_threads": threads,
                         "full_size_record": full_parity_record()}
    sc["cloud"].close()
    for g in grids:
        g.close()
    return out



def full_parity_record():
    """What tools/config5_full_parity.py found when it ran the oracle's FULL loop on this scene (committed record)."""
    import json
    path = os.path.join(ROOT, "profiles", "r6_config5_full_parity.json")
    try:
        r = json.load(open(path))
        return {"file": "profiles/r6_config5_full_parity.json", "ok": r.get("ok"), "candidates_compared": r.get("C"),
                "volume_mismatches": r.get("volume_mismatches"), "winner_equal": r.get("winner_equal")}
    except Exception:
        return None
