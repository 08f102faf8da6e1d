"""Parity at the sizes bench.py measures ("pin parity on the configurations you benchmark").

Config 2 (W-dense): the exact bench scene -- 20 map scans, a 64 x 1024 scan with every return matched,
C = 35 937 candidates, N = 65 536 points -- against the FULL oracle: the reference's candidate loop over
all candidates (8 host threads over contiguous ranges, combined in generation order), the whole integer
score volume, CeresScanMatcher3D, and both grids after the insertion.  The same scene with the search
window cut into 8 candidate shards.  Config 5 (128 x 2048 @ 5 cm, bits = 4 mirror, T = 343 translations =
13 passes of the box kernel) on a reduced angular window so that the oracle finishes in seconds, and at the
benchmarked 1 degree window (C = 3 176 523) on 10 000 random candidates plus the neighbourhood of the winner.
The "redo on the dense kernel" path of the box kernel (fault injected) and the three fallback score kernels at
config 2's size.
"""
import os

import numpy as np
import pytest

from helpers import (DEFAULT_CSM, DEFAULT_RTCSM, FREE, build_device_scene, device_cells_sorted,
                     device_grid_to_oracle, oracle_cells_sorted, pose_distance)

pytestmark = pytest.mark.gpu
THREADS = min(8, os.cpu_count() or 1)
THREADS_BIG = min(32, os.cpu_count() or 1)


@pytest.fixture(scope="module")
def dl():
    import dliom
    dliom.load_library()
    assert dliom.device_count() > 0, "no HIP device: the GPU tests must not pass on a fallback"
    return dliom


@pytest.fixture(scope="module")
def ctx(dl):
    c = dl.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def bench_scene(dl, ctx, orc):
    ins, g_hi, g_lo, scans = build_device_scene(dl, ctx, 64, 1024, 0.10, 0.45, map_scans=20)
    sc = scans[0]
    og_hi, og_lo = device_grid_to_oracle(orc, g_hi), device_grid_to_oracle(orc, g_lo)
    ref = orc.rtcsm3d_match_parallel(DEFAULT_RTCSM, sc["init"], sc["pts"], og_hi, threads=THREADS)
    yield dict(ins=ins, g_hi=g_hi, g_lo=g_lo, sc=sc, og_hi=og_hi, og_lo=og_lo, ref=ref)
    sc["cloud"].close()
    g_hi.close()
    g_lo.close()


def test_config2_full_oracle_match(dl, ctx, orc, bench_scene):
    s = bench_scene
    sc, ref = s["sc"], s["ref"]
    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, DEFAULT_RTCSM)
    score, pose = rt.Match(sc["init"], sc["cloud"], s["g_hi"])
    st = rt.last_stats()
    assert st.window.num_candidates == 35937 == ref["num_candidates"] and st.num_points == 65536
    assert st.score_kernel == 3 and st.box_kernel_status == dl.BOX_RAN  # the LDS-box kernel is what bench.py times
    assert st.box_kernel_variant == 0  # one pass of 27 translations: the four-waves-per-SIMD instantiation
    assert st.best_index == ref["best_index"], (st.best_index, ref["best_index"])
    assert np.float32(score).tobytes() == np.float32(ref["score"]).tobytes()
    assert np.array_equal(pose, ref["pose"])
    assert rt.box_error() == 0


def test_config2_full_score_volume(dl, ctx, orc, bench_scene):
    s = bench_scene
    sc = s["sc"]
    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, DEFAULT_RTCSM)
    got = rt.score_volume(sc["init"], sc["pts"], s["g_hi"])
    want = orc.rtcsm3d_value_sums_parallel(DEFAULT_RTCSM, sc["init"], sc["pts"], s["og_hi"], threads=THREADS)
    assert got.shape == want.shape == (35937,)
    assert np.array_equal(got.astype(np.uint64), want)
    assert rt.box_error() == 0


def test_config2_dense_rerun_after_box_fault():
    """The box kernel cross-checks its fast index; if the check ever fails the match is redone on the dense kernel
    (rtcsm3d.hip `box_overflowed`).  The kernel cannot be made to fail, so the flag is injected -- by a TEST BUILD of the
    library (-DDLIOM_TEST_HOOKS, libdliom_hooks.so: the shipped libdliom.so has no such switch and refuses the knob): this
    test runs tests/hooks_box_fault.py in a process of its own with that library loaded."""
    import subprocess
    import sys
    import dliom
    assert os.path.exists(dliom.HOOKS_LIB_PATH), "libdliom_hooks.so not built (make -C d-liom_amd hooks)"
    env = dict(os.environ, DLIOM_LIB=dliom.HOOKS_LIB_PATH)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hooks_box_fault.py")],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "hooks_box_fault ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_shipped_library_has_no_fault_injection(dl, ctx):
    with pytest.raises(dl.DliomError):
        ctx.set_tuning(dl.TUNE_RESERVED_TEST_HOOK, 1)
    assert ctx.get_tuning(dl.TUNE_RESERVED_TEST_HOOK) == 0


@pytest.mark.parametrize("n", [50017, 36001, 4129])
def test_config2_ragged_clouds_through_the_box_kernel(dl, ctx, orc, bench_scene, n):
    """Cloud sizes that are no multiple of the 32-point chunks the box kernel's dispenser hands out (most expensive first,
    a permutation built per cloud) nor of anything else: a last partial chunk, an odd number of chunks.  2 000 random
    candidates' integer sums against the oracle, and the winner."""
    s = bench_scene
    sc = s["sc"]
    pts = sc["pts"][np.sort(np.random.RandomState(n).choice(len(sc["pts"]), n, replace=False))]
    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, DEFAULT_RTCSM)
    sums = rt.score_volume(sc["init"], pts, s["g_hi"])
    idx = np.random.RandomState(n + 1).randint(0, len(sums), size=2000)
    want, _ = orc.rtcsm3d_at(DEFAULT_RTCSM, sc["init"], pts, s["og_hi"], idx, threads=THREADS)
    assert np.array_equal(sums[idx].astype(np.uint64), want)
    score, pose = rt.Match(sc["init"], pts, s["g_hi"])
    ref = orc.rtcsm3d_match_parallel(DEFAULT_RTCSM, sc["init"], pts, s["og_hi"], threads=THREADS)
    st = rt.last_stats()
    assert st.score_kernel == 3 and st.box_kernel_status == dl.BOX_RAN and st.num_points == n
    assert st.best_index == ref["best_index"]
    assert np.float32(score).tobytes() == np.float32(ref["score"]).tobytes() and np.array_equal(pose, ref["pose"])
    assert rt.box_error() == 0


def test_config2_every_score_kernel_gives_the_same_volume(dl, ctx, orc, bench_scene):
    """The fallback kernels (dense mirror, leaf table by rotation, leaf table by point: what runs when the search does
    not suit the box kernel or the grid has no mirror) produce the box kernel's volume bit for bit at the bench size."""
    s = bench_scene
    sc = s["sc"]
    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, DEFAULT_RTCSM)
    want = rt.score_volume(sc["init"], sc["pts"], s["g_hi"])
    try:
        for kernel in (2, 1, 0):
            ctx.set_tuning(dl.TUNE_SCORE_KERNEL, kernel)
            got = rt.score_volume(sc["init"], sc["pts"], s["g_hi"])
            assert np.array_equal(got, want), kernel
            score, pose = rt.Match(sc["init"], sc["cloud"], s["g_hi"])
            assert rt.last_stats().score_kernel == kernel
            assert np.float32(score).tobytes() == np.float32(s["ref"]["score"]).tobytes() and np.array_equal(pose, s["ref"]["pose"])
    finally:
        ctx.set_tuning(dl.TUNE_SCORE_KERNEL, 3)


def test_config2_eight_candidate_shards(dl, ctx, orc, bench_scene):
    """BASELINE config 4's split at config 2's size: 8 shards, each its own context; the max of the shards'
    words is the unsharded winner, which is the oracle's."""
    s = bench_scene
    sc, ref = s["sc"], s["ref"]
    ctxs = [dl.Context(0) for _ in range(8)]
    clouds = [dl.PointCloud(c, sc["pts"]) for c in ctxs]
    shards = [dl.RtcsmShard(c, DEFAULT_RTCSM, k, 8) for k, c in enumerate(ctxs)]
    glo = max(sh.begin(sc["init"], cl, s["g_hi"]) for sh, cl in zip(shards, clouds))
    gbest = max(sh.finish(glo) for sh in shards)
    for sh in shards:
        score, pose = sh.decode(gbest)
        assert np.float32(score).tobytes() == np.float32(ref["score"]).tobytes()
        assert np.array_equal(pose, ref["pose"])
    for c in clouds:
        c.close()
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("angular_deg,num_shards", [(0.2, 8), (1.0, 8), (1.0, 3)])
def test_wide_window_candidate_shards(dl, ctx, orc, bench_scene, angular_deg, num_shards):
    """Config 4's split on a window of 343 translations (the big-box instantiations, round 6): shards of 3-4 rotations
    (27 rotations over 8 shards: one wave per workgroup) and of 166 / 444 (1 331 rotations: three- and four-wave
    workgroups, the last block short) -- the max of the shards' words is the unsharded device winner; with the small
    window also the oracle's full loop."""
    s = bench_scene
    sc = s["sc"]
    opts = dict(DEFAULT_RTCSM, linear_search_window=0.35, angular_search_window=float(np.deg2rad(angular_deg)))
    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, opts)
    ref_score, ref_pose = rt.Match(sc["init"], sc["cloud"], s["g_hi"])
    st = rt.last_stats()
    assert st.window.num_translations == 343 and st.score_kernel == 3 and st.box_kernel_variant == 2
    ctxs = [dl.Context(0) for _ in range(num_shards)]
    clouds = [dl.PointCloud(c, sc["pts"]) for c in ctxs]
    shards = [dl.RtcsmShard(c, opts, k, num_shards) for k, c in enumerate(ctxs)]
    glo = max(sh.begin(sc["init"], cl, s["g_hi"]) for sh, cl in zip(shards, clouds))
    gbest = max(sh.finish(glo) for sh in shards)
    for sh in shards:
        score, pose = sh.decode(gbest)
        assert np.float32(score).tobytes() == np.float32(ref_score).tobytes() and np.array_equal(pose, ref_pose)
    if angular_deg < 0.5:
        flat = orc.FlatGridIndex(s["og_hi"])
        _, want = orc.rtcsm3d_volume_fair(opts, sc["init"], sc["pts"], flat, threads=THREADS_BIG)
        assert st.best_index == int(np.argmax(want)) and np.float32(ref_score).tobytes() == want[int(np.argmax(want))].tobytes()
    for c in clouds:
        c.close()
    for c in ctxs:
        c.close()


def test_config2_ceres_and_insertion(dl, ctx, orc, bench_scene):
    s = bench_scene
    sc, ref = s["sc"], s["ref"]
    cs = dl.CeresScanMatcher3D(ctx, DEFAULT_CSM)
    p2, summ = cs.Match(sc["init"][:3], ref["pose"], [(sc["cloud"], s["g_hi"]), (sc["cloud"], s["g_lo"])])
    r2 = orc.csm3d_match(DEFAULT_CSM, sc["init"][:3], ref["pose"], [(sc["pts"], s["og_hi"]), (sc["pts"], s["og_lo"])])
    dt, dr = pose_distance(p2, r2["pose"])
    assert dt <= 1e-6 and dr <= 1e-6, (dt, dr)  # metres / radians; north-star bar 1e-4 m
    assert summ["num_iterations"] == r2["num_iterations"]
    # insertion at the matched pose: both grids bit-equal afterwards (must run last: it changes the grids)
    pf = np.asarray(p2, dtype=np.float32)
    dl.insert_cloud_multi(s["ins"], sc["cloud"], [(s["g_hi"], [pf], 20.0), (s["g_lo"], [pf], 0.0)])
    world = orc.transform_points(pf, sc["pts"])
    origin = orc.transform_points(pf, np.zeros((1, 3), np.float32))[0]
    d = (world - origin).astype(np.float32)  # FilterRangeDataByMaxRange: float norm, Eigen's reduction order
    nrm = np.sqrt(d[:, 0] * d[:, 0] + (d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]), dtype=np.float32)
    near = world[nrm <= np.float32(20.0)]
    s["og_hi"].insert_tables(origin, near, s["ins"].hit_table, s["ins"].miss_table, FREE)
    s["og_lo"].insert_tables(origin, world, s["ins"].hit_table, s["ins"].miss_table, FREE)
    for dg, og in ((s["g_hi"], s["og_hi"]), (s["g_lo"], s["og_lo"])):
        dk, dv = device_cells_sorted(dg)
        ok, ov = oracle_cells_sorted(og)
        assert np.array_equal(dk, ok) and np.array_equal(dv, ov)


def test_config5_reduced_window(dl, ctx, orc):
    """128 x 2048 @ 5 cm: N = 262 144, the 2 GiB bits = 4 mirror, T = 343 translations (13 box-kernel passes);
    angular window 0.2 degrees so that the oracle's candidate loop finishes in seconds."""
    opts = dict(DEFAULT_RTCSM, angular_search_window=float(np.deg2rad(0.2)))
    ins, g_hi, g_lo, scans = build_device_scene(dl, ctx, 128, 2048, 0.05, 0.45, map_scans=3)
    sc = scans[0]
    assert g_hi.bits == 4
    og_hi = device_grid_to_oracle(orc, g_hi)
    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, opts)
    score, pose = rt.Match(sc["init"], sc["cloud"], g_hi)
    st = rt.last_stats()
    assert st.window.num_translations == 343 and st.num_points == 262144
    assert st.score_kernel == 3  # LDS-box kernel; N > box::kFlushPoints exercises the accumulator flush
    assert st.box_kernel_variant == 2  # 343 translations: 49 per pass, one z-plane of the window (round 6)
    ref = orc.rtcsm3d_match_parallel(opts, sc["init"], sc["pts"], og_hi, threads=THREADS)
    assert st.best_index == ref["best_index"], (st.best_index, ref["best_index"])
    assert np.float32(score).tobytes() == np.float32(ref["score"]).tobytes()
    assert np.array_equal(pose, ref["pose"])
    sums = rt.score_volume(sc["init"], sc["pts"], g_hi)
    rng = np.random.RandomState(5)
    for c in rng.randint(0, len(sums), size=48):
        assert sums[c] == orc.rtcsm3d_value_sums(opts, sc["init"], sc["pts"], og_hi, first=int(c), count=1)[0]
    assert rt.box_error() == 0
    sc["cloud"].close()
    g_hi.close()
    g_lo.close()


@pytest.mark.parametrize("linear_window,translations,variant", [(0.25, 125, 1), (0.35, 343, 2), (0.45, 729, 2)])
def test_box_kernel_instantiations_for_windows_of_several_passes(dl, ctx, orc, bench_scene, linear_window, translations, variant):
    """Round 6: a translation window of more than one 27-translation pass runs the box kernel at three waves per SIMD with
    21 000-cell boxes and 64-point chunks -- 27 translations per pass (variant 1) or 49 (variant 2: one staged box and one
    rotation per point for 49 accumulators; short last passes are padded).  On the config-2 scene with a
    0.2 degree angular window so that the oracle's FULL loop is seconds: every candidate's integer sum, the winner's
    index, score bits and pose."""
    g_hi, sc, og_hi = bench_scene["g_hi"], bench_scene["sc"], bench_scene["og_hi"]
    opts = dict(DEFAULT_RTCSM, linear_search_window=linear_window, angular_search_window=float(np.deg2rad(0.2)))
    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, opts)
    score, pose = rt.Match(sc["init"], sc["cloud"], g_hi)
    st = rt.last_stats()
    assert st.window.num_translations == translations and st.num_points == 65536
    assert st.score_kernel == 3 and st.box_kernel_status == dl.BOX_RAN and st.box_kernel_variant == variant, (
        st.score_kernel, st.box_kernel_status, st.box_kernel_variant)
    flat = orc.FlatGridIndex(og_hi)
    want_sums, want_scores = orc.rtcsm3d_volume_fair(opts, sc["init"], sc["pts"], flat, threads=THREADS_BIG)
    sums = rt.score_volume(sc["init"], sc["pts"], g_hi)
    assert np.array_equal(sums.astype(np.uint64), want_sums)
    best = int(np.argmax(want_scores))
    _, ca = orc.rtcsm3d_candidates(opts, 0.1, sc["pts"], sc["init"])
    assert st.best_index == best and np.float32(score).tobytes() == want_scores[best].tobytes()
    assert np.array_equal(pose, ca[best].astype(np.float64))
    assert rt.box_error() == 0


def test_config5_benchmarked_window_sampled(dl, ctx, orc):
    """Config 5 with the window bench.py --config 5 searches: 128 x 2048 @ 5 cm, the full 1 degree angular window
    (17^3 rotations on this scene, 21^3 on bench.py's longer-range one) x 343 translations, C = 1 685 159 candidates,
    N = 262 144.  The oracle's full loop would take hours, so: (a) 10 000 random
    candidates' integer sums against the oracle; (b) the winner: the device's index, score bits and pose against the
    oracle's ScoreCandidate of that candidate, and against the first maximum (generation order, strict >) of the oracle's
    exact scores over the 512 candidates that rank highest by the real-valued score computed from the (sample-checked)
    integer volume -- the float rounding of the reference's sequential sum moves a score by ~1e-6 relative, the 512th
    candidate is orders of magnitude further down."""
    opts = dict(DEFAULT_RTCSM)
    ins, g_hi, g_lo, scans = build_device_scene(dl, ctx, 128, 2048, 0.05, 0.45, map_scans=3)
    sc = scans[0]
    assert g_hi.bits == 4
    og_hi = device_grid_to_oracle(orc, g_hi)
    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, opts)
    score, pose = rt.Match(sc["init"], sc["cloud"], g_hi)
    st = rt.last_stats()
    C = st.window.num_candidates
    assert st.window.num_translations == 343 and st.num_points == 262144
    assert C == 343 * (2 * st.window.angular_window_size + 1) ** 3 and C > 1500000, C
    assert st.score_kernel == 3
    sums = rt.score_volume(sc["init"], sc["pts"], g_hi)
    assert len(sums) == C
    # (a)
    idx = np.random.RandomState(11).randint(0, C, size=10000)
    want, _ = orc.rtcsm3d_at(opts, sc["init"], sc["pts"], og_hi, idx, threads=THREADS_BIG)
    assert np.array_equal(sums[idx].astype(np.uint64), want)
    # (b)
    tr, ca = orc.rtcsm3d_candidates(opts, 0.05, sc["pts"], sc["init"])
    assert len(tr) == C
    t_norm = np.linalg.norm(tr[:, :3].astype(np.float64), axis=1)
    angle = 2.0 * np.arctan2(np.linalg.norm(tr[:, 4:7].astype(np.float64), axis=1), np.abs(tr[:, 3].astype(np.float64)))
    arg = t_norm * opts["translation_delta_cost_weight"] + angle * opts["rotation_delta_cost_weight"]
    k_scale = (0.9 - 0.1) / 32766.0
    real = (sums.astype(np.float64) * k_scale + (0.1 - k_scale) * st.num_points) / st.num_points * np.exp(-arg * arg)
    top = np.sort(np.argsort(-real, kind="stable")[:512])  # generation order
    assert st.best_index in top
    _, exact = orc.rtcsm3d_at(opts, sc["init"], sc["pts"], og_hi, top, threads=THREADS_BIG)
    best = int(top[int(np.argmax(exact))])  # argmax returns the first maximum
    assert real[np.argsort(-real, kind="stable")[511]] < real[st.best_index] * (1.0 - 1e-4)  # the cut is far below the top
    assert best == st.best_index, (best, st.best_index)
    assert np.float32(score).tobytes() == np.float32(exact.max()).tobytes()
    assert np.array_equal(pose, ca[best].astype(np.float64))
    assert rt.box_error() == 0
    sc["cloud"].close()
    g_hi.close()
    g_lo.close()


@pytest.mark.parametrize("case", ["bits5_10cm", "bits6_5cm"])
def test_windowed_mirror_runs_the_box_kernel_on_large_grids(dl, ctx, orc, case):
    """VERDICT r3, item 7.  A 10 cm grid with more than 51 m of extent has DynamicGrid bits 5 (a 5 cm grid: bits 6): no
    whole-grid mirror there, and until round 4 the correlative matcher fell to the leaf-table kernels (4.4 x slower).
    The matcher only reads within (farthest point + window) of its initial pose: that cube of the grid is mirrored
    (grid.hip, ensure_dense_for), the LDS-box kernel runs on it, and the winner is the oracle's.  Yard scene, returns
    inserted to 80 m; the matched scan cut at 55 m (30 m for the 5 cm grid) so that its cube fits the window's 1264 cells.
    Also: a second match a few metres on reuses the window, a third one far away rebuilds it, and an insertion in between
    writes through to the cells inside the window -- every score volume equals the leaf-table kernel's."""
    from dliom import synth
    res, cut = (0.10, 55.0) if case == "bits5_10cm" else (0.05, 30.0)
    ins = dl.RangeDataInserter3D(0.55, 0.49, 2, ctx=ctx)
    grid = dl.HybridGrid(ctx, res)
    opts = dict(DEFAULT_RTCSM)
    opts["angular_search_window"] = float(np.deg2rad(0.35))
    with synth.scene("ground"):
        for s in range(4):
            pose = synth.trajectory_pose(0.25 * s)
            pts, _ = synth.scan(pose, 32, 512)
            cloud = dl.PointCloud(ctx, pts)
            ins.InsertCloud(grid, cloud, poses=[pose.astype(np.float32)])  # no range cut: the grid reaches 80 m
            cloud.close()
        scans = []
        # 4 m/s: 2.4 m on, then far enough for a new window (the box kernel itself refuses lookups beyond ~880 cells of
        # the grid's origin -- its scaled coordinates must stay below 1024 --, 44 m at 5 cm: the third pose respects that)
        for t in (1.0, 1.6, 9.0 if case == "bits5_10cm" else 3.0):
            truth = synth.trajectory_pose(t)
            pts, _ = synth.scan(truth, 32, 512)
            pts = pts[np.linalg.norm(pts.astype(np.float64), axis=1) <= cut]
            scans.append((truth, pts, synth.perturb_pose(truth, 0.1, 0.3, seed=int(10 * t))))
        extra_pose = synth.trajectory_pose(1.3)
        extra, _ = synth.scan(extra_pose, 32, 512)
    assert grid.bits == (5 if case == "bits5_10cm" else 6), grid.bits
    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, opts)
    og = device_grid_to_oracle(orc, grid)
    for k, (truth, pts, init) in enumerate(scans):
        if k == 1:  # an insertion while the window exists: write-through inside it, nothing lost outside
            cloud = dl.PointCloud(ctx, extra)
            ins.InsertCloud(grid, cloud, poses=[extra_pose.astype(np.float32)])
            cloud.close()
            og = device_grid_to_oracle(orc, grid)
        score, pose = rt.Match(init, pts, grid)
        st = rt.last_stats()
        assert st.score_kernel == 3 and st.box_kernel_status == dl.BOX_RAN, (k, st.score_kernel, st.box_kernel_status)
        ref = orc.rtcsm3d_match_parallel(opts, init, pts, og, threads=THREADS)
        assert st.window.num_candidates == ref["num_candidates"]
        assert st.best_index == ref["best_index"], (k, st.best_index, ref["best_index"])
        assert np.float32(score).tobytes() == np.float32(ref["score"]).tobytes() and np.array_equal(pose, ref["pose"])
        got = rt.score_volume(init, pts, grid)
        ctx.set_tuning(dl.TUNE_SCORE_KERNEL, 1)  # the rotation-per-lane kernel over the leaf table
        try:
            leaf = rt.score_volume(init, pts, grid)
            score1, pose1 = rt.Match(init, pts, grid)
            assert rt.last_stats().score_kernel == 1
            assert np.float32(score1).tobytes() == np.float32(score).tobytes() and np.array_equal(pose1, pose)
        finally:
            ctx.set_tuning(dl.TUNE_SCORE_KERNEL, 3)
        assert np.array_equal(got, leaf), (k, int((got != leaf).sum()))
        assert rt.box_error() == 0
    # what the window buys (the matched scan resident; the first box match of a pose may build the window)
    import sys
    import time
    truth, pts, init = scans[0]
    cloud = dl.PointCloud(ctx, pts)
    times = {}
    for kernel in (3, 1):
        ctx.set_tuning(dl.TUNE_SCORE_KERNEL, kernel)
        rt.Match(init, cloud, grid)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            rt.Match(init, cloud, grid)
        ctx.synchronize()
        times[kernel] = (time.perf_counter() - t0) / 5
    ctx.set_tuning(dl.TUNE_SCORE_KERNEL, 3)
    sys.stderr.write("windowed mirror %s: RTCSM3D match %.3f ms on the box kernel, %.3f ms on the leaf-table kernel (C = %d, N = %d)\n" %
                     (case, 1e3 * times[3], 1e3 * times[1], rt.last_stats().window.num_candidates, len(pts)))
    assert times[3] < times[1]
    cloud.close()
    grid.close()


@pytest.mark.parametrize("offset", [(120.0, -60.0, 10.0), (-300.0, 250.0, -40.0)])
def test_box_kernel_far_from_the_grid_origin(dl, ctx, orc, offset):
    """Until round 5 the LDS-box kernel refused searches whose ABSOLUTE cell coordinates exceeded ~880 (88 m from the
    origin of a 10 cm grid) -- a limit that only the scan's range has to respect (Kb = 128 - rotated point's cell); the
    position enters the rounding band's width alone.  The yard scene shifted by 120 ... 300 m (cells up to +-3800: bits 6 / 7,
    windowed mirror): the box kernel runs, and winner, score bits, pose and the whole score volume are the oracle's /
    the leaf-table kernel's."""
    from dliom import synth
    off = np.array(list(offset) + [0.0, 0.0, 0.0, 0.0])
    ins = dl.RangeDataInserter3D(0.55, 0.49, 2, ctx=ctx)
    grid = dl.HybridGrid(ctx, 0.10)
    opts = dict(DEFAULT_RTCSM)
    opts["angular_search_window"] = float(np.deg2rad(0.35))
    with synth.scene("ground"):
        for s in range(3):
            pose = synth.trajectory_pose(0.25 * s)
            pts, _ = synth.scan(pose, 32, 512)
            cloud = dl.PointCloud(ctx, pts)
            ins.InsertCloud(grid, cloud, poses=[(pose + off).astype(np.float32)])  # the same world, moved by `offset`
            cloud.close()
        truth = synth.trajectory_pose(1.0)
        pts, _ = synth.scan(truth, 32, 512)
        pts = pts[np.linalg.norm(pts.astype(np.float64), axis=1) <= 40.0]
        init = synth.perturb_pose(truth, 0.1, 0.3, seed=5) + off
    assert grid.bits >= 6, grid.bits
    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, opts)
    og = device_grid_to_oracle(orc, grid)
    score, pose = rt.Match(init, pts, grid)
    st = rt.last_stats()
    assert st.score_kernel == 3 and st.box_kernel_status == dl.BOX_RAN, (st.score_kernel, st.box_kernel_status)
    ref = orc.rtcsm3d_match_parallel(opts, init, pts, og, threads=THREADS)
    assert st.window.num_candidates == ref["num_candidates"]
    assert st.best_index == ref["best_index"], (st.best_index, ref["best_index"])
    assert np.float32(score).tobytes() == np.float32(ref["score"]).tobytes() and np.array_equal(pose, ref["pose"])
    got = rt.score_volume(init, pts, grid)
    ctx.set_tuning(dl.TUNE_SCORE_KERNEL, 1)
    try:
        leaf = rt.score_volume(init, pts, grid)
    finally:
        ctx.set_tuning(dl.TUNE_SCORE_KERNEL, 3)
    assert np.array_equal(got, leaf), int((got != leaf).sum())
    assert rt.box_error() == 0
    assert grid.mirror_stats()[2]  # a window of the grid is mirrored, not the grid
    grid.close()
