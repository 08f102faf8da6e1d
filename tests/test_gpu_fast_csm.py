"""FastCorrelativeScanMatcher3D on the device vs the oracle restatement (tests/test_oracle_kat.py
pins the oracle to the reference's own known-answer tests): pyramid levels bit-exact, match results
(found, score bits, pose, rotational and low-resolution scores) identical."""
import numpy as np
import pytest

from helpers import build_oracle_submap, to_device_grid

pytestmark = pytest.mark.gpu

KAT_CLOUD = np.array([[4, 0, 0], [4.5, 0, 0], [5, 0, 0], [5.5, 0, 0], [0, 4, 0], [0, 4.5, 0], [0, 5, 0], [0, 5.5, 0],
                      [0, 0, 4], [0, 0, 4.5], [0, 0, 5], [0, 0, 5.5]], dtype=np.float32)
KAT_OPTS = dict(branch_and_bound_depth=6, full_resolution_depth=6, min_rotational_score=0.1,
                min_low_resolution_score=0.15, linear_xy_search_window=0.8, linear_z_search_window=0.8,
                angular_search_window=0.3)
IDENT = np.array([0, 0, 0, 1.0, 0, 0, 0])


@pytest.fixture(scope="module")
def dl():
    import dliom
    return dliom


@pytest.fixture(scope="module")
def ctx(dl):
    c = dl.Context(0)
    yield c
    c.close()


def level_dict(lo, values):
    z, y, x = np.nonzero(values)
    return {(int(a + lo[0]), int(b + lo[1]), int(c + lo[2])): int(v) for a, b, c, v in zip(x, y, z, values[z, y, x])}


def kat_scene(orc, pose_t, theta):
    """The reference test's submap: the 12-point cloud inserted at `pose` (hit 0.7, miss 0.4, 5 free voxels)."""
    q = np.array([np.cos(0.5 * theta), 0, 0, np.sin(0.5 * theta)])
    pose = np.concatenate([pose_t, q]).astype(np.float32)
    g = orc.HybridGrid(0.05)
    hit = orc.lookup_table_to_apply_odds(orc.odds(0.7))
    miss = orc.lookup_table_to_apply_odds(orc.odds(0.4))
    g.insert_tables(pose[:3], orc.transform_points(pose, KAT_CLOUD), hit, miss, 5)
    return g, pose


def same_result(a, b):
    assert a["found"] == b["found"]
    if a["found"]:
        assert np.float32(a["score"]) == np.float32(b["score"])
        assert np.array_equal(np.asarray(a["pose"]), np.asarray(b["pose"]))
        assert np.float32(a["rotational_score"]) == np.float32(b["rotational_score"])
        assert np.float32(a["low_resolution_score"]) == np.float32(b["low_resolution_score"])
    assert a["num_discrete_scans"] == b["num_discrete_scans"]


@pytest.mark.parametrize("frd", [6, 3, 1])
def test_pyramid_levels_equal_oracle(dl, ctx, orc, frd):
    og = build_oracle_submap(orc, 0.1, num_scans=3, beams=16, azimuths=128, max_range=20.0)
    dg = to_device_grid(dl, ctx, og)
    opts = dict(KAT_OPTS, full_resolution_depth=frd)
    hist = np.zeros((1, 10), np.float32)
    om = orc.FastCorrelativeScanMatcher3D(og, og, hist, [0.0], opts)
    dm = dl.FastCorrelativeScanMatcher3D(ctx, dg, dg, hist, [0.0], opts)
    for depth in range(opts["branch_and_bound_depth"]):
        xyz, v = om.stack_cells(depth)
        want = {(int(c[0]), int(c[1]), int(c[2])): int(val) for c, val in zip(xyz, v)}
        lo, values = dm.level(depth)
        assert level_dict(lo, values) == want, depth
    dm.close()
    dg.close()


def test_reference_kat_poses_match_oracle(dl, ctx, orc):
    """fast_correlative_scan_matcher_3d_test.cc: Match and MatchFullSubmap on the 12-point scene for a few
    poses, and the far low-resolution cloud that must not match."""
    rng = np.random.RandomState(5)
    for i in range(4):
        t = 0.7 * rng.uniform(-1, 1, 3)
        theta = 0.2 * rng.uniform(-1, 1)
        og, pose = kat_scene(orc, t, theta)
        dg = to_device_grid(dl, ctx, og)
        hist = np.zeros((1, 10), np.float32)
        yaw = [float(theta)]
        om = orc.FastCorrelativeScanMatcher3D(og, og, hist, yaw, KAT_OPTS)
        dm = dl.FastCorrelativeScanMatcher3D(ctx, dg, dg, hist, yaw, KAT_OPTS)
        data = dict(gravity_alignment=[1, 0, 0, 0], high_resolution_point_cloud=KAT_CLOUD,
                    low_resolution_point_cloud=KAT_CLOUD, rotational_scan_matcher_histogram=np.zeros(10, np.float32))
        ro = om.Match(IDENT, IDENT, data, 0.1)
        rd = dm.Match(IDENT, IDENT, data, 0.1)
        assert ro["found"]
        same_result(rd, ro)
        assert np.linalg.norm(rd["pose"][:3] - t) < 0.05
        far = dict(data, low_resolution_point_cloud=np.array([[42, 42, 42]], np.float32))
        same_result(dm.Match(IDENT, IDENT, far, 0.1), om.Match(IDENT, IDENT, far, 0.1))
        if i == 0:
            same_result(dm.MatchFullSubmap(IDENT[3:], IDENT[3:], data, 0.1), om.MatchFullSubmap(IDENT[3:], IDENT[3:], data, 0.1))
            guess = np.concatenate([t + 0.1, [1, 0, 0, 0]])
            same_result(dm.MatchWith3DofInitial(guess, data, 0.1), om.MatchWith3DofInitial(guess, data, 0.1))
        dm.close()
        dg.close()


@pytest.mark.parametrize("frd,depth", [(3, 6), (2, 5)])
def test_loop_closure_on_synthetic_submap_equals_oracle(dl, ctx, orc, frd, depth):
    """A 16x256 scan against a 6-scan submap with half-resolution pyramid levels, real rotational
    histograms (ComputeHistogram) and a displaced, yawed node pose."""
    from dliom import synth
    og_hi = build_oracle_submap(orc, 0.2, num_scans=6, beams=16, azimuths=256, max_range=40.0)
    og_lo = build_oracle_submap(orc, 0.5, num_scans=6, beams=16, azimuths=256)
    g_hi, g_lo = to_device_grid(dl, ctx, og_hi), to_device_grid(dl, ctx, og_lo)
    opts = dict(branch_and_bound_depth=depth, full_resolution_depth=frd, min_rotational_score=0.3,
                min_low_resolution_score=0.3, linear_xy_search_window=3.0, linear_z_search_window=1.0,
                angular_search_window=np.deg2rad(20.0))
    node_hists, node_yaws = [], []
    for s in range(6):
        pose = synth.trajectory_pose(0.1 * s)
        pts, _ = synth.scan(pose, 16, 256)
        node_hists.append(orc.compute_histogram(pts, 30))
        node_yaws.append(float(np.arctan2(2 * (pose[3] * pose[6] + pose[4] * pose[5]), 1 - 2 * (pose[5] ** 2 + pose[6] ** 2))))
    om = orc.FastCorrelativeScanMatcher3D(og_hi, og_lo, np.array(node_hists), node_yaws, opts)
    dm = dl.FastCorrelativeScanMatcher3D(ctx, g_hi, g_lo, np.array(node_hists), node_yaws, opts)
    truth = synth.trajectory_pose(0.35)
    pts, _ = synth.scan(truth, 16, 256)
    hi_pts = orc.adaptive_voxel_filter(2.0, 150, 15.0, pts)
    lo_pts = orc.adaptive_voxel_filter(4.0, 200, 60.0, pts)
    data = dict(gravity_alignment=[1, 0, 0, 0], high_resolution_point_cloud=hi_pts, low_resolution_point_cloud=lo_pts,
                rotational_scan_matcher_histogram=orc.compute_histogram(pts, 30))
    node_pose = synth.perturb_pose(truth, 1.5, 8.0, seed=3)
    for min_score in (0.2, 0.45):
        ro = om.Match(node_pose, IDENT, data, min_score)
        rd = dm.Match(node_pose, IDENT, data, min_score)
        same_result(rd, ro)
    assert ro["num_discrete_scans"] > 3
    dm.close()
    for g in (g_hi, g_lo):
        g.close()


def test_one_matcher_many_threads(dl, ctx, orc):
    """FastCorrelativeScanMatcher3D::Match is const and called concurrently from the ConstraintBuilder3D pool
    threads on ONE per-submap matcher (constraint_builder_3d.cc:270-275): four threads, each with its own
    context, hammer the same matcher with different node poses; every result equals the serial one."""
    import threading
    from dliom import synth
    og_hi = build_oracle_submap(orc, 0.2, num_scans=6, beams=16, azimuths=256, max_range=40.0)
    og_lo = build_oracle_submap(orc, 0.5, num_scans=6, beams=16, azimuths=256)
    g_hi, g_lo = to_device_grid(dl, ctx, og_hi), to_device_grid(dl, ctx, og_lo)
    opts = dict(branch_and_bound_depth=6, full_resolution_depth=3, min_rotational_score=0.3, min_low_resolution_score=0.3,
                linear_xy_search_window=3.0, linear_z_search_window=1.0, angular_search_window=np.deg2rad(20.0))
    hists, yaws = [], []
    for s in range(6):
        pose = synth.trajectory_pose(0.1 * s)
        pts, _ = synth.scan(pose, 16, 256)
        hists.append(orc.compute_histogram(pts, 30))
        yaws.append(float(np.arctan2(2 * (pose[3] * pose[6] + pose[4] * pose[5]), 1 - 2 * (pose[5] ** 2 + pose[6] ** 2))))
    dm = dl.FastCorrelativeScanMatcher3D(ctx, g_hi, g_lo, np.array(hists), yaws, opts)
    truth = synth.trajectory_pose(0.35)
    pts, _ = synth.scan(truth, 16, 256)
    data = dict(gravity_alignment=[1, 0, 0, 0], high_resolution_point_cloud=orc.adaptive_voxel_filter(2.0, 150, 15.0, pts),
                low_resolution_point_cloud=orc.adaptive_voxel_filter(4.0, 200, 60.0, pts),
                rotational_scan_matcher_histogram=orc.compute_histogram(pts, 30))
    node_poses = [synth.perturb_pose(truth, 1.0 + 0.2 * k, 6.0, seed=30 + k) for k in range(4)]
    serial = [dm.Match(p, IDENT, data, 0.2) for p in node_poses]
    results = [[None] * 6 for _ in node_poses]
    ctxs = [dl.Context(0) for _ in node_poses]

    def worker(k):
        for rep in range(6):
            results[k][rep] = dm.Match(node_poses[k], IDENT, data, 0.2, ctx=ctxs[k])

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(node_poses))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k, want in enumerate(serial):
        for got in results[k]:
            assert got is not None
            same_result(got, want)
    for c in ctxs:
        c.close()
    dm.close()
    for g in (g_hi, g_lo):
        g.close()
