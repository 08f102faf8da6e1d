"""Golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py from the pinned
oracle): the oracle must keep reproducing them (CPU), and the device path must reproduce them
through the C ABI (GPU) -- bit-exact for cells, sums, survivors and the correlative pose, 1e-6 for
the Ceres pose."""
import os

import numpy as np
import pytest

from helpers import DEFAULT_CSM, DEFAULT_RTCSM, pose_distance

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def sorted_cells(xyz, v):
    order = np.lexsort((xyz[:, 0], xyz[:, 1], xyz[:, 2]))
    return xyz[order].astype(np.int32), v[order].astype(np.uint16)


def oracle_grid_from(orc, resolution, xyz, values):
    g = orc.HybridGrid(float(resolution))
    g.set_values(xyz, values)
    return g


def test_oracle_reproduces_insertion_fixture(orc):
    f = load("insertion.npz")
    g = orc.HybridGrid(float(f["resolution"]))
    hit = orc.lookup_table_to_apply_odds(orc.odds(float(f["hit_probability"])))
    miss = orc.lookup_table_to_apply_odds(orc.odds(float(f["miss_probability"])))
    for o, r in zip(f["origins"], f["returns"]):
        g.insert_tables(o, r, hit, miss, int(f["num_free_space_voxels"]))
    xyz, v = sorted_cells(*g.export_cells())
    assert np.array_equal(xyz, f["cell_xyz"]) and np.array_equal(v, f["cell_value"])


def test_oracle_reproduces_matching_fixture(orc):
    f = load("matching.npz")
    g_hi = oracle_grid_from(orc, 0.1, f["hi_cell_xyz"], f["hi_cell_value"])
    g_lo = oracle_grid_from(orc, 0.45, f["lo_cell_xyz"], f["lo_cell_value"])
    r = orc.rtcsm3d_match(DEFAULT_RTCSM, f["initial_pose"], f["points"], g_hi)
    assert np.array_equal(r["pose"], f["rtcsm_pose"]) and np.float32(r["score"]) == f["rtcsm_score"]
    sums = orc.rtcsm3d_value_sums(DEFAULT_RTCSM, f["initial_pose"], f["points"], g_hi)
    assert np.array_equal(sums, f["score_volume_sums"])
    c = orc.csm3d_match(DEFAULT_CSM, f["initial_pose"][:3], r["pose"], [(f["points"], g_hi), (f["points"], g_lo)])
    assert np.array_equal(c["pose"], f["csm_pose"]) and c["num_iterations"] == int(f["csm_iterations"])


def test_oracle_reproduces_voxel_filter_fixture(orc):
    f = load("voxel_filter.npz")
    assert np.array_equal(orc.voxel_filter(0.15, f["points"]), f["kept_015"])
    assert np.array_equal(orc.adaptive_voxel_filter(2.0, 150, 15.0, f["points"]), f["adaptive_hi"])
    assert np.array_equal(orc.adaptive_voxel_filter(4.0, 200, 60.0, f["points"]), f["adaptive_lo"])


@pytest.fixture(scope="module")
def dl():
    import dliom
    return dliom


@pytest.fixture(scope="module")
def ctx(dl):
    c = dl.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
def test_device_reproduces_insertion_fixture(dl, ctx):
    f = load("insertion.npz")
    g = dl.HybridGrid(ctx, float(f["resolution"]))
    ins = dl.RangeDataInserter3D(float(f["hit_probability"]), float(f["miss_probability"]),
                                 int(f["num_free_space_voxels"]))
    for o, r in zip(f["origins"], f["returns"]):
        ins.Insert(o, r, g)
    cells = g.cells()
    want = {(int(c[0]), int(c[1]), int(c[2])): int(v) for c, v in zip(f["cell_xyz"], f["cell_value"])}
    assert cells == want
    g.close()


@pytest.mark.gpu
def test_device_reproduces_matching_fixture(dl, ctx):
    f = load("matching.npz")
    g_hi, g_lo = dl.HybridGrid(ctx, 0.1), dl.HybridGrid(ctx, 0.45)
    g_hi.set_values(f["hi_cell_xyz"], f["hi_cell_value"])
    g_lo.set_values(f["lo_cell_xyz"], f["lo_cell_value"])
    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, DEFAULT_RTCSM)
    score, pose = rt.Match(f["initial_pose"], f["points"], g_hi)
    assert np.array_equal(pose, f["rtcsm_pose"])
    assert np.float32(score) == f["rtcsm_score"]
    sums = rt.score_volume(f["initial_pose"], f["points"], g_hi)
    assert np.array_equal(sums, f["score_volume_sums"])
    cs = dl.CeresScanMatcher3D(ctx, DEFAULT_CSM)
    p2, summ = cs.Match(f["initial_pose"][:3], pose, [(f["points"], g_hi), (f["points"], g_lo)])
    dt, da = pose_distance(p2, f["csm_pose"])
    assert dt <= 1e-6 and da <= 1e-6
    assert summ["num_iterations"] == int(f["csm_iterations"])
    for g in (g_hi, g_lo):
        g.close()


@pytest.mark.gpu
def test_device_reproduces_voxel_filter_fixture(dl, ctx):
    f = load("voxel_filter.npz")
    cloud = dl.PointCloud(ctx, f["points"])
    out = cloud.voxel_filter(0.15)
    assert np.array_equal(out.download(), f["points"][f["kept_015"]])
    hi = cloud.adaptive_voxel_filter(2.0, 150, 15.0)
    lo = cloud.adaptive_voxel_filter(4.0, 200, 60.0)
    assert np.array_equal(hi.download(), f["adaptive_hi"]) and np.array_equal(lo.download(), f["adaptive_lo"])
    for c in (out, hi, lo, cloud):
        c.close()


def test_oracle_fair_volume_equals_reference_layout_loop(orc):
    """orc_rtcsm3d_range_fair_volume (every candidate's integer sum and reference score from one pass over a flat leaf
    table -- what tools/config5_full_parity.py runs at full size) against the reference-layout loop of the golden fixture:
    same sums, same scores to the bit, same first maximum."""
    f = load("matching.npz")
    g_hi = oracle_grid_from(orc, 0.1, f["hi_cell_xyz"], f["hi_cell_value"])
    flat = orc.FlatGridIndex(g_hi)
    sums, scores = orc.rtcsm3d_volume_fair(DEFAULT_RTCSM, f["initial_pose"], f["points"], flat, threads=3)
    assert np.array_equal(sums, f["score_volume_sums"])
    r = orc.rtcsm3d_match(DEFAULT_RTCSM, f["initial_pose"], f["points"], g_hi, want_scores=True)
    assert scores.tobytes() == r["scores"].tobytes()
    assert int(np.argmax(scores)) == r["best_index"] and scores[r["best_index"]] == np.float32(f["rtcsm_score"])
