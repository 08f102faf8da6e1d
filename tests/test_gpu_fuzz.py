"""A slice of the randomised differential testers of tools/ inside `pytest -m gpu`: fixed seeds,
both scene families, device vs CPU oracle bit for bit -- so that the driver's GPU test record, not a tool log, says so."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _run(module, argv, capsys):
    import importlib
    from dliom import synth
    mod = importlib.import_module("tools." + module)
    try:
        status = mod.main(argv)
    finally:
        synth.set_scene("cube")
        synth.set_trajectory()
    out = capsys.readouterr().out
    assert status == 0, out
    sys.stderr.write(out)  # the case counts, for the test log (each runs up to 300 cases or its time cap)
    return out


def test_fuzz_parity_slice(capsys):
    """Insertion, score volumes, RTCSM3D matches, voxel filters, adaptive filters, CeresScanMatcher3D on random grids,
    clouds, poses and options (tools/fuzz_parity.py)."""
    out = _run("fuzz_parity", ["--cases", "300", "--seed", "91000", "--seconds", "60"], capsys)
    assert "fuzz ok" in out


def test_fuzz_fast_csm_slice(capsys):
    """FastCorrelativeScanMatcher3D Match / MatchWith3DofInitial on random submaps (cube and yard scenes), pyramid depths,
    windows and thresholds (tools/fuzz_fast_csm.py)."""
    out = _run("fuzz_fast_csm", ["--cases", "300", "--seed", "92000", "--seconds", "45"], capsys)
    assert "fast csm fuzz ok" in out


def test_fuzz_round3_slice(capsys):
    """ComputeHistogram (cube and yard scans incl. slices above 4096 points, crops, duplicates, returns on common rays),
    std::sort's order of equal keys up to 30 000 keys, AddRangeData under random motion (tools/fuzz_round3.py)."""
    out = _run("fuzz_round3", ["--cases", "300", "--seed", "93000", "--seconds", "75"], capsys)
    assert "round-3 fuzz ok" in out


def test_fuzz_box_variants_slice(capsys):
    """The box score kernel's three instantiations (round 6) against the oracle's FULL candidate loop on random scenes:
    linear windows of 1 to 4 cells, random resolutions and initial orientations, 20 000 - 70 000 returns
    (tools/fuzz_box_variants.py): every integer sum, winner index, score bits, pose."""
    out = _run("fuzz_box_variants", ["--cases", "40", "--seed", "96000", "--seconds", "60"], capsys)
    assert "box variant fuzz ok" in out
