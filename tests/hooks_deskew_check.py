"""Body of tests/test_gpu_parity.py::test_deskew_check_paths_forced_by_the_hooks_build: runs with DLIOM_LIB pointing at
libdliom_hooks.so.  Knob 2 of dliom_ctx_set_tuning = 2 widens the de-skew's ambiguity bound so that EVERY hit is recorded
(the ring overflows: the records-only pass over all hits and the full read-back run); = 3 additionally treats every record
as different, so that every hit is redone by the fix kernel with the host's (identical) quaternion and the compaction
runs a second time.  Results must equal the oracle's bit for bit in both modes, for dliom_deskew and dliom_add_range_data."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "d-liom_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import dliom as dl  # noqa: E402
from dliom import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    assert os.environ.get("DLIOM_LIB", "").endswith("libdliom_hooks.so")
    dl.load_library()
    ctx = dl.Context(0)
    k = 5
    prev, cur = synth.trajectory_pose(0.1 * (k - 1)), synth.trajectory_pose(0.1 * k)
    pts, rel_t = synth.scan(cur, 32, 256)
    ranges = np.concatenate([pts, rel_t.reshape(-1, 1)], axis=1).astype(np.float32)
    vfs, min_r, max_r, T = 0.15, 1.0, 20.0, 0.1
    ref = orc.deskew_and_filter(T, min_r, max_r, vfs, prev, cur, ranges)
    hits = ranges[orc.voxel_filter(0.5 * np.float32(vfs), ranges[:, :3])]
    for mode in (2, 3):
        ctx.set_tuning(dl.TUNE_RESERVED_TEST_HOOK, mode)
        c0, o0, f0 = ctx.deskew_check_stats()
        xyz, kind, cur_f = dl.deskew(ctx, prev, cur, T, hits, (0, 0, 0), min_r, max_r)
        c1, o1, f1 = ctx.deskew_check_stats()
        assert c1 - c0 == len(hits) and o1 == o0 + 1, (mode, c1 - c0, len(hits), o1 - o0)  # every hit recorded: the overflow pass ran
        assert (f1 - f0 == len(hits)) if mode == 3 else (f1 == f0), (mode, f1 - f0)
        ret = kind == 1
        assert np.array_equal(xyz[ret].view(np.uint32), ref["hits_in_local"][ret].astype(np.float32).view(np.uint32)), mode
        assert np.array_equal(cur_f, ref["current_pose"].astype(np.float32)) and np.array_equal(kind, ref["kind"].astype(np.uint8)), mode
        # the device chain (records ride along with the compaction's read-back; mode 3 compacts twice)
        cloud, origin_d, cur_d = dl.add_range_data(ctx, prev, cur, T, ranges, (0, 0, 0), min_r, 100.0, vfs)
        ref2 = orc.deskew_and_filter(T, min_r, 100.0, vfs, prev, cur, ranges)
        assert np.array_equal(cloud.download().view(np.uint32), ref2["returns_in_tracking"].astype(np.float32).view(np.uint32)), mode
        assert np.array_equal(cur_d, ref2["current_pose"].astype(np.float32)), mode
        cloud.close()
    ctx.set_tuning(dl.TUNE_RESERVED_TEST_HOOK, 0)
    c0, o0, f0 = ctx.deskew_check_stats()
    dl.deskew(ctx, prev, cur, T, hits, (0, 0, 0), min_r, max_r)
    c1, o1, f1 = ctx.deskew_check_stats()
    assert o1 == o0 and f1 == f0 and c1 - c0 < len(hits) // 10  # the real bound: a handful of borderline hits, none different
    print("hooks_deskew_check ok (real bound: %d of %d hits re-examined on the host)" % (c1 - c0, len(hits)))
    ctx.close()


if __name__ == "__main__":
    main()
