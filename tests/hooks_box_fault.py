"""Body of tests/test_gpu_full_size.py::test_config2_dense_rerun_after_box_fault: runs with DLIOM_LIB pointing at
libdliom_hooks.so (a build of the library with -DDLIOM_TEST_HOOKS, `make -C d-liom_amd hooks`), the only build in which
knob 2 of dliom_ctx_set_tuning injects the box kernel's inconsistency word.  The rerun on the dense kernel must give the
oracle's winner, report the dense kernel, and leave the next match on the box kernel."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "d-liom_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import dliom as dl  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from helpers import DEFAULT_RTCSM, build_device_scene, device_grid_to_oracle  # noqa: E402

THREADS = min(8, os.cpu_count() or 1)


def main():
    assert os.environ.get("DLIOM_LIB", "").endswith("libdliom_hooks.so"), "run with DLIOM_LIB=<...>/libdliom_hooks.so"
    dl.load_library()
    assert dl.device_count() > 0
    ctx = dl.Context(0)
    ins, g_hi, g_lo, scans = build_device_scene(dl, ctx, 64, 1024, 0.10, 0.45, map_scans=20)
    sc = scans[0]
    og_hi = device_grid_to_oracle(orc, g_hi)
    ref = orc.rtcsm3d_match_parallel(DEFAULT_RTCSM, sc["init"], sc["pts"], og_hi, threads=THREADS)
    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, DEFAULT_RTCSM)
    HOOK = dl.TUNE_RESERVED_TEST_HOOK
    ctx.set_tuning(HOOK, 1)
    score, pose = rt.Match(sc["init"], sc["cloud"], g_hi)
    st = rt.last_stats()
    assert ctx.get_tuning(HOOK) == 0  # consumed
    assert st.score_kernel == 2, st.score_kernel          # the rerun ran on the dense-mirror kernel
    assert st.box_kernel_status == dl.BOX_REFUSED_FLAGGED  # ... and says why
    assert st.best_index == ref["best_index"]
    assert np.float32(score).tobytes() == np.float32(ref["score"]).tobytes()
    assert np.array_equal(pose, ref["pose"])
    # the score-volume entry point takes the same path
    ctx.set_tuning(HOOK, 1)
    got = rt.score_volume(sc["init"], sc["pts"], g_hi)
    idx = np.random.RandomState(3).randint(0, len(got), size=64)
    want, _ = orc.rtcsm3d_at(DEFAULT_RTCSM, sc["init"], sc["pts"], og_hi, idx, threads=THREADS)
    assert np.array_equal(got[idx].astype(np.uint64), want)
    # and the sharded phases (begin reads the word, finish does not)
    ctx.set_tuning(HOOK, 1)
    sh = dl.RtcsmShard(ctx, DEFAULT_RTCSM, 0, 1)
    score2, pose2 = sh.decode(sh.finish(sh.begin(sc["init"], sc["cloud"], g_hi)))
    assert rt.last_stats().score_kernel == 2
    assert np.float32(score2).tobytes() == np.float32(ref["score"]).tobytes() and np.array_equal(pose2, ref["pose"])
    ctx.set_tuning(HOOK, 1)
    score2, pose2 = sh.match(sc["init"], sc["cloud"], g_hi, lambda v: v)
    assert rt.last_stats().score_kernel == 2
    assert np.float32(score2).tobytes() == np.float32(ref["score"]).tobytes() and np.array_equal(pose2, ref["pose"])
    ctx.set_tuning(HOOK, 0)
    score3, _ = rt.Match(sc["init"], sc["cloud"], g_hi)
    assert rt.last_stats().score_kernel == 3 and np.float32(score3).tobytes() == np.float32(ref["score"]).tobytes()
    assert rt.last_stats().box_kernel_status == dl.BOX_RAN
    assert rt.box_error() == 0
    print("hooks_box_fault ok")


if __name__ == "__main__":
    main()
