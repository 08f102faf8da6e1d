#!/usr/bin/env python3
"""BASELINE config 5's search (benchlib.config5.config5_scene: 128 x 2048 returns, 5 cm, C = 2 352 637) on whatever
library DLIOM_LIB names: score-kernel time per match, pairs per second, a few candidates checked against the oracle.
The A/B tool of round 6's wide-pass kernel (tools/box_sweep.sh)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "d-liom_amd")):
    sys.path.insert(0, p)
import benchlib  # noqa: E402
import dliom as dl  # noqa: E402
from dliom import synth  # noqa: E402
from benchlib.config5 import config5_scene, device_grid_to_oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--check", type=int, default=12, help="random candidates compared with the oracle (0: none)")
    a = ap.parse_args()
    ctx = dl.Context(0)
    ins, grids, sc, (res_hi, _) = config5_scene(dl, synth, ctx)
    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, benchlib.RTCSM_OPTS)
    rt.Match(sc["init"], sc["cloud"], grids[0])
    ctx.set_profiling(2)
    ctx.reset_profiling()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        score, pose = rt.Match(sc["init"], sc["cloud"], grids[0])
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / a.reps * 1e3
    k_ms, k_n = ctx.kernel_time(dl.KERNEL_RTCSM_SCORE)
    ctx.set_profiling(0)
    st = rt.last_stats()
    C, n = int(st.window.num_candidates), int(st.num_points)
    k = k_ms / max(k_n, 1)
    print("config5 match wall %.2f ms | score kernel %.2f ms (kernel id %d) | C=%d N=%d pairs/s=%.3e best=%d score=%.6f flags=%d" %
          (wall, k, int(st.score_kernel), C, n, float(C) * n / (k * 1e-3), int(st.best_index), score, int(rt.box_error())))
    if a.check > 0:
        from oracle import oracle as orc
        og = device_grid_to_oracle(orc, grids[0], res_hi)
        sums = rt.score_volume(sc["init"], sc["pts"], grids[0])
        idx = np.concatenate([np.random.RandomState(5).randint(0, C, size=a.check), [int(st.best_index)]])
        want, _ = orc.rtcsm3d_at(benchlib.RTCSM_OPTS, sc["init"], sc["pts"], og, idx, threads=min(16, os.cpu_count() or 1))
        ok = np.array_equal(sums[idx].astype(np.uint64), want)
        print("check %s: %d candidates' integer sums vs the oracle" % ("ok" if ok else "FAILED", len(idx)))
        if not ok:
            sys.exit(1)


if __name__ == "__main__":
    main()
