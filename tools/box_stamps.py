#!/usr/bin/env python3
"""Experiments build only: per-workgroup (start, end, tickets, XCC) stamps of one box-kernel launch on the bench scene."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "d-liom_amd")):
    sys.path.insert(0, p)
import benchlib as bench  # noqa: E402  (constants + build_scene)
import dliom as dl  # noqa: E402
from dliom import synth  # noqa: E402


class A:
    beams, azimuths, high_resolution, low_resolution, map_scans, distinct_scans = 64, 1024, 0.10, 0.45, 20, 2


def main():
    lib = dl.load_library()
    ctx = dl.Context(0)
    ins, g_hi, g_lo, scans = bench.build_scene(A, dl, synth, ctx)
    rt = dl.RealTimeCorrelativeScanMatcher3D(ctx, bench.RTCSM_OPTS)
    for _ in range(3):
        rt.Match(scans[0]["init"], scans[0]["cloud"], g_hi)
    ctx.synchronize()
    n = 1024
    buf = np.zeros((n, 4), np.uint64)
    lib.dliom_exp_box_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    assert lib.dliom_exp_box_stamps(ctx.h, buf.ctypes.data, n) == 0
    b, e, t, x = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64), buf[:, 2], buf[:, 3]
    ok = e > 0
    t0 = b[ok].min()
    b, e = (b - t0) * 0.01, (e - t0) * 0.01  # us
    print("workgroups stamped %d; kernel span %.1f us" % (ok.sum(), e[ok].max()))
    print("start  us: p0 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(b[ok], [0, 50, 90, 99, 100])))
    print("end    us: p0 %.1f p1 %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(e[ok], [0, 1, 10, 50, 90, 100])))
    print("life   us: mean %.1f; resident fraction %.3f" % ((e - b)[ok].mean(), (e - b)[ok].sum() / (ok.sum() * e[ok].max())))
    print("tickets: min %d p50 %d max %d sum %d" % (t[ok].min(), np.median(t[ok]), t[ok].max(), t[ok].sum()))
    for xc in range(8):
        m = ok & ((x & 0xF) == xc)
        if m.any():
            print("  XCC %d: %d workgroups, end p50 %.1f max %.1f, tickets %d" % (xc, m.sum(), np.median(e[m]), e[m].max(), t[m].sum()))
    units = 6
    for u in range(units):
        m = ok & (np.arange(n) % units == u)
        print("  home unit %d: end p50 %.1f max %.1f tickets/wg %.1f" % (u, np.median(e[m]), e[m].max(), t[m].mean()))


if __name__ == "__main__":
    main()
