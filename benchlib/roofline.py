"""The `roofline` object of the bench line: counters measured NOW by child runs, nothing read from a committed file."""
import os
import sys
import time

import numpy as np

from benchlib import (ROOT, RTCSM_OPTS, CSM_OPTS, HIT_P, MISS_P, FREE, HIGH_RES_MAX_RANGE, HBM_PEAK_GBS, VALU_PEAK_LANE_OPS,
                      build_scene, insertion_targets)


def score_kernel_counters(args, kernel):
    """Hardware counters of the score kernel measured NOW: child runs of this script (--pmc-child: the same scene,
    three matches) under `rocprofv3 --pmc`, one per counter group (tools/pmc_live.py).  Nothing is read from a
    committed file; what cannot be measured is absent and printed as null."""
    if args.no_pmc:
        return {}, ["skipped (--no-pmc or N > 1)"]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_live
    child = [os.path.join(ROOT, "bench.py"), "--pmc-child", "--config", str(args.config), "--beams", str(args.beams),
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

