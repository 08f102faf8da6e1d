#!/usr/bin/env python3
"""Hardware counters of the score kernel, measured in the run that prints them.

rocprofv3 cannot attach to a running process, so bench.py starts a CHILD of itself under `rocprofv3 --pmc` (never
combined with a trace domain) that builds the same scene and launches the score kernel a few times; one child per
counter group (FETCH_SIZE and WRITE_SIZE do not share a pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").  Per-dispatch
means of the dispatches whose kernel name matches are returned.  Nothing here reads a committed file: if rocprofv3 is
missing or a pass fails the counters are simply absent and bench.py prints null for what depends on them.
"""
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PASSES = (
    ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY",
     "SQ_WAIT_ANY"),
    ("FETCH_SIZE",),
    ("WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"),
    ("GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"),
)


def measure(child_args, kernel_regex, passes=PASSES, timeout=150, keep_dir=None):
    """-> ({counter: {"mean": per-dispatch mean, "launches": n}}, [problem strings])."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return {}, ["rocprofv3 not found"]
    out, problems = {}, []
    base = keep_dir or tempfile.mkdtemp(prefix="dliom_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for i, counters in enumerate(passes):
        d = os.path.join(base, "pass%d" % i)
        cmd = [rocprof, "--pmc"] + list(counters) + ["--kernel-include-regex", kernel_regex, "--output-format", "csv",
                                                     "-d", d, "-o", "p", "--", sys.executable] + list(child_args)
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
        except subprocess.TimeoutExpired:
            problems.append("pass %d (%s): timeout" % (i, " ".join(counters)))
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            problems.append("pass %d (%s): rc %d %s" % (i, " ".join(counters), r.returncode, (r.stderr or "")[-300:]))
            continue
        acc = {}
        for f in files:
            for row in csv.DictReader(open(f)):
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for k, v in acc.items():
            out[k] = {"mean": sum(v) / len(v), "launches": len(v)}
    if keep_dir is None:
        shutil.rmtree(base, ignore_errors=True)
    return out, problems


def derive(counters, pairs, launch_seconds):
    """Figures bench.py prints, from the raw counters (all per launch)."""
    d = {}
    c = {k: v["mean"] for k, v in counters.items()}
    wave_pairs = pairs / 64.0
    if "SQ_INSTS_VALU" in c:
        d["valu_instructions_per_pair"] = c["SQ_INSTS_VALU"] / wave_pairs
    if "SQ_INSTS_LDS" in c:
        d["lds_instructions_per_pair"] = c["SQ_INSTS_LDS"] / wave_pairs
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # KiB; gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section)
        d["traffic_bytes"] = 2.0 * c["FETCH_SIZE"] * 1024.0 + c["WRITE_SIZE"] * 1024.0
        d["fetch_bytes_raw"] = c["FETCH_SIZE"] * 1024.0
        d["write_bytes_raw"] = c["WRITE_SIZE"] * 1024.0
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "SQ_WAVE_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
        # SQ_* cycle counters are in quad-cycles; GRBM_GUI_ACTIVE sums the 8 XCDs; 1024 SIMDs
        d["resident_waves_per_simd"] = 4.0 * c["SQ_WAVE_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        d["lds_bank_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if "SQ_ACTIVE_INST_VALU" in c and c.get("GRBM_GUI_ACTIVE", 0) > 0:
        # quad-cycles in which a SIMD's vector ALU executes an instruction, over the launch's SIMD-cycles: how busy the
        # pipe that bounds the kernel is (every VALU instruction counted at 4 cycles -- most of this kernel's are)
        d["valu_busy_frac"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
    if "SQ_ACTIVE_INST_ANY" in c and c.get("SQ_WAVE_CYCLES", 0) > 0:
        d["wave_issue_frac"] = c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    return d
