#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into the
small tracked files under profiles/: kernel stats CSV, PMC summary JSON and pmc_traffic.json
(HBM bytes per score-kernel launch, read by bench.py for roofline.traffic)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "%s_rocprofv3_kernel_stats.csv" % tag))
bj = os.path.join(src, "bench_under_trace.json")
if os.path.exists(bj) and os.path.getsize(bj) > 0:
    shutil.copy(bj, os.path.join(dst, "%s_bench_under_kernel_trace.json" % tag))

counters = {}
for f in sorted(glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        if "rtcsm_score" not in row["Kernel_Name"]:
            continue
        counters.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
summary = {k: {"launches": len(v), "mean": sum(v) / len(v)} for k, v in counters.items()}
meta = {}
if os.path.exists(bj) and os.path.getsize(bj) > 0:
    b = json.load(open(bj))
    meta = {"num_points": b["config"]["N_hi"], "num_candidates": b["config"]["C"],
            "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"]}
json.dump({"kernel": "rtcsm_score_dense_kernel", "workload": meta, "counters": summary,
           "notes": "rocprofv3 --pmc, one pass per counter group, values are per-dispatch means. FETCH_SIZE / "
                    "WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced streams by 2x "
                    "(MI355X_MICROARCH.md HBM section): the x2 correction is applied in pmc_traffic.json "
                    "as an UPPER estimate for this gather-dominated kernel."},
          open(os.path.join(dst, "%s_pmc_score_kernel.json" % tag), "w"), indent=1)
if "FETCH_SIZE" in summary and meta:
    fetch = summary["FETCH_SIZE"]["mean"] * 1024.0
    write = summary.get("WRITE_SIZE", {"mean": 0.0})["mean"] * 1024.0
    json.dump({"num_points": meta["num_points"], "num_candidates": meta["num_candidates"],
               "fetch_bytes_raw": fetch, "write_bytes_raw": write,
               "hbm_bytes_per_launch": 2.0 * fetch + write,
               "source": "profiles/%s_pmc_score_kernel.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, x2 gfx950 "
                         "read correction)" % tag},
              open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
for name, out in (("bench_full.json", "%s_bench.json"), ("wref.json", "%s_wref.json"),
                  ("wref_stages.json", "%s_wref_stages.json")):
    f = os.path.join(src, name)
    if os.path.exists(f) and os.path.getsize(f) > 0:
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(dst, out % tag), "w").write(lines[-1] + "\n")
print(sorted(os.listdir(dst)))
