#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into the
small tracked files under profiles/: kernel stats CSV, the PMC summary of the score kernel (read by
bench.py for roofline.valu_instructions_per_pair / roofline.traffic, labelled there with this file as
their source), bench / W-ref lines."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "%s_rocprofv3_kernel_stats.csv" % tag))
bj = os.path.join(src, "bench_under_trace.json")
if os.path.exists(bj) and os.path.getsize(bj) > 0:
    shutil.copy(bj, os.path.join(dst, "%s_bench_under_kernel_trace.json" % tag))

counters, kernel = {}, None
for f in sorted(glob.glob(os.path.join(src, "pmc[0-9]", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        if "rtcsm_score" not in row["Kernel_Name"]:
            continue
        kernel = "rtcsm_score_box_kernel" if "score_box" in row["Kernel_Name"] else "rtcsm_score_dense_kernel"
        counters.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
summary = {k: {"launches": len(v), "mean": sum(v) / len(v)} for k, v in counters.items()}
meta = {}
if os.path.exists(bj) and os.path.getsize(bj) > 0:
    b = json.load(open(bj))
    meta = {"num_points": b["config"]["N_hi"], "num_candidates": b["config"]["C"]}
out = {"kernel": kernel, "workload": meta, "counters": summary,
       "notes": "rocprofv3 --pmc, one pass per counter group (never combined with tracing), per-dispatch means over the "
                "launches of `bench.py --steps 5 --warmup 2`.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE "
                "under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section): hbm_bytes_per_launch = "
                "2 x FETCH_SIZE + WRITE_SIZE is therefore an UPPER estimate for this kernel."}
if meta and "SQ_INSTS_VALU" in summary:
    wave_pairs = meta["num_points"] * meta["num_candidates"] / 64.0
    out["valu_instructions_per_wave_pair"] = summary["SQ_INSTS_VALU"]["mean"] / wave_pairs
    for k in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_ANY"):
        if k in summary:
            out[k.lower() + "_per_wave_pair"] = summary[k]["mean"] / wave_pairs
if "FETCH_SIZE" in summary:
    out["fetch_bytes_raw"] = summary["FETCH_SIZE"]["mean"] * 1024.0
    out["write_bytes_raw"] = summary.get("WRITE_SIZE", {"mean": 0.0})["mean"] * 1024.0
    out["hbm_bytes_per_launch"] = 2.0 * out["fetch_bytes_raw"] + out["write_bytes_raw"]
if summary:  # (a run of selected parts without the PMC passes keeps the file there is)
    json.dump(out, open(os.path.join(dst, "%s_pmc_score_kernel.json" % tag), "w"), indent=1)
# config 5's score kernel (tools/c5bench.py under rocprofv3 --pmc: the wide instantiation, round 6)
c5 = {}
for f in sorted(glob.glob(os.path.join(src, "pmc5_*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        if "rtcsm_score_box" in row["Kernel_Name"]:
            c5.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            c5_kernel = row["Kernel_Name"].split("(")[0]
if c5:
    import re
    s5 = {k: {"launches": len(v), "mean": sum(v) / len(v)} for k, v in c5.items()}
    o5 = {"kernel": c5_kernel, "counters": s5, "notes": "rocprofv3 --pmc over tools/c5bench.py (BASELINE config 5's search), per-dispatch means"}
    for lg in sorted(glob.glob(os.path.join(src, "pmc5_*.log"))):
        m = re.search(r"score kernel ([\d.]+) ms .* C=(\d+) N=(\d+)", open(lg).read())
        if m:
            o5["workload"] = {"num_candidates": int(m.group(2)), "num_points": int(m.group(3)), "score_kernel_ms_under_pmc": float(m.group(1))}
            break
    if "workload" in o5 and "SQ_INSTS_VALU" in s5:
        wp = o5["workload"]["num_points"] * o5["workload"]["num_candidates"] / 64.0
        o5["valu_instructions_per_wave_pair"] = s5["SQ_INSTS_VALU"]["mean"] / wp
    if "SQ_ACTIVE_INST_VALU" in s5 and "SQ_BUSY_CYCLES" in s5:
        o5["sq_active_inst_valu_over_busy_cycles"] = s5["SQ_ACTIVE_INST_VALU"]["mean"] / s5["SQ_BUSY_CYCLES"]["mean"]
    if "SQ_LDS_BANK_CONFLICT" in s5 and "SQ_LDS_IDX_ACTIVE" in s5:
        o5["lds_bank_conflict_frac"] = s5["SQ_LDS_BANK_CONFLICT"]["mean"] / s5["SQ_LDS_IDX_ACTIVE"]["mean"]
    if "FETCH_SIZE" in s5:
        o5["hbm_bytes_per_launch"] = 2.0 * s5["FETCH_SIZE"]["mean"] * 1024.0 + s5.get("WRITE_SIZE", {"mean": 0.0})["mean"] * 1024.0
    json.dump(o5, open(os.path.join(dst, "%s_pmc_score_kernel_config5.json" % tag), "w"), indent=1)
for name, o in (("wref_cpp.json", "%s_wref_cpp.json"), ("imu_window_cost.json", "%s_imu_window_cost.json")):
    f = os.path.join(src, name)
    if os.path.exists(f) and os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(dst, o % tag))
for name, o in (("bench_full.json", "%s_bench.json"), ("wref_full.json", "%s_wref_full.json"), ("wref.json", "%s_wref.json"),
                ("wref_stages.json", "%s_wref_stages.json"), ("stream.json", "%s_stream_config3.json"),
                ("stream_gentle.json", "%s_stream_config3_gentle.json"), ("config5_bench.json", "%s_config5_bench.json"),
                ("bench_gloo2.json", "%s_bench_gloo2.json"), ("fast_csm.json", "%s_fast_csm.json"),
                ("fast_csm_full.json", "%s_fast_csm_full.json"), ("fast_csm_dense.json", "%s_fast_csm_dense.json"),
                ("hist_bench.json", "%s_hist_bench.json"), ("bench_rccl_1rank.json", "%s_bench_rccl_1rank.json"),
                ("mirror_window_stream.json", "%s_mirror_window_stream.json"), ("voxel_filter.json", "%s_voxel_filter.json")):
    f = os.path.join(src, name)
    if os.path.exists(f) and os.path.getsize(f) > 0:
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(dst, o % tag), "w").write(lines[-1] + "\n")
gl = os.path.join(src, "graph_latency.txt")
if os.path.exists(gl) and os.path.getsize(gl) > 0:
    shutil.copy(gl, os.path.join(dst, "%s_graph_latency.txt" % tag))
h = os.path.join(src, "histogram.txt")
if os.path.exists(h):
    keep = [l for l in open(h).read().splitlines() if "histogram" in l or "rothist" in l or l.startswith('"Name"') or l.startswith("==")
            or l.startswith("{") or "rocprim" in l or "fill_multi" in l or "gather_to_pinned" in l]
    open(os.path.join(dst, "%s_histogram_kernels.txt" % tag), "w").write("\n".join(keep) + "\n")
for name, o in (("wref_trace", "%s_wref_full_rocprofv3_kernel_stats.csv"), ("wref_trace_yard", "%s_wref_full_yard_rocprofv3_kernel_stats.csv"),
                ("voxel_trace", "%s_voxel_filter_kernel_stats.csv")):
    for f in glob.glob(os.path.join(src, name, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, o % tag))
g = os.path.join(src, "gputest.log")
if os.path.exists(g):
    lines = [l for l in open(g).read().splitlines() if l.strip()]
    keep = [l for l in lines if "passed" in l or "failed" in l or "fuzz ok" in l or "windowed mirror" in l or "manifold vs tangent" in l]
    open(os.path.join(dst, "%s_gputest_summary.txt" % tag), "w").write("\n".join(keep) + "\n")
print(sorted(os.listdir(dst)))
