#!/usr/bin/env python3
"""Copies the judged summaries of a tools/final_batch.sh run from gpurun_out/<dir> into profiles/ under a tag:
bench JSON lines, rocprofv3 kernel-stats CSVs, the bench line printed under rocprofv3, and the PMC counter
collections slimmed to (dispatch, kernel, grid, counter, KiB, start, end).
usage: python tools/collect_profiles.py gpurun_out/final4 r01_i"""
import csv
import glob
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
names = {"bench_default": "bench_default_1080p_with_2160p_pcie_cpu_bitexact", "bench_1080p": "bench_1080p", "bench_2160p": "bench_2160p",
         "bench_1080p_serial": "bench_1080p_serial_steps",
         "bench_2160p_serial": "bench_2160p_serial_steps", "bench_slices1": "bench_2160p_slices_1rank",
         "bench_slices2": "bench_2160p_slices_2rank_gloo_rehearsal", "bench_streams2": "bench_1080p_2rank_gloo_rehearsal_self_launched_with_slices_leg",
         "bench_streams2_torchrun": "bench_1080p_2rank_gloo_rehearsal_torchrun_with_slices_leg"}
for p in glob.glob(os.path.join(src, "bench_tool_*.json")):
    names[os.path.basename(p)[:-5]] = os.path.basename(p)[:-5]
ks = glob.glob(os.path.join(src, "stats_tools", "*", "*kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(dst, "%s_kernel_stats_1080p_all_tools_isolated.csv" % tag))
for a, b in names.items():
    p = os.path.join(src, a + ".json")
    if os.path.exists(p):
        line = [l for l in open(p).read().splitlines() if l.startswith('{"metric"')][-1]
        open(os.path.join(dst, "%s_%s.json" % (tag, b)), "w").write(line + "\n")
for wl in ("1080p", "2160p"):
    ks = glob.glob(os.path.join(src, "stats_" + wl, "*", "*kernel_stats.csv"))
    if ks:
        shutil.copy(ks[0], os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, wl)))
    log = os.path.join(src, "stats_%s.log" % wl)
    if os.path.exists(log):
        lines = [l for l in open(log).read().splitlines() if l.startswith('{"metric"')]
        if lines:
            open(os.path.join(dst, "%s_bench_%s_under_rocprofv3.json" % (tag, wl)), "w").write(lines[-1] + "\n")
    for kind, cname in (("pmcf", "fetch_size"), ("pmcw", "write_size")):
        cc = glob.glob(os.path.join(src, "%s_%s" % (kind, wl), "*", "*counter_collection.csv"))
        if not cc:
            continue
        with open(os.path.join(dst, "%s_pmc_%s_%s.csv" % (tag, cname, wl)), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value_KiB", "Start_Timestamp", "End_Timestamp"])
            for r in csv.DictReader(open(cc[0])):
                w.writerow([r["Dispatch_Id"], r["Kernel_Name"].split("(")[0], r["Grid_Size"], r["Counter_Name"], r["Counter_Value"],
                            r["Start_Timestamp"], r["End_Timestamp"]])
for wl in ("1080p", "2160p"):
    p = os.path.join(src, "sq_counters_%s_isolated.txt" % wl)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, "%s_sq_counters_%s_isolated.txt" % (tag, wl)))
for a, b in (("me_phase_1080p.txt", "me_phase_1080p.txt"), ("valu_ubench.txt", "valu_ubench.txt"), ("h2d_ubench.txt", "h2d_ubench.txt"),
             ("deblock_step_budget.txt", "deblock_step_budget.txt"), ("deblock_sq_counters.txt", "deblock_sq_counters.txt"),
             ("ifetch_ubench.txt", "ifetch_ubench.txt"), ("ab_search_variants.txt", "ab_search_variants.txt"), ("ab_deblock_kernels.txt", "ab_deblock_kernels.txt"),
             ("timeline_1080p.txt", "timeline_1080p.txt"), ("timeline_2160p.txt", "timeline_2160p.txt"), ("ab_launcher_1080p.txt", "ab_launcher_1080p.txt"),
             ("ab_launcher_2160p.txt", "ab_launcher_2160p.txt"), ("ingest_split.txt", "ingest_split.txt"), ("entropy_block_lifetimes_1080p.txt", "entropy_block_lifetimes_1080p.txt")):
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, b)))
# the kept counter files: taken from the batch, stamped with the commit they were collected at (the GPU box's copy has no .git)
import json
import subprocess
try:
    head = subprocess.check_output(["git", "-C", dst, "rev-parse", "--short", "HEAD"]).decode().strip()
except Exception:
    head = None
for name in ("pmc_traffic.json", "sq_counters.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        doc = json.load(open(p))
        if isinstance(doc.get("source"), dict) and not doc["source"].get("git_head"):
            doc["source"]["git_head"] = head
        json.dump(doc, open(os.path.join(dst, name), "w"), indent=1)
print("\n".join(sorted(f for f in os.listdir(dst) if f.startswith(tag))))
