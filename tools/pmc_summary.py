#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM / rocprofv3 section:
separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` runs; both counters are in KiB; on gfx950 FETCH_SIZE
counts the 128-byte requests of wide reads at 64 bytes, so it is doubled).
usage: pmc_summary.py <fetch_dir> <write_dir> <workload> [profiles/pmc_traffic.json]  -- merges into the JSON"""
import collections, csv, glob, json, os, sys


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not f:
        raise SystemExit("no counter_collection.csv under " + d)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0]
        if name.startswith("void "):
            name = name[5:]
        name = name.split("<")[0]
        if name == "k_me_pool4":  # the search on four blocks per CU (steps below 200 000 macroblocks): reported as the search kernel
            name = "k_me_pool"
        acc[name][0] += float(r["Counter_Value"])
        acc[name][1] += 1
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    if k in fetch:
        out[k + "_hbm_read_bytes_per_launch"] = int(2 * 1024 * fetch[k][0] / fetch[k][1])
    if k in write:
        out[k + "_hbm_write_bytes_per_launch"] = int(1024 * write[k][0] / write[k][1])
    out[k + "_launches"] = (fetch.get(k) or write.get(k))[1]
me = "k_me_pool"
if me + "_hbm_read_bytes_per_launch" in out:
    out["k_me_hbm_bytes_per_launch"] = out[me + "_hbm_read_bytes_per_launch"] + out.get(me + "_hbm_write_bytes_per_launch", 0)
path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
doc = json.load(open(path)) if os.path.exists(path) else {}
doc["note"] = ("HBM bytes per launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of bench.py "
               "--steps 3 --warmup 0 --no-profile --no-check-pass: whole units only, the encoder's own item count per launch (30 at 1080p, 15 at 2160p); "
               "tools/pmc_summary.py). FETCH_SIZE and WRITE_SIZE are in KiB; "
               "FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-B requests tallied at 64 B).")
doc[sys.argv[3]] = out
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, subprocess, time
try:
    head = subprocess.check_output(["git", "-C", os.path.dirname(path), "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
except Exception:
    head = None  # (the GPU box's copy has no .git: tools/collect_profiles.py fills it in when the summary is kept)
doc["source"] = {"kernel_sources_sha16": importlib.import_module("video-encoder_amd").kernel_sources_sha16(), "git_head": head,
                 "collected": time.strftime("%Y-%m-%d"), "how": "tools/final_batch.sh -> tools/pmc_summary.py"}
json.dump(doc, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
