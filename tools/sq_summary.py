#!/usr/bin/env python3
"""VALU / LDS utilisation per kernel from three rocprofv3 --pmc passes (isolated kernels: VENC_NO_OVERLAP=1).
usage: sq_summary.py <dir with SQ_INSTS_VALU/ SQ_BUSY_CYCLES/ SQ_LDS_BANK_CONFLICT/ subdirs> <label>
Collect with (on the GPU box, cd /tmp; export TMPDIR=/tmp):
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \\
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
    VENC_NO_OVERLAP=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${c%% *} -- python3 bench.py --workload W --steps 3 --warmup 0 --cpu-frames 0 --no-profile --no-check-pass
  done"""
import collections
import csv
import glob
import sys

d, label = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for sub in ("SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT"):
    f = glob.glob("%s/%s/*/*counter_collection.csv" % (d, sub))[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if k == "k_me_pool4":  # the search on four blocks per CU (steps below 200 000 macroblocks): reported as the search kernel
            k = "k_me_pool"
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
        if sub == "SQ_INSTS_VALU" and r["Counter_Name"] == "SQ_INSTS_VALU":
            dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            dur[k][1] += 1
print("# rocprofv3 --pmc (three separate passes, --kernel-trace only) of: VENC_NO_OVERLAP=1 python3 bench.py --steps 3 --warmup 0 --cpu-frames 0 --no-profile --no-check-pass")
print("# %s, whole units of the encoder's own item count per launch (30 at 1080p, 15 at 2160p), every kernel alone on the chip (one stream).  Counter values are sums over all shader engines, averaged per launch." % label)
print("# valu_util = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x duration x 2.4 GHz); lds_busy = SQ_LDS_IDX_ACTIVE / (256 CUs x duration x 2.4 GHz)")
names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_BUSY_CYCLES",
         "SQ_WAVE_CYCLES", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS"]
print("%-14s %8s %9s %8s  " % ("kernel", "avg_us", "valu_util", "lds_busy") + " ".join("%20s" % n for n in names))
VALU_ISSUE_CYCLES = 4.3  # measured cost of one wave-instruction on a SIMD with >= 2 waves (tools/valu_ubench.hip: 4.2-4.5 for all but four opcodes)
doc_k = {}
for k in ("k_me_pool", "k_me_fix", "k_inter_fused", "k_entropy_p", "k_pack_p", "k_scan", "k_pack", "k_intra_fwd", "k_intra_dc", "k_intra_rec", "k_cavlc"):
    if not dur[k][1]:
        continue
    us = dur[k][0] / dur[k][1]
    v = lambda c: acc[k][c][0] / max(1, acc[k][c][1])
    doc_k[k] = {"avg_us": round(us, 2), "valu_wave_instructions_per_launch": int(v("SQ_INSTS_VALU")),
                "valu_issue_frac": round(v("SQ_INSTS_VALU") * VALU_ISSUE_CYCLES / (1024 * us * 2400), 4),
                "lds_busy_frac": round(v("SQ_LDS_IDX_ACTIVE") / (256 * us * 2400), 4)}
    print("%-14s %8.1f %8.1f%% %7.1f%%  " % (k, us, 100 * v("SQ_INSTS_VALU") * 4 / (1024 * us * 2400), 100 * v("SQ_LDS_IDX_ACTIVE") / (256 * us * 2400)) +
          " ".join("%20.0f" % v(n) for n in names))

if len(sys.argv) > 3:  # sq_summary.py <dir> <label> <profiles/sq_counters.json>: keep the utilisation figures bench.py quotes
    import importlib
    import json
    import os
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    path = sys.argv[3]
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc["note"] = ("per launch, every kernel alone on the chip (VENC_NO_OVERLAP=1), the encoder's own item count per launch (30 at 1080p, 15 at 2160p); valu_issue_frac = SQ_INSTS_VALU x %.1f cycles "
                   "(measured issue cost of a wave-instruction, tools/valu_ubench.hip) / (1024 SIMDs x duration x 2.4 GHz)" % VALU_ISSUE_CYCLES)
    doc[label] = doc_k
    doc["source"] = {"kernel_sources_sha16": importlib.import_module("video-encoder_amd").kernel_sources_sha16(), "collected": time.strftime("%Y-%m-%d"),
                     "how": "tools/final_batch.sh -> tools/sq_summary.py"}
    json.dump(doc, open(path, "w"), indent=1)
