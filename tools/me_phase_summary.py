#!/usr/bin/env python3
"""Table of tools/me_phase_pmc.sh: per truncation point of k_me_pool the average launch time and SQ counters per launch,
and the differences between consecutive points = the cost of each phase.  usage: me_phase_summary.py <dir> <label>"""
import collections
import csv
import glob
import sys

d, label = sys.argv[1], sys.argv[2]
PH = {1: "P0 staging", 2: "P1 horizontal sums", 3: "P2 vertical sums", 4: "P3 start points", "F": "P4 bounds, survivors, vectors (search only, VENC_FUSE=0)",
      0: "P5 luma transform (whole kernel)"}
rows = {}
for stop in (1, 2, 3, 4, "F", 0):
    acc = collections.defaultdict(lambda: [0.0, 0])
    dur = [0.0, 0]
    for i in (0, 1):
        fs = glob.glob("%s/s%s_%d/*/*counter_collection.csv" % (d, stop, i))
        if not fs:
            continue
        seen = set()
        rows_ = [r for r in csv.DictReader(open(fs[0])) if "k_me_pool" in r["Kernel_Name"]]
        first = {min((int(r["Dispatch_Id"]) for r in rows_), default=-1)}
        first = {str(x) for x in first}
        for r in rows_:
            if r["Dispatch_Id"] in first:
                continue  # the first P picture of the GOPs (poor start points: twice the instructions) is left out of the steady-state budget
            a = acc[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
            if i == 0 and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                dur[1] += 1
    rows[stop] = (dur[0] / max(1, dur[1]), {k: v[0] / max(1, v[1]) for k, v in acc.items()}, dur[1])
names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
         "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS"]
print("# k_me_pool phase budget (tools/me_phase_pmc.sh), %s x 10 items per launch, kernel alone on the chip, steady-state P pictures (the GOPs' first P picture left out);" % label)
print("# P0-P3 by truncation (VENC_ME_STOP), the search as a whole = the kernel without its fused transform (VENC_FUSE=0); counters = sums over the chip per launch")
print("# (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_* count quad-cycles).  'delta' rows = this point minus the previous one = the phase's own cost.")
print("%-58s %8s " % ("up to and including", "avg_us") + " ".join("%14s" % n.replace("SQ_", "") for n in names))
prev = None
for stop in (1, 2, 3, 4, "F", 0):
    us, c, n = rows[stop]
    print("%-58s %8.1f " % (PH[stop], us) + " ".join("%14.0f" % c.get(k, 0) for k in names) + "   (%d launches)" % n)
    if prev is not None:
        pus, pc, _ = prev
        print("%-58s %8.1f " % ("   delta", us - pus) + " ".join("%14.0f" % (c.get(k, 0) - pc.get(k, 0)) for k in names))
    prev = rows[stop]
