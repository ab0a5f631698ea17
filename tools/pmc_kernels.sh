#!/bin/bash
# SQ counters per launch for chosen kernels, every kernel alone on the chip: usage  bash tools/pmc_kernels.sh <tag> "<kernel name prefixes>" [bench flags]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; K=$2; shift 2
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM"; do
  VENC_NO_OVERLAP=1 timeout -k 10 250 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${c%% *} -- python3 $R/bench.py --workload 1080p --steps 1 --warmup 0 --cpu-frames 0 --no-profile --no-extras "$@" > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
done
python3 - $O "$K" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for f in glob.glob(sys.argv[1] + "/*/*/*kernel_trace.csv")[:1]:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; dur[k][1] += 1
for k in sys.argv[2].split():
    for kk in sorted(acc):
        if kk.startswith(k):
            print("%-14s %6.1f us x %3d | %s" % (kk, dur[kk][0] / max(1, dur[kk][1]), dur[kk][1], " ".join("%s=%.3g" % (c.replace("SQ_", ""), v[0] / v[1]) for c, v in sorted(acc[kk].items()))))
PY
