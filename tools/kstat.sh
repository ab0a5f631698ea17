#!/bin/bash
# per-kernel rocprofv3 stats of a short bench run: usage  bash tools/kstat.sh <tag> "<ENV=..>" <bench flags...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; V=$2; shift 2
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export $V
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-extras --cpu-frames 0 --steps 2 --warmup 1 --no-profile "$@" > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/prof/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print("%-60s calls %6s avg %10.1f us  total %8.2f ms  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
# KSTAT_KERNEL=name: the distribution of that kernel's durations (us) from the trace
if [ -n "$KSTAT_KERNEL" ]; then python3 - $O $KSTAT_KERNEL <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/prof/*/*kernel_trace.csv")[0]
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"])
n = len(d)
print(sys.argv[2], "n", n, "min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (d[0], d[n // 10], d[n // 2], d[9 * n // 10], d[-1]))
print(" ".join("%.0f" % x for x in d[:: max(1, n // 40)]))
PY
fi
