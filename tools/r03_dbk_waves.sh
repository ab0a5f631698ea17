#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for w in 1 2 3 5 9; do
  VENC_DBK_WAVES=$w VENC_NO_OVERLAP=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/w$w -- python3 $R/bench.py --workload 1080p --steps 1 --warmup 0 --cpu-frames 0 --no-profile --no-extras --deblock > $O/w$w.log 2>&1 || { tail -5 $O/w$w.log; exit 1; }
  python3 - $O/w$w $w <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_deblock" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
# launches alternate: per P step luma (main chain) then chroma; print in order to see the two populations
print("waves", sys.argv[2], "n", len(d), "first 12:", " ".join("%.0f" % x for x in d[:12]), "| sorted median %.0f" % sorted(d)[len(d) // 2])
PY
done
