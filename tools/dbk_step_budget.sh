#!/bin/bash
# Step budget of k_deblock_ls (venc_deblock2.hip): isolated launch durations with parts of a step left out (VENC_DL_DIAG: 1 no
# filtering, 2 no strengths (and so no filtering), 4 no loads: the pictures are then wrong, only the times mean something), and
# the strip kernel (VENC_DBK=strip) beside it.  usage (GPU box, repo root): bash tools/dbk_step_budget.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
echo "# 1080p, ten items per launch (80 blocks), every kernel alone on the chip; k_deblock* launches of one pass over the clip (29 luma, 29 chroma, 1 intra)"
for v in "VENC_DBK=strip" "VENC_DL_DIAG=0" "VENC_DL_DIAG=1" "VENC_DL_DIAG=2" "VENC_DL_DIAG=7"; do
  env $v VENC_NO_OVERLAP=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/${v#*=}_${v%%=*} -- python3 $R/bench.py --workload 1080p --steps 1 --warmup 0 --cpu-frames 0 --no-profile --no-extras --deblock --gop-batch 10 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
  python3 - $O/${v#*=}_${v%%=*} "$v" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_deblock" in r["Kernel_Name"]]
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows)
what = {"VENC_DBK=strip": "strip kernel (one wave per row, flags, hand-over through memory)", "VENC_DL_DIAG=0": "k_deblock_ls",
        "VENC_DL_DIAG=1": "k_deblock_ls without the filter phases", "VENC_DL_DIAG=2": "k_deblock_ls without strengths (hence without phases)",
        "VENC_DL_DIAG=7": "k_deblock_ls without loads, strengths, phases: barrier, tile writes, stores"}[sys.argv[2]]
print("%-16s launches %3d  median %6.1f us  min %6.1f  max %6.1f   %s" % (sys.argv[2], len(d), d[len(d) // 2], d[0], d[-1], what))
PY
done
