#!/bin/bash
# what a step of k_deblock_ls costs without its parts (VENC_DL_DIAG: 1 no filtering, 2 no strengths, 4 no loads; pictures then wrong)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_dbk3
mkdir -p $O
cd $R
for d in 0 1 2 3 7; do
    VENC_DL_DIAG=$d timeout -k 10 300 python bench.py --workload 1080p --no-extras --cpu-frames 0 --steps 3 --deblock --gop-batch 10 > $O/b$d.json 2> $O/b$d.err || { echo "diag $d FAILED"; tail -5 $O/b$d.err; }
    python3 - "$O/b$d.json" "VENC_DL_DIAG=$d" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ki = d.get("kernel_ms_per_step_isolated", {})
    print("%-20s %9.1f Mpx/s  k_deblock iso %.2f ms per pass = %.0f us per launch" % (sys.argv[2], d["value"], ki.get("k_deblock", 0), ki.get("k_deblock", 0) * 1000 / 59))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
done
