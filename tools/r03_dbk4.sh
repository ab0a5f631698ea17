#!/bin/bash
# k_deblock_ls with more GOPs in lock-step (two small blocks per CU run at the latency of one)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_dbk4
mkdir -p $O
cd $R
for fl in "--deblock --gop-batch 30" "--deblock --gop-batch 60" "--deblock --gop-batch 40" "--deblock --subpel 2 --crf 26 --intra-modes 3 --gop-batch 30" "--deblock --subpel 2 --crf 26 --intra-modes 3 --gop-batch 60"; do
    timeout -k 10 300 python bench.py --workload 1080p --no-extras --cpu-frames 0 --steps 6 $fl > $O/b.json 2> $O/b.err || { echo "$fl FAILED"; tail -5 $O/b.err; continue; }
    python3 - "$O/b.json" "$fl" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernel_ms_per_step", {})
print("%-70s %9.1f Mpx/s %7.3f ms/step  kernels %s" % (sys.argv[2], d["value"], d["ms_per_step"], " ".join("%s=%.2f" % (a, b) for a, b in k.items() if b)))
PY
done
