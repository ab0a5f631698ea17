#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r03_cus
for v in "VENC_ME_BLOCKS=3" "VENC_ME_BLOCKS=4" "VENC_ME_BLOCKS=4 VENC_ENTROPY_CUS=256" "VENC_ME_BLOCKS=4 VENC_ENTROPY_CUS=192" "VENC_ME_BLOCKS=4 VENC_ENTROPY_CUS=64" "VENC_ME_BLOCKS=3 VENC_ENTROPY_CUS=256"; do
  env $v timeout -k 10 200 python bench.py --workload 2160p --no-extras --cpu-frames 0 --steps 10 > gpurun_out/r03_cus/b.json 2> gpurun_out/r03_cus/b.err || { echo "$v FAILED"; tail -3 gpurun_out/r03_cus/b.err; continue; }
  python3 - gpurun_out/r03_cus/b.json "$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernel_ms_per_step", {})
print("%-45s %9.1f Mpx/s %7.3f ms/step  %s" % (sys.argv[2], d["value"], d["ms_per_step"], " ".join("%s=%.2f" % (a, b) for a, b in k.items() if b)))
PY
done
