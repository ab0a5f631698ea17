#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_cus2
mkdir -p $O
cd $R
for v in "VENC_ENTROPY_CUS=0" "VENC_ENTROPY_CUS=192" "VENC_ENTROPY_CUS=128"; do
  for fl in "--deblock" "--deblock --subpel 2 --crf 26 --intra-modes 3"; do
    env $v timeout -k 10 300 python bench.py --workload 1080p --no-extras --cpu-frames 0 --steps 9 $fl > $O/b.json 2> $O/b.err || { echo "$v $fl FAILED"; tail -5 $O/b.err; continue; }
    python3 - "$O/b.json" "$v $fl" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernel_ms_per_step", {})
print("%-75s %9.1f Mpx/s %7.3f ms/step  %s" % (sys.argv[2], d["value"], d["ms_per_step"], " ".join("%s=%.2f" % (a, b) for a, b in k.items() if b)))
PY
  done
done
