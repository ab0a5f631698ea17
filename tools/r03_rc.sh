#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_rc
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/test.log 2>&1 || { tail -30 $O/test.log; exit 1; }
tail -2 $O/test.log
for fl in "--crf 26" "--crf 26 --intra-modes 3 --subpel 2 --deblock" ""; do
    timeout -k 10 300 python bench.py --workload 1080p --no-extras --cpu-frames 0 --steps 10 $fl > $O/b.json 2> $O/b.err || { echo "$fl FAILED"; tail -5 $O/b.err; continue; }
    python3 - "$O/b.json" "$fl" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("%-50s %9.1f Mpx/s %7.3f ms/step  k_me %s us (iso %s)" % (sys.argv[2], d["value"], d["ms_per_step"], r.get("avg_launch_us"), r.get("isolated_avg_launch_us")))
PY
done
