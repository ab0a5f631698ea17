#!/bin/bash
# round 3: the lock-step in-loop filter (k_deblock_ls) against the strip kernel: parity tests, then same-box A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_dbk2
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_deblock.py -m gpu -x -q > $O/test.log 2>&1 || { tail -40 $O/test.log; exit 1; }
tail -3 $O/test.log
for v in "VENC_DBK=strip" "VENC_DBK=ls"; do
  for fl in "--deblock" "--deblock --gop-batch 10" "--deblock --subpel 2 --crf 26 --intra-modes 3"; do
    env $v timeout -k 10 300 python bench.py --workload 1080p --no-extras --cpu-frames 0 --steps 6 $fl > $O/b.json 2> $O/b.err || { echo "$v $fl FAILED"; tail -5 $O/b.err; exit 1; }
    python3 - "$O/b.json" "$v $fl" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernel_ms_per_step", {})
ki = d.get("kernel_ms_per_step_isolated", {})
print("%-70s %9.1f Mpx/s %7.3f ms/step gop_batch %s  k_deblock %.2f (iso %.2f) sum %.2f" % (sys.argv[2], d["value"], d["ms_per_step"], d["config"].get("gop_batch"), k.get("k_deblock", 0), ki.get("k_deblock", 0), k.get("sum", 0)))
PY
  done
done
