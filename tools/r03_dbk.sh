#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_deblock.py -x -q > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest.log | head -20; exit 1; }
for t in "--deblock" "--crf 26 --intra-modes 2 --subpel 2 --deblock" "--deblock --gop-batch 30" "--crf 26 --intra-modes 2 --subpel 2 --deblock --gop-batch 30"; do
  for wl in 1080p ${DBK_WL2}; do
    timeout -k 10 250 python bench.py --workload $wl --no-extras --cpu-frames 30 --steps 6 $t > $O/dbk.json 2> $O/dbk.err || { echo "$t FAILED"; tail -3 $O/dbk.err; continue; }
    python3 - $O/dbk.json "$wl $t" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernel_ms_per_step", {}); ki = d.get("kernel_ms_per_step_isolated", {})
print("%-52s %9.1f Mpx/s %7.3f ms/step bit_exact %s | %s | iso %s" % (sys.argv[2], d["value"], d["ms_per_step"], d.get("bit_exact"), " ".join("%s=%.2f" % (a, b) for a, b in k.items() if b), " ".join("%s=%.2f" % (a, b) for a, b in ki.items() if b)))
PY
  done
done
