#!/bin/bash
# usage: bash tools/r03_ab.sh <tag> [pytest-args...]   -- GPU tests (optional selection) then headline bench at both sizes
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift; mkdir -p $O
cd $R
if [ "$1" != "notest" ]; then
  timeout -k 10 600 python -m pytest tests -m gpu -x -q "$@" > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log
  [ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest.log | head -20; exit 1; }
fi
for wl in 1080p 2160p; do
  timeout -k 10 200 python bench.py --workload $wl --no-extras --cpu-frames 0 --steps 12 $BENCH_FLAGS > $O/b_$wl.json 2> $O/b_$wl.err || { tail -3 $O/b_$wl.err; exit 1; }
  python3 - $O/b_$wl.json $wl <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {}); k = d.get("kernel_ms_per_step", {}); ki = d.get("kernel_ms_per_step_isolated", {})
print("%-6s %9.1f Mpx/s %7.3f ms/step k_me %6.1f us (iso %6.1f) median %s | %s | iso: %s" % (sys.argv[2], d["value"], d["ms_per_step"], r.get("avg_launch_us", 0), r.get("isolated_avg_launch_us", 0), (d.get("per_pass") or {}).get("median"), " ".join("%s=%.2f" % (a, b) for a, b in k.items() if b), " ".join("%s=%.2f" % (a, b) for a, b in ki.items() if b)))
PY
done
