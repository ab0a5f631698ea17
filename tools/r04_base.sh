#!/bin/bash
# round 4 baseline on today's box: headline at gop-batch 10 / 20 / 30, both sizes
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_base; mkdir -p $O; cd $R
for gb in 10 20 30; do
  for wl in 1080p 2160p; do
    timeout -k 10 200 python bench.py --workload $wl --no-extras --cpu-frames 0 --steps 10 --gop-batch $gb > $O/gb${gb}_$wl.json 2> $O/gb${gb}_$wl.err || { echo "gb $gb $wl FAILED"; tail -5 $O/gb${gb}_$wl.err; exit 1; }
    python3 - $O/gb${gb}_$wl.json $gb $wl <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {}); k = d.get("kernel_ms_per_step", {}); ki = d.get("kernel_ms_per_step_isolated", {})
print("gb %s %s %9.1f Mpx/s %7.3f ms/step k_me %6.1f us (iso %6.1f) timed: %s | iso: %s" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], r.get("avg_launch_us", 0), r.get("isolated_avg_launch_us", 0),
  " ".join("%s=%.2f" % (a, b) for a, b in k.items() if b), " ".join("%s=%.2f" % (a, b) for a, b in ki.items() if b)))
PY
  done
done
