#!/bin/bash
# A/B of launcher variants on one box (same device, back to back): usage  bash tools/ab_bench.sh <tag> "<ENV=.. ENV=..>" ...
# AB_FLAGS="--crf 26" adds bench flags to every run.
# every remaining argument is one variant's environment; prints value / ms_per_step / k_me launch for 1080p and 2160p.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd $R
i=0
for v in "$@"; do
  for wl in 1080p 2160p; do
    env $v timeout -k 10 200 python bench.py --workload $wl --no-extras --cpu-frames 0 --steps 10 $AB_FLAGS > $O/ab_${i}_$wl.json 2> $O/ab_${i}_$wl.err || { echo "variant $i ($v) $wl FAILED"; tail -5 $O/ab_${i}_$wl.err; exit 1; }
    python3 - "$O/ab_${i}_$wl.json" "$v" $wl <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
k = d.get("kernel_ms_per_step", {})
print("%-40s %s  %9.1f Mpx/s  %7.3f ms/step  k_me %6.1f us (iso %6.1f)  kernels: %s" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"],
      r.get("avg_launch_us", 0), r.get("isolated_avg_launch_us", 0), " ".join("%s=%.2f" % (a, b) for a, b in k.items() if b)))
PY
  done
  i=$((i+1))
done
