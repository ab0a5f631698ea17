#!/bin/bash
# same-box A/B of environment variants, REPS rounds in rotation (noise between leases and within one is +-5 % on a 40 ms timed region):
# usage  WL=1080p STEPS=40 REPS=3 bash tools/ab_env.sh <tag> "<ENV=.. ENV=..>" ...     prints every run and the per-variant median
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift; mkdir -p $O; cd $R
WL=${WL:-1080p}; STEPS=${STEPS:-40}; REPS=${REPS:-3}
for rep in $(seq 1 $REPS); do
  i=0
  for v in "$@"; do
    env $v timeout -k 10 200 python bench.py --workload $WL --no-extras --cpu-frames 0 --steps $STEPS $AB_FLAGS > $O/v${i}_r$rep.json 2> $O/v${i}_r$rep.err || { echo "variant $i ($v) FAILED"; tail -5 $O/v${i}_r$rep.err; exit 1; }
    i=$((i+1))
  done
done
python3 - $O $WL "$@" <<'PY'
import json, sys, glob, statistics
O, wl, vs = sys.argv[1], sys.argv[2], sys.argv[3:]
for i, v in enumerate(vs):
    vals, me = [], []
    for f in sorted(glob.glob("%s/v%d_r*.json" % (O, i))):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        vals.append(d["value"]); me.append(d.get("roofline", {}).get("avg_launch_us", 0))
    print("%-44s %s median %9.1f Mpx/s  runs %s  k_me %s us" % (v, wl, statistics.median(vals), " ".join("%.0f" % x for x in vals), " ".join("%.0f" % x for x in me)))
PY
