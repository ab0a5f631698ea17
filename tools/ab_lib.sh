#!/bin/bash
# A/B of whole-library variants on one box: usage  bash tools/ab_lib.sh <tag> "<bench flags>" var_a.so var_b.so ...
# (files in video-encoder_amd/csrc, selected through VENC_HIP_LIB: the shipped libvenc_hip.so is never overwritten)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; F=$2; shift 2
mkdir -p $O; cd $R
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do
    VENC_HIP_LIB=$v timeout -k 10 200 python bench.py --no-extras --cpu-frames 0 --steps 10 $F > $O/${v%.so}_$rep.json 2> $O/${v%.so}_$rep.err || { echo "$v FAILED"; tail -3 $O/${v%.so}_$rep.err; exit 1; }
    python3 - "$O/${v%.so}_$rep.json" "$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernel_ms_per_step", {}); ki = d.get("kernel_ms_per_step_isolated", {}); r = d.get("roofline", {})
print("%-22s %9.1f Mpx/s %7.3f ms/step k_me %5.1f (iso %5.1f) | timed %s | iso %s" % (sys.argv[2], d["value"], d["ms_per_step"], r.get("avg_launch_us", 0), r.get("isolated_avg_launch_us", 0),
      " ".join("%s=%.2f" % (a.replace("k_", ""), b) for a, b in k.items() if b), " ".join("%s=%.2f" % (a.replace("k_", ""), b) for a, b in ki.items() if b)))
PY
  done
done
