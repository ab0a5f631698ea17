#!/bin/bash
# A/B of whole-library variants (compile-time parameters) on one box: usage  bash tools/ab_lib.sh <tag> "<bench flags>" var_a.so var_b.so ...
# (files in video-encoder_amd/csrc; each is copied over libvenc_hip.so for its runs, the original is restored at the end)
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/video-encoder_amd/csrc
O=$R/gpurun_out/$1; F=$2; shift 2
mkdir -p $O; cd $R
cp $C/libvenc_hip.so $O/orig.so
for rep in 1 2; do
  for v in "$@"; do
    cp $C/$v $C/libvenc_hip.so
    timeout -k 10 200 python bench.py --no-extras --cpu-frames 0 --steps 10 $F > $O/${v%.so}_$rep.json 2> $O/${v%.so}_$rep.err || { echo "$v FAILED"; tail -3 $O/${v%.so}_$rep.err; cp $O/orig.so $C/libvenc_hip.so; exit 1; }
    python3 - "$O/${v%.so}_$rep.json" "$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernel_ms_per_step", {})
print("%-16s %9.1f Mpx/s %7.3f ms/step  %s" % (sys.argv[2], d["value"], d["ms_per_step"], " ".join("%s=%.2f" % (a, b) for a, b in k.items() if b)))
PY
  done
done
cp $O/orig.so $C/libvenc_hip.so
