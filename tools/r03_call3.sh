#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_c3; mkdir -p $O
cd $R
show() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {}); k = d.get("kernel_ms_per_step", {})
print("%-34s %9.1f Mpx/s %7.3f ms/step k_me %6.1f us (iso %6.1f) per_pass %s kernels: %s" % (sys.argv[2], d["value"], d["ms_per_step"], r.get("avg_launch_us", 0), r.get("isolated_avg_launch_us", 0), {k_: v for k_, v in (d.get("per_pass") or {}).items() if k_ in ("min", "median", "max")}, " ".join("%s=%.2f" % (a, b) for a, b in k.items() if b)))
PY
}
for gb in 10 20 30; do
  for wl in 1080p 2160p; do
    timeout -k 10 200 python bench.py --workload $wl --no-extras --cpu-frames 0 --steps 12 --gop-batch $gb > $O/gb${gb}_$wl.json 2> $O/gb${gb}_$wl.err || { tail -3 $O/gb${gb}_$wl.err; exit 1; }
    show $O/gb${gb}_$wl.json "gop-batch $gb $wl"
  done
done
# 2-rank rehearsal of the self-launcher on the one GPU (gloo, both ranks on device 0): streams leg + slices leg
timeout -k 10 400 python bench.py --gpus 2 --device 0 --backend gloo --steps 3 --warmup 1 --cpu-frames 0 --no-extras --slices-steps 2 > $O/launch2.json 2> $O/launch2.err; echo "self-launch rc=$?"; cut -c1-600 $O/launch2.json; python3 -c "
import json,sys
d=json.loads(open('$O/launch2.json').read().strip().splitlines()[-1]); print('n_gpus',d['n_gpus'],'value',d['value'],'slices',json.dumps(d.get('slices_2160p'))[:900])"
tail -3 $O/launch2.err
