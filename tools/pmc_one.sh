#!/bin/bash
# one rocprofv3 --pmc pass per library variant, kernels isolated; prints per-launch averages for k_me_pool / k_me_fix
# usage: bash tools/pmc_one.sh <tag> "<counters>" lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; C=$2; shift 2
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for l in "$@"; do
  VENC_HIP_LIB=$l VENC_NO_OVERLAP=1 timeout -k 10 250 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$l -- python3 $R/bench.py --workload 1080p --steps 1 --warmup 0 --cpu-frames 0 --no-profile --no-extras $PMC_FLAGS > $O/$l.log 2>&1 || { tail -5 $O/$l.log; exit 1; }
  python3 - $O/$l $l <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in ("k_me_pool", "k_me_fix", "k_deblock", "k_subpel", "k_hpel"):
    print(sys.argv[2], k, " ".join("%s=%.0f" % (c, v[0] / v[1]) for c, v in sorted(acc[k].items())), "launches", max(v[1] for v in acc[k].values()) if acc[k] else 0)
PY
done
