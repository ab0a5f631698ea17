#!/bin/bash
# round-3 measurement batch 2: instruction issue costs, search-only kernel counters, baseline, deblock vs items per launch
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_c2; mkdir -p $O
cd $R
timeout -k 10 120 tools/valu_ubench > $O/valu_ubench.txt 2>&1; cat $O/valu_ubench.txt
cd /tmp; export TMPDIR=/tmp
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA"; do
  VENC_FUSE=0 VENC_NO_OVERLAP=1 timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/f0_$i -- python3 $R/bench.py --workload 1080p --steps 1 --warmup 0 --cpu-frames 0 --no-profile --no-extras > $O/f0_$i.log 2>&1 || { tail -5 $O/f0_$i.log; exit 1; }
  i=$((i+1))
done
python3 - $O <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for i in (0, 1):
    f = glob.glob(sys.argv[1] + "/f0_%d/*/*counter_collection.csv" % i)[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in ("k_me_pool", "k_me_fix", "k_inter_fused"):
    print("VENC_FUSE=0", k, " ".join("%s=%.0f" % (c.replace("SQ_", ""), v[0] / v[1]) for c, v in sorted(acc[k].items())))
PY
cd $R
for wl in 1080p 2160p; do
  timeout -k 10 200 python bench.py --workload $wl --no-extras --cpu-frames 0 --steps 10 > $O/base_$wl.json 2> $O/base_$wl.err || { tail -3 $O/base_$wl.err; exit 1; }
  python3 - $O/base_$wl.json $wl <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {}); k = d.get("kernel_ms_per_step", {})
print("base %s %9.1f Mpx/s %7.3f ms/step k_me %6.1f us (iso %6.1f) kernels: %s" % (sys.argv[2], d["value"], d["ms_per_step"], r.get("avg_launch_us", 0), r.get("isolated_avg_launch_us", 0), " ".join("%s=%.2f" % (a, b) for a, b in k.items() if b)))
PY
done
for gb in 10 20 30; do
  for t in "--deblock" "--crf 26 --intra-modes 2 --subpel 2 --deblock"; do
    timeout -k 10 200 python bench.py --no-extras --cpu-frames 0 --steps 6 --gop-batch $gb $t > $O/dbk.json 2> $O/dbk.err || { echo "gop-batch $gb $t FAILED"; tail -3 $O/dbk.err; continue; }
    python3 - $O/dbk.json "$gb $t" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernel_ms_per_step", {})
print("gop-batch %-50s %9.1f Mpx/s %7.3f ms/step  %s" % (sys.argv[2], d["value"], d["ms_per_step"], " ".join("%s=%.2f" % (a, b) for a, b in k.items() if b)))
PY
  done
done
