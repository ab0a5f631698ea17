#!/bin/bash
# SQ counters + isolated durations of k_deblock for library variants: bash tools/r03_dbk_pmc.sh <tag> var_a.so var_b.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/video-encoder_amd/csrc
O=$R/gpurun_out/$1; shift; mkdir -p $O
cp $C/libvenc_hip.so $O/orig.so
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  cp $C/$v $C/libvenc_hip.so
  i=0
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA"; do
    VENC_NO_OVERLAP=1 timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${v%.so}_$i -- python3 $R/bench.py --workload 1080p --steps 1 --warmup 0 --cpu-frames 0 --no-profile --no-extras --deblock > $O/${v%.so}_$i.log 2>&1 || { tail -5 $O/${v%.so}_$i.log; cp $O/orig.so $C/libvenc_hip.so; exit 1; }
    i=$((i+1))
  done
  python3 - $O/${v%.so} $v <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0]); dur = collections.defaultdict(list)
for i in (0, 1):
    f = glob.glob(sys.argv[1] + "_%d/*/*counter_collection.csv" % i)[0]
    seen = set()
    for r in csv.DictReader(open(f)):
        if "k_deblock" not in r["Kernel_Name"]: continue
        g = int(r["Grid_Size"]) if "Grid_Size" in r else 0
        a = acc[(g, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
        if i == 0 and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dur[g].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for g in sorted(dur):
    d = sorted(dur[g])
    print(sys.argv[2], "grid", g, "launches", len(d), "us min %.0f med %.0f max %.0f" % (d[0], d[len(d) // 2], d[-1]), " ".join("%s=%.0f" % (c.replace("SQ_", ""), v[0] / v[1]) for (gg, c), v in sorted(acc.items()) if gg == g))
PY
done
cp $O/orig.so $C/libvenc_hip.so
for d in $O/*_[01]; do [ -d "$d" ] && find $d -name "*agent_info.csv" -delete; done
