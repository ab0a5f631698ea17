#!/bin/bash
# Phase budget of the shipped k_me_pool by truncation (VENC_ME_STOP=n: the kernel returns after phase n; 0 = whole kernel),
# every kernel alone on the chip (VENC_NO_OVERLAP=1), 1080p x 10 items: duration from the kernel trace and SQ counters
# from two rocprofv3 --pmc passes per stop (counters never together with --stats).  Truncated runs produce meaningless
# bitstreams; nothing checks them.   usage: bash tools/me_phase_pmc.sh <tag> [workload]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; WL=${2:-1080p}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
# stops 1-4 by truncation (they do not depend on the picture content); the bound test and the survivors' SADs cannot be
# truncated without changing the following pictures' references, so "search" = the whole kernel without its fused transform
# (VENC_FUSE=0: the transform runs as k_inter_fused, the pictures stay the same) and the transform = whole - search
for stop in 1 2 3 4 F 0; do
  i=0
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA"; do
    if [ $stop = F ]; then export VENC_FUSE=0; export VENC_ME_STOP=0; else unset VENC_FUSE; export VENC_ME_STOP=$stop; fi
    VENC_NO_OVERLAP=1 timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/s${stop}_$i -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --cpu-frames 0 --no-profile --no-extras > $O/s${stop}_$i.log 2>&1 || { tail -5 $O/s${stop}_$i.log; exit 1; }
    i=$((i+1))
  done
  echo "stop $stop done"
done
python3 $R/tools/me_phase_summary.py $O $WL | tee $O/me_phase_$WL.txt
for d in $O/s*_*; do [ -d "$d" ] && find $d -name "*agent_info.csv" -delete; done
