#!/bin/bash
# kernel timeline of a steady-state unit (the GOPs the encoder advances in lock-step: three passes of the 1080p clip) (rocprofv3 kernel trace): tools/timeline.py output under gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-timeline}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $R/bench.py --workload ${2:-1080p} --no-extras --cpu-frames 0 --steps 12 --warmup 3 --no-profile --no-check-pass > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 $R/tools/timeline.py $O/prof ${3:-6} > $O/timeline_${2:-1080p}.txt; cat $O/timeline_${2:-1080p}.txt
find $O/prof -name "*agent_info.csv" -delete
