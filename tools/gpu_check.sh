#!/bin/bash
# One GPU-box call: the -m gpu suite, then the default bench line.  usage (from the repo root on the box):
#   bash tools/gpu_check.sh <tag> [pytest -k expression]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-check}
mkdir -p $O
cd $R
if [ -n "$2" ]; then K=(-k "$2"); else K=(); fi
timeout -k 10 800 python -m pytest tests -m gpu -x -q --durations=15 "${K[@]}" > $O/test.log 2>&1
rc=$?
tail -25 $O/test.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err
rc=$?
tail -3 $O/bench.err
cut -c1-1500 $O/bench.json
exit $rc
