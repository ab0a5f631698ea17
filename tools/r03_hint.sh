#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_hint
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/test.log 2>&1 || { tail -30 $O/test.log; exit 1; }
tail -2 $O/test.log
bash tools/ab_bench.sh r03_hint "VENC_GOP_HINT=0" "VENC_GOP_HINT=1" "VENC_GOP_HINT=0" "VENC_GOP_HINT=1"
