#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_defer2
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/test.log 2>&1 || { tail -30 $O/test.log; exit 1; }
tail -3 $O/test.log
bash tools/ab_bench.sh r03_defer2 "VENC_ME_BLOCKS=3 VENC_FUSE=1" "VENC_ME_BLOCKS=3 VENC_FUSE=3" "VENC_ME_BLOCKS=3 VENC_FUSE=3 VENC_NZ_RPB=128" "VENC_ME_BLOCKS=4 VENC_FUSE=3" "VENC_ME_BLOCKS=3 VENC_FUSE=1" "VENC_ME_BLOCKS=3 VENC_FUSE=3"
for i in 0 1 2 3; do python3 - $O/ab_${i}_1080p.json $O/ab_${i}_2160p.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "iso:", {k: v for k, v in d.get("kernel_ms_per_step_isolated", {}).items() if v})
PY
done
