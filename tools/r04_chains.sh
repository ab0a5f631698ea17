#!/bin/bash
# two chains: correctness under VENC_CHAINS=2, then same-box A/B of one / two chains (4 and 8 hardware queues)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_chains}; mkdir -p $O; cd $R
VENC_CHAINS=2 GPU_MAX_HW_QUEUES=8 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "x264_api or config2 or launcher_variants or rate_control" > $O/test.log 2>&1; rc=$?
tail -4 $O/test.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab_bench.sh ${1:-r04_chains}/ab "VENC_CHAINS=1" "VENC_CHAINS=2" "VENC_CHAINS=2 GPU_MAX_HW_QUEUES=8" "VENC_CHAINS=1 GPU_MAX_HW_QUEUES=8"
grep -o '"bitstream_fnv1a": "[0-9a-f]*"' $O/ab/*.json | sort | uniq -c | sort -rn | head
