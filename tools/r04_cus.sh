#!/bin/bash
# side chain on a CU-masked stream: sweep of VENC_ENTROPY_CUS at both sizes, same box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/ab_bench.sh ${1:-r04_cus}/ab "VENC_ENTROPY_CUS=0" "VENC_ENTROPY_CUS=64" "VENC_ENTROPY_CUS=96" "VENC_ENTROPY_CUS=128" "VENC_ENTROPY_CUS=160" "VENC_ENTROPY_CUS=192" "VENC_ENTROPY_CUS=224" "VENC_ENTROPY_CUS=128 VENC_ENTROPY_BATCH=2"
