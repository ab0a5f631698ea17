#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_c6}; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "launcher_variants" > $O/test.log 2>&1; rc=$?
tail -4 $O/test.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab_bench.sh ${1:-r04_c6}/ab "VENC_PRIO=0" "VENC_PRIO=1" "VENC_ENTROPY_CUS=128" "VENC_ENTROPY_CUS=192"
bash tools/pmc_kernels.sh ${1:-r04_c6}/pmc "k_inter_fused k_entropy_p k_pack_p k_me_pool4 k_me_fix"
