#!/bin/bash
# entropy / parity subset of the GPU tests, isolated kernel stats, then a short same-box bench at both sizes:  bash tools/gpu_quick.sh <tag> [pytest -k expression]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-quick}; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${2:-stages or x264_api or white_noise or beyond_their_lds or rate_control}" > $O/test.log 2>&1; rc=$?
tail -4 $O/test.log
[ $rc -eq 0 ] || exit $rc
bash tools/kstat.sh ${1:-quick}/ks "VENC_NO_OVERLAP=1" --workload 1080p | grep -v "rocclr\|k_synth"
WL=1080p REPS=2 STEPS=30 bash tools/ab_env.sh ${1:-quick}/b1080 "VENC_X=0"
WL=2160p REPS=2 STEPS=12 bash tools/ab_env.sh ${1:-quick}/b2160 "VENC_X=0"
