#!/bin/bash
# round 4, first call on the new P entropy chain: the stage and stream tests, then same-box A/B against round 3's library
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_c1; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stages or x264_api or white_noise" > $O/test.log 2>&1; rc=$?
tail -15 $O/test.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab_lib.sh r04_c1/ab1080 "--workload 1080p" libvenc_hip_r03.so libvenc_hip.so
bash tools/ab_lib.sh r04_c1/ab2160 "--workload 2160p" libvenc_hip_r03.so libvenc_hip.so
