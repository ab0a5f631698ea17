#!/bin/bash
# tests of the entropy chain, isolated kernel stats, same-box A/B against round 3's library
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04_c2}; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${2:-stages or x264_api or white_noise}" > $O/test.log 2>&1; rc=$?
tail -5 $O/test.log
[ $rc -eq 0 ] || exit $rc
bash tools/kstat.sh ${1:-r04_c2}/ks "VENC_NO_OVERLAP=1" --workload 1080p | grep -v "rocclr\|k_synth"
bash tools/ab_lib.sh ${1:-r04_c2}/ab1080 "--workload 1080p" libvenc_hip_r03.so libvenc_hip.so
bash tools/ab_lib.sh ${1:-r04_c2}/ab2160 "--workload 2160p" libvenc_hip_r03.so libvenc_hip.so
