#!/bin/bash
# Host-side C under AddressSanitizer + UBSan (CPU build only; GPU sanitizers are not available on the pool):
# builds libvenc_io.so and libx264_venc.so with the sanitizers IN PLACE, runs the CPU tests that exercise them
# (NAL wrapping / emulation prevention, parameter sets, MP4 writer, frame source, ABI), then rebuilds the normal libraries.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/video-encoder_amd/csrc
F="-O1 -g -std=c99 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer"
(cd $C && gcc $F -o libvenc_io.so mp4_mux.c yuv_source.c && gcc $F -o libx264_venc.so x264_shim.c h264_syntax.c -L. -lvenc_hip -Wl,-rpath,'$ORIGIN' -lpthread)
rc=0
(cd $R && LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest tests/test_host_syntax.py tests/test_io.py tests/test_abi.py -x -q -m "not gpu") || rc=$?
make -C $C -B libx264_venc.so libvenc_io.so > /dev/null
exit $rc
