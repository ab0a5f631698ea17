#!/usr/bin/env python3
"""Phase shares of k_me_tile from in-kernel cycle stamps (diagnostic build only).
Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DVENC_ME_STAMPS \
              -o video-encoder_amd/csrc/libvenc_hip_stamps.so video-encoder_amd/csrc/venc_hip.hip
Run:    VENC_HIP_LIB=libvenc_hip_stamps.so python tools/me_stamps.py
Read the SHARES, not the absolute time (stamps forbid overlaps the real kernel has)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
venc = importlib.import_module("video-encoder_amd")
w, h = 1920, 1080
eng = venc.Engine(w, h, qp=26, n_slices=8, max_items=1, in_slots=2, out_slots=2)
for t in range(6):
    eng.upload(t % 2, *venc.synth_frame(w, h, t))
    eng.encode_step([t % 2], [t % 2], t == 0, t & 15, [0])
eng.sync()
st = eng.debug(11).astype(np.float64)
names = ["P0 stage window", "P1 horizontal sums", "P2 vertical sums", "stage current tile", "C8 + P3 lower bounds",
         "P4a starts + survivor scan", "P4b list evaluation"]
tot = st[:7].sum()
for n, v in zip(names, st[:7]):
    print("%-28s %6.1f %%" % (n, 100 * v / tot))
print("cycles per wave (5 P frames): %.0f" % (tot / (5 * 4080)))
