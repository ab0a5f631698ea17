#!/usr/bin/env python3
"""entries and bits per segment of P pictures (venc_debug_read selectors 16 / 17): usage  python tools/ent_probe.py [w h]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
venc = importlib.import_module("video-encoder_amd")
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
eng = venc.Engine(w, h, qp=26, n_slices=8, max_items=1, in_slots=2, out_slots=2)
for t in range(6):
    y, u, v = venc.synth_frame(w, h, t)
    eng.upload(t % 2, y, u, v)
    eng.encode_step([t % 2], [t % 2], t == 0, t & 15, [0])
    eng.sync()
    if t == 0:
        continue
    nseg = eng.n_slices * ((eng.nmb // eng.n_slices + 255) // 256 + 1)
    a = np.empty(4096, np.uint32)
    eng._ok(eng.L.venc_debug_read(eng.h, 16, 0, a.ctypes.data, a.nbytes), "read")
    b = np.empty(4096, np.uint32)
    eng._ok(eng.L.venc_debug_read(eng.h, 17, 0, b.ctypes.data, b.nbytes), "read")
    rec = eng.debug(2)
    cbp, cbpc = rec[:, 24] & 15, rec[:, 26] & 3
    n = 32
    print("t %d entries/segment min %d mean %.0f max %d  sum %d (%.2f per macroblock)  bits/segment mean %.0f  | luma 8x8 coded %.3f per MB, chroma dc %.3f ac %.3f of MBs" % (
        t, a[:n].min(), a[:n].mean(), a[:n].max(), a[:n].sum(), a[:n].sum() / eng.nmb, b[:n].mean(),
        sum(((cbp >> k) & 1).mean() for k in range(4)), (cbpc > 0).mean(), (cbpc == 2).mean()))
eng.close()
