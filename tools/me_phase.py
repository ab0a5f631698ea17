#!/usr/bin/env python3
"""Phase costs of k_me_tile by truncation: VENC_ME_STOP=n makes the kernel return after phase n
(1 window staged, 2 horizontal sums, 3 vertical sums, 4 lower bounds + argmin, 5 starts + survivor
scan, 0 = whole kernel).  Prints the average k_me launch time of P steps 2..N for each n, 1080p x 10 items.
Diagnostic only: truncated runs produce meaningless vectors."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
venc = importlib.import_module("video-encoder_amd")
import torch
w, h, items, steps = 1920, 1080, 10, 8
fb = w * h * 3 // 2
clip = torch.empty(items * steps * fb, dtype=torch.uint8, device="cuda")
for i in range(items):
    venc.hip_lib().venc_synth_device(clip.data_ptr() + i * steps * fb, fb, w, h, 30 * i, steps, 0x1234ABCD)
torch.cuda.synchronize()
H = venc.hip_lib()
import ctypes as C
for stop in (1, 2, 3, 4, 6, 5, 0):  # 6 = after the start-point SADs, before the survivor scan  # (VENC_ME=tile for the predecessor kernel)
    os.environ["VENC_ME_STOP"] = str(stop)
    eng = venc.Engine(w, h, qp=26, n_slices=8, max_items=items, in_slots=0, out_slots=2 * items)
    eng.set_input(clip.data_ptr(), fb, items * steps)
    H.venc_set_overlap(eng.h, 0)
    for t in range(steps):
        if t == 2:
            eng.sync()
            H.venc_profile(eng.h, 1, 4096)
        eng.encode_step([i * steps + t for i in range(items)], [2 * i + (t & 1) for i in range(items)], t == 0, t & 15, [0] * items)
    eng.sync()
    n = H.venc_profile_kernels()
    ms, cnt = (C.c_double * n)(), (C.c_int * n)()
    H.venc_profile_read(eng.h, ms, cnt)
    names = [H.venc_profile_name(i).decode() for i in range(n)]
    k = names.index("k_me")
    print("stop %d: k_me %.1f us per launch (%d launches)" % (stop, 1e3 * ms[k] / max(1, cnt[k]), cnt[k]))
    eng.close()
