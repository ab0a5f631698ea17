#!/usr/bin/env python3
"""Survivor statistics of the motion search (diagnostic): per P frame of the 1080p synthetic clip, the
distribution of me_cnt (candidates besides the start points whose lower bound passed; saturates at 255)
and the number of macroblocks flagged for the exhaustive fix-up."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
venc = importlib.import_module("video-encoder_amd")
w, h = 1920, 1080
eng = venc.Engine(w, h, qp=26, n_slices=8, max_items=1, in_slots=2, out_slots=2)
for t in range(8):
    eng.upload(t % 2, *venc.synth_frame(w, h, t))
    eng.encode_step([t % 2], [t % 2], t == 0, t & 15, [0])
    eng.sync()
    if t:
        c = eng.debug(9).astype(np.int64); f = eng.debug(10)
        q = np.percentile(c, [50, 90, 99, 99.9])
        print("frame %d: mean %.2f  p50 %d p90 %d p99 %d p99.9 %d  >16: %d >32: %d >64: %d  >=255: %d  flagged %d of %d  total survivors %d" % (
            t, c.mean(), q[0], q[1], q[2], q[3], (c > 16).sum(), (c > 32).sum(), (c > 64).sum(), (c >= 255).sum(), int(f.sum()), c.size, c.sum()))
