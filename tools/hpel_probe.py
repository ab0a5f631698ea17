#!/usr/bin/env python3
"""diagnostic: k_hpel's planes against a numpy restatement; prints where and how they differ"""
import importlib, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "oracle"))
venc = importlib.import_module("video-encoder_amd")
import orc_py
w, h = 304, 208
enc = orc_py.Encoder(w, h, qp=28, n_slices=2, keyint=30, threads=8, subpel=1)
eng = venc.Engine(w, h, qp=28, n_slices=2, max_items=1, in_slots=2, out_slots=2, subpel=1)
for t in range(2):
    y, u, v = venc.synth_frame(w, h, t, 3)
    if t == 1:
        ref = enc.recon()[0]
    enc.encode(y, u, v)
    eng.upload(t % 2, y, u, v); eng.encode_step([t % 2], [t % 2], t == 0, t & 15, [0]); eng.sync()
got = eng.debug(15).astype(np.int64)
H, W = ref.shape
q = np.pad(ref.astype(np.int64), 8, mode="edge")
taps = [1, -5, 20, 20, -5, 1]
def filt(a, axis):
    out = np.zeros_like(a)
    for k, t in enumerate(taps):
        out += t * np.roll(a, 2 - k, axis=axis)
    return out
b1 = filt(q, 1); j1 = filt(b1, 0)
jw = np.clip((j1 + 512) >> 10, 0, 255)[8:8 + H, 8:8 + W]
d = got[2] - jw
print("j diffs:", np.unique(d, return_counts=True))
bad = np.argwhere(d != 0)
print("x%4 of mismatches:", np.unique(bad[:, 1] % 4, return_counts=True))
for (yy, xx) in bad[:6]:
    col = b1[8 + yy - 2:8 + yy + 4, 8 + xx]
    print("at", yy, xx, "b1 column", col, "j1", j1[8 + yy, 8 + xx], "want", jw[yy, xx], "got", got[2][yy, xx], " (j1+512)/1024 = %.4f" % ((j1[8 + yy, 8 + xx] + 512) / 1024.0))
