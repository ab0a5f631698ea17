#!/usr/bin/env python3
"""per-macroblock bits / records of P pictures, engine against oracle; VENC_HIP_LIB selects the library; dumps to gpurun_out/probe_<tag>.npz"""
import importlib, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
venc = importlib.import_module("video-encoder_amd")
import orc_py as orc
w, h, qp, slices = 352, 288, 26, 4
enc = orc.Encoder(w, h, qp=qp, n_slices=slices, keyint=4, threads=8)
eng = venc.Engine(w, h, qp=qp, n_slices=slices, max_items=1, in_slots=2, out_slots=2)
out = {}
for t in range(3):
    y, u, v = venc.synth_frame(w, h, t)
    enc.encode(y, u, v)
    eng.upload(t % 2, y, u, v)
    eng.encode_step([t % 2], [t % 2], t == 0, t & 15, [0])
    eng.sync()
    if t == 0:
        continue
    bg, bo = eng.debug(3).astype(np.int64), enc.mb_bits().astype(np.int64)
    rec = eng.debug(2)
    out["rec%d" % t], out["bits%d" % t], out["lv%d" % t] = rec, bg, eng.debug(1)
    bad = np.nonzero(bg != bo)[0]
    print("t", t, "bad", len(bad), "levels equal", np.array_equal(out["lv%d" % t], enc.levels()))
eng.close()
np.savez(os.path.join(R, "gpurun_out", "probe_%s.npz" % sys.argv[1]), **out)
