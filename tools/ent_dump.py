#!/usr/bin/env python3
import importlib, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
venc = importlib.import_module("video-encoder_amd")
import orc_py as orc
w, h, qp, slices = 352, 288, 26, 4
enc = orc.Encoder(w, h, qp=qp, n_slices=slices, keyint=4, threads=8)
eng = venc.Engine(w, h, qp=qp, n_slices=slices, max_items=1, in_slots=2, out_slots=2)
for t in range(2):
    y, u, v = venc.synth_frame(w, h, t)
    enc.encode(y, u, v)
    eng.upload(t % 2, y, u, v)
    eng.encode_step([t % 2], [t % 2], t == 0, t & 15, [0])
    eng.sync()
def rd(what, n):
    a = np.empty(n, np.uint32)
    eng._ok(eng.L.venc_debug_read(eng.h, what, 0, a.ctypes.data, a.nbytes), "read")
    return a
spp, cap = 1, 27 * 110
cnt = rd(16, slices * spp * 4).reshape(-1, 4)
lst = rd(18, slices * spp * cap).reshape(-1, cap)
off = rd(19, slices * spp * (cap + 4)).reshape(-1, cap + 4)
cls = rd(20, slices * spp * cap).reshape(-1, cap)
bg, bo = eng.debug(3).astype(np.int64), enc.mb_bits().astype(np.int64)
rec = eng.debug(2)
print("counts", cnt.tolist())
for sg in range(slices):
    n, nh, nn = cnt[sg][:3]
    c = cls[sg][:n]
    ranks = np.sort(c >> 13)
    print("seg", sg, "ranks are a permutation", np.array_equal(ranks, np.arange(n)), "list matches cls", all((lst[sg][r >> 13] & 31) == (r & 31) for r in c))
bad = np.nonzero(bg != bo)[0]
for m in bad[:6]:
    sg = [i for i in range(slices) if (lst[i][:cnt[i][0]] >> 5 == m).any()][0]
    ks = np.nonzero(lst[sg][:cnt[sg][0]] >> 5 == m)[0]
    print("mb", m, "gpu", bg[m], "orc", bo[m], "cbp", rec[m, 24] & 15, "cbpc", rec[m, 26], "entries (slot, len):", [(int(lst[sg][k] & 31), int(off[sg][k + 1] - off[sg][k])) for k in ks], "nnz", rec[m, :24].tolist())
eng.close()
