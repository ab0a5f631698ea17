"""Diagnostic: I pictures through the three-kernel intra path and through the legacy wavefront kernel
(VENC_INTRA=legacy), compared plane by plane.  usage: python tools/intra_cmp.py [w h qp slices]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
venc = importlib.import_module("video-encoder_amd")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import noise_frames  # noqa: E402

w, h, qp, slices = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (176, 144, 51, 2)))
frames = noise_frames(w, h, 6, 0x1234ABCD)
os.environ["VENC_INTRA"] = "legacy"
ea = venc.Engine(w, h, qp=qp, n_slices=slices, max_items=1, in_slots=2, out_slots=2)
os.environ["VENC_INTRA"] = "new"
eb = venc.Engine(w, h, qp=qp, n_slices=slices, max_items=1, in_slots=2, out_slots=2)
for t, (y, u, v) in enumerate(frames):
    for e in (ea, eb):
        e.upload(t % 2, y, u, v)
        e.encode_step([t % 2], [t % 2], True, 0, [t & 1])
        e.sync()
    for p in range(3):
        a, b = ea.debug(4 + p).astype(int), eb.debug(4 + p).astype(int)
        bad = np.argwhere(a != b)
        sz = 16 if p == 0 else 8
        print("t%d plane %d: %d differing samples" % (t, p, len(bad)))
        for (yy, xx) in bad[:6]:
            print("   y %d x %d (mb %d,%d in-mb %d,%d) legacy %d new %d" % (yy, xx, xx // sz, yy // sz, xx % sz, yy % sz, a[yy, xx], b[yy, xx]))
    la, lb = ea.debug(1), eb.debug(1)
    print("t%d levels differ: %d; mbinfo differ: %d" % (t, int((la != lb).sum()), int((ea.debug(2)[:, :27] != eb.debug(2)[:, :27]).sum())))
    if t == 3:
        d0 = eb.debug(12)
        rc = eb.debug(13)
        lv = eb.debug(1)
        m = eb.nmb - 1
        print("last mb d0p", d0[m, :24].tolist())
        print("last mb wdc", rc[m, :24].tolist())
        print("last mb dc levels", lv[m, 16].tolist(), lv[m, 17, :4].tolist(), lv[m, 18, :4].tolist())
        print("prev mb d0p", d0[m - 1, :24].tolist())
        print("legacy rec block", ea.debug(4)[136:140, 160:176].tolist())
        print("new rec block", eb.debug(4)[136:140, 160:176].tolist())
