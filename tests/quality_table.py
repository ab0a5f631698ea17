"""What each optional coding tool buys, measured with the oracle (CPU): bytes and luma PSNR of short CIF clips, one tool at a
time and all together.  Not a test (no assertions): `python tests/quality_table.py` prints the table quoted in DESIGN.md."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib

from oracle import orc_py
from subpel_clip import subpel_clip

venc = importlib.import_module("video-encoder_amd")
W, H, N = 352, 288, 12


def clips():
    yy, xx = np.mgrid[0:H, 0:W]
    smooth = []
    for t in range(N):
        y = (128 + 70 * np.sin((xx + 1.5 * t) / 37.0) + 40 * np.cos((yy + 0.5 * t) / 23.0)).clip(0, 255).astype(np.uint8)
        c = (128 + 20 * np.sin((xx[::2, ::2] + t) / 31.0)).astype(np.uint8)
        smooth.append((y, c, c))
    return {"synthetic (integer pan, objects, noise)": [venc.synth_frame(W, H, t, 5) for t in range(N)],
            "quarter-sample pan of the same texture": subpel_clip(W, H, N, factor=4, sx=5, sy=3, seed=5),
            "smooth gradients, slow drift": smooth}


def run(frames, qp=30, **kw):
    e = orc_py.Encoder(W, H, qp=qp, n_slices=2, keyint=N, threads=8, **kw)
    n, se = 0, 0.0
    for (y, u, v) in frames:
        n += len(e.encode(y, u, v).data)
        se += float(np.mean((e.recon()[0][:H, :W].astype(np.float64) - y.astype(np.float64)) ** 2))
    return n, 10 * np.log10(255.0 ** 2 / (se / len(frames)))


if __name__ == "__main__":
    tools = [("none (the headline configuration)", {}), ("Intra16x16 V/H/DC/plane", dict(intra_modes=2)), ("half-sample vectors", dict(subpel=1)),
             ("quarter-sample vectors", dict(subpel=2)), ("in-loop filter", dict(deblock=True)),
             ("all of them", dict(intra_modes=2, subpel=2, deblock=True))]
    for name, fr in clips().items():
        print("\n%s, %d frames %dx%d, QP 30" % (name, N, W, H))
        b0, p0 = run(fr)
        for tname, kw in tools:
            b, p = run(fr, **kw)
            print("  %-36s %8d bytes (%+6.1f %%)  PSNR-Y %.2f dB (%+.2f)" % (tname, b, 100.0 * (b - b0) / b0, p, p - p0))
