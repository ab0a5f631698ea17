#!/usr/bin/env python3
"""Regenerates tests/golden/*.  The fixtures are OUTPUTS OF THIS REPOSITORY'S
CPU ORACLE (oracle/), produced in the build container -- the reference ships no
golden vectors for this path, so these pin CPU<->GPU
agreement and regressions only; they say nothing about libx264.

Inputs are fully determined by (size, frame index, seed) through the synthetic
source of video-encoder_amd/csrc/venc_synth.h, so only expected outputs are stored.
"""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import orc_py  # noqa: E402

venc = importlib.import_module("video-encoder_amd")

CASES = [
    dict(name="cif_i_only", w=352, h=288, qp=26, slices=1, keyint=1, frames=3, seed=0x1234ABCD),    # BASELINE config 1 shape
    dict(name="cif_ippp", w=352, h=288, qp=26, slices=8, keyint=30, frames=5, seed=0x1234ABCD),
    dict(name="qcif_qp20_crop", w=178, h=146, qp=20, slices=3, keyint=3, frames=5, seed=77),
    dict(name="tiny_qp40", w=64, h=48, qp=40, slices=2, keyint=4, frames=6, seed=5),
    # every optional coding tool the C API's presets switch on (round 3: with Intra4x4): rate control, Intra4x4 / Intra16x16 with
    # plane prediction, quarter-sample vectors, in-loop filter
    dict(name="cif_all_tools", w=352, h=288, qp=26, slices=4, keyint=4, frames=7, seed=0x1234ABCD,
         tools=dict(rc=True, intra_modes=3, subpel=2, deblock=True)),
]


def run(case):
    enc = orc_py.Encoder(case["w"], case["h"], qp=case["qp"], n_slices=case["slices"], keyint=case["keyint"],
                         fps=(30, 1), sar=(1, 1), **case.get("tools", {}))
    out = dict(case)
    out["frames_sha256"], out["frame_bytes"] = [], []
    dump = {}
    stream = b""
    for t in range(case["frames"]):
        y, u, v = venc.synth_frame(case["w"], case["h"], t, case["seed"])
        fr = enc.encode(y, u, v)
        stream += fr.data
        out["frames_sha256"].append(hashlib.sha256(fr.data).hexdigest())
        out["frame_bytes"].append(len(fr.data))
        n = enc.mbw                      # first macroblock row
        dump["mv_%d" % t] = enc.mv()[:n]
        dump["sad_%d" % t] = enc.sad()[:n]
        dump["levels_%d" % t] = enc.levels()[:n]
        dump["recon_y_rows16_%d" % t] = enc.recon()[0][:16]
    out["stream_sha256"] = hashlib.sha256(stream).hexdigest()
    return out, dump, stream


def main():
    index = []
    for case in CASES:
        meta, dump, stream = run(case)
        np.savez_compressed(os.path.join(HERE, case["name"] + "_row0.npz"), **dump)
        if case["name"] == "tiny_qp40":
            open(os.path.join(HERE, "tiny_qp40.h264lp"), "wb").write(stream)   # length-prefixed NAL stream, 6 frames
        index.append(meta)
    json.dump(index, open(os.path.join(HERE, "golden.json"), "w"), indent=1)
    print("wrote", len(index), "cases")


if __name__ == "__main__":
    main()
