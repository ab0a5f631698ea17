#!/usr/bin/env python3
"""Ingest-inclusive rate (host pictures through x264_encoder_encode, bench.py pcie_inclusive) against hip.i_input_depth and the number of
passes (the pipeline's fill and drain are inside the timing: more passes = a smaller share).  usage: python tools/ingest_depth.py"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

venc = importlib.import_module("video-encoder_amd")
venc.build()
wl = bench.WORKLOADS["1080p"]
w, h, frames = wl["w"], wl["h"], wl["frames"]
fb = w * h * 3 // 2
clip = torch.empty(frames * fb, dtype=torch.uint8, device="cuda")
assert venc.hip_lib().venc_synth_device(clip.data_ptr(), fb, w, h, 0, frames, bench.SEED if hasattr(bench, "SEED") else 0x1234ABCD) == 0
for depth in (1, 2, 3, 4):
    for passes in (3, 10):
        r = bench.pcie_inclusive(venc, torch, wl, clip, fb, 10, 0, passes=passes, input_depth=depth)
        print("input_depth %d passes %2d: %8.1f Mpx/s  %5.2f GB/s of H2D  %.3f ms per clip" % (depth, passes, r.get("value", 0), r.get("h2d_GBps", 0), r.get("ms_per_clip", 0)), flush=True)
