#!/usr/bin/env python3
"""the PCIe-inclusive leg of bench.py alone, with the shim's per-call time split (VENC_TRACE=1): usage  python tools/ingest_trace.py [depth]"""
import importlib, os, sys, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ["VENC_TRACE"] = "1"
import torch
import bench
venc = importlib.import_module("video-encoder_amd")
wl = bench.WORKLOADS["1080p"]
w, h, frames = wl["w"], wl["h"], wl["frames"]
fb = w * h * 3 // 2
clip = torch.empty(frames * fb, dtype=torch.uint8, device="cuda")
assert venc.hip_lib().venc_synth_device(clip.data_ptr(), fb, w, h, 0, frames, 0x1234ABCD) == 0
for depth in (1, 2):
    r = bench.pcie_inclusive(venc, torch, wl, clip, fb, 10, 0, passes=5, input_depth=depth)
    print("depth", depth, r["value"], "Mpx/s", r["h2d_GBps"], "GB/s")
