#!/usr/bin/env python3
"""PCIe-inclusive rate (never bench.py's `value`): host frames through the reference's call sequence.
Writes the 1080p synthetic clip as YUV4MPEG2 into a RAM-backed file, then runs the C driver
(tools/encode_yuv: fread into the pinned picture, x264_encoder_encode with its H2D upload, flush) over it.
usage: python tools/pcie_rate.py [frames] [gop_batch]"""
import importlib
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
venc = importlib.import_module("video-encoder_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
gb = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w, h = 1920, 1080
fb = w * h * 3 // 2
buf = torch.empty(n * fb, dtype=torch.uint8, device="cuda")
assert venc.hip_lib().venc_synth_device(buf.data_ptr(), fb, w, h, 0, n, 0x1234ABCD) == 0
host = buf.cpu().numpy().reshape(n, fb)
del buf
path = ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp") + "/venc_pcie_%d.y4m" % os.getpid()
with open(path, "wb") as f:
    f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C420jpeg\n" % (w, h))
    for t in range(n):
        f.write(b"FRAME\n")
        f.write(host[t].tobytes())
try:
    out = {}
    for rep in range(3):  # the first run pays page-cache and context warm-up
        r = subprocess.run([os.path.join(ROOT, "tools", "encode_yuv"), "-i", path, "-n", str(n), "-g", str(gb), "-o", "-"],
                           capture_output=True, text=True, check=True)
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    out["what"] = "1080p, %d frames, YUV4MPEG2 from a RAM-backed file through tools/encode_yuv (fread + H2D upload per frame + encode + flush), gop_batch %d; third of three runs" % (n, gb)
    print(json.dumps(out))
finally:
    os.unlink(path)
