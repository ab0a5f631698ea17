"""Shared fixtures.  GPU tests are marked `gpu`; everything else runs on CPU."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def venc():
    mod = importlib.import_module("video-encoder_amd")
    if not os.path.exists(mod.LIB_HIP):
        mod.build()
    return mod


@pytest.fixture(scope="session")
def legacy_lib(venc):
    """TEST-ONLY build of the launcher with the first-generation kernels compiled in (csrc/Makefile: libvenc_hip_legacy.so),
    selected per engine by VENC_ME / VENC_INTER / VENC_INTRA.  Loaded RTLD_LOCAL beside the product library."""
    import ctypes
    path = os.path.join(venc.CSRC, "libvenc_hip_legacy.so")
    if not os.path.exists(path):
        venc.build()
    return venc.bind_hip(ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL))


@pytest.fixture(scope="session")
def orc():
    import orc_py
    orc_py.build()
    return orc_py


def unescape(nal_body):
    """Strip emulation-prevention bytes from a NAL body (after the header byte)."""
    out = bytearray()
    z = 0
    for c in nal_body:
        if z >= 2 and c == 3:
            z = 0
            continue
        out.append(c)
        z = z + 1 if c == 0 else 0
    return bytes(out)


def split_nals(data):
    """[(type, ref_idc, body_bytes)] of a length-prefixed NAL sequence."""
    out, o = [], 0
    while o < len(data):
        n = int.from_bytes(data[o:o + 4], "big")
        hdr = data[o + 4]
        out.append((hdr & 31, (hdr >> 5) & 3, data[o + 5:o + 4 + n]))
        o += 4 + n
    assert o == len(data)
    return out


def noise_frames(w, h, n, seed=7, motion=True):
    """Small deterministic test clip with real motion and texture (numpy only)."""
    rng = np.random.default_rng(seed)
    big = rng.integers(0, 256, (h // 4 + 40, w // 4 + 40)).astype(np.float64)
    big = np.kron(big, np.ones((4, 4)))
    k = np.ones(5) / 5
    big = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 1, big)
    big = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 0, big)
    frames = []
    for t in range(n):
        ox, oy = (40 + 3 * t, 30 + 2 * t) if motion else (40, 30)
        y = big[oy:oy + h, ox:ox + w] + rng.integers(-2, 3, (h, w))
        u = big[oy // 2 + 5:oy // 2 + 5 + h // 2, ox // 2:ox // 2 + w // 2] * 0.5 + 64
        v = 255 - big[oy // 2:oy // 2 + h // 2, ox // 2 + 7:ox // 2 + 7 + w // 2] * 0.5 - 64
        frames.append(tuple(np.clip(p, 0, 255).astype(np.uint8) for p in (y, u, v)))
    return frames
