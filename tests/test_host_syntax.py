"""CPU tests of the host-side C of the boundary (csrc/h264_syntax.c): parameter sets,
NAL wrapping with emulation prevention and the 4-byte length prefixes the reference's muxer relies on
(av_encode.c:635-640, :692-705, :722-727), against the oracle's own writers (oracle/: this repository's restatement;
parity with libx264 unpinned)."""
import ctypes as C

import numpy as np
import pytest

from conftest import split_nals, unescape


class Seq(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("width", "height", "qp", "fps_num", "fps_den", "sar_w", "sar_h")]


@pytest.fixture(scope="module")
def syn(venc):
    L = venc.x264_lib()
    L.h264_write_sps.restype = C.c_size_t
    L.h264_write_sps.argtypes = [C.POINTER(Seq), C.c_void_p, C.c_size_t]
    L.h264_write_pps.restype = C.c_size_t
    L.h264_write_pps.argtypes = [C.POINTER(Seq), C.c_void_p, C.c_size_t]
    L.h264_wrap_nal.restype = C.c_size_t
    L.h264_wrap_nal.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    L.h264_level_idc.argtypes = [C.c_int, C.c_int, C.c_int]
    return L


@pytest.mark.parametrize("w,h,qp,fps,sar", [
    (352, 288, 26, (30, 1), (1, 1)), (1920, 1080, 26, (30, 1), (1, 1)), (3840, 2160, 20, (60, 1), (1, 1)),
    (178, 146, 51, (30000, 1001), (16, 15)), (16, 16, 12, (25, 1), (0, 0)), (640, 360, 33, (24, 1), (4, 3)),
])
def test_parameter_sets_equal_the_oracles(syn, orc, w, h, qp, fps, sar):
    enc = orc.Encoder(w, h, qp=qp, n_slices=1, keyint=1, fps=fps, sar=sar)
    z = (np.zeros((h, w), np.uint8), np.zeros((h // 2, w // 2), np.uint8), np.zeros((h // 2, w // 2), np.uint8))
    nals = split_nals(enc.encode(*z).data)
    want_sps = [unescape(b) for (ty, _, b) in nals if ty == 7][0]
    want_pps = [unescape(b) for (ty, _, b) in nals if ty == 8][0]
    s = Seq(w, h, qp, fps[0], fps[1], sar[0], sar[1])
    buf = (C.c_uint8 * 128)()
    n = syn.h264_write_sps(C.byref(s), buf, 128)
    assert bytes(buf[:n]) == want_sps
    assert want_sps[0] == 66 and want_sps[1] == 0xC0  # the bytes the reference copies into avcC (av_encode.c:692-715)
    mbs = ((w + 15) // 16) * ((h + 15) // 16)
    assert want_sps[2] == syn.h264_level_idc(mbs, fps[0], fps[1])
    n = syn.h264_write_pps(C.byref(s), buf, 128)
    assert bytes(buf[:n]) == want_pps
    assert syn.h264_write_sps(C.byref(s), buf, 3) == 0  # too small a buffer is reported, not overrun


def test_nal_wrapping_escapes_like_the_oracle(syn, orc):
    rng = np.random.default_rng(5)
    cases = [b"", b"\x00", b"\x00\x00", b"\x00\x00\x00", b"\x00\x00\x01", b"\x00\x00\x02\x00\x00\x03", b"\x00" * 40,
             b"\x01\x00\x00\x03\x00\x00", bytes(range(256))]
    for _ in range(40):  # zero-rich random payloads: runs of zeros of every length next to 0..3
        a = rng.integers(0, 4, rng.integers(1, 400)).astype(np.uint8)
        a[rng.random(a.size) < 0.6] = 0
        cases.append(a.tobytes())
    for rbsp in cases:
        for annexb in (0, 1):
            out = (C.c_uint8 * (len(rbsp) * 2 + 16))()
            src = (C.c_uint8 * max(1, len(rbsp))).from_buffer_copy(rbsp or b"\0")
            n = syn.h264_wrap_nal(out, annexb, 2, 1, src, len(rbsp))
            got = bytes(out[:n])
            body = orc.escape(rbsp)
            assert got[4] == (2 << 5) | 1 and got[5:] == body
            assert got[:4] == (b"\x00\x00\x00\x01" if annexb else (1 + len(body)).to_bytes(4, "big"))
            assert unescape(got[5:]) == rbsp
            # no start-code emulation inside the NAL
            assert b"\x00\x00\x00" not in got[4:] and b"\x00\x00\x01" not in got[4:] and b"\x00\x00\x02" not in got[4:]


def test_level_table(syn):
    assert syn.h264_level_idc(396, 30, 1) == 13           # CIF 30 fps
    assert syn.h264_level_idc(8160, 30, 1) == 40          # 1080p30
    assert syn.h264_level_idc(8160, 60, 1) == 42          # 1080p60
    assert syn.h264_level_idc(32400, 30, 1) == 51         # 2160p30
    assert syn.h264_level_idc(32400, 60, 1) == 52         # 2160p60
    assert syn.h264_level_idc(99, 15, 1) == 10            # QCIF 15 fps
