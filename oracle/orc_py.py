"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED: see oracle/orc.h.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liborc.so")

MB_COEFS = 27 * 16
MAX_NALS = 512


class OrcParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("width", "height", "qp", "n_slices", "keyint", "fps_num", "fps_den",
                                       "sar_w", "sar_h", "threads", "idr_base", "rc_mode", "intra_modes", "deblock", "subpel")]


class OrcNal(C.Structure):
    _fields_ = [("type", C.c_int), ("ref_idc", C.c_int), ("offset", C.c_size_t), ("size", C.c_size_t)]


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("orc_enc.c", "orc_dec.c", "orc.h", "orc_interp.h", "orc_deblock.h", "orc_tables.h", "Makefile")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_enc_open.restype = C.c_void_p
        L.orc_enc_open.argtypes = [C.POINTER(OrcParams)]
        L.orc_enc_close.argtypes = [C.c_void_p]
        L.orc_enc_encode.restype = C.c_long
        L.orc_enc_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_size_t, C.POINTER(OrcNal), C.POINTER(C.c_int),
                                     C.POINTER(C.c_int)]
        for name, rt in (("orc_enc_mv", C.POINTER(C.c_int8)), ("orc_enc_mvq", C.POINTER(C.c_int8)), ("orc_enc_sad", C.POINTER(C.c_uint16)),
                         ("orc_enc_levels", C.POINTER(C.c_int16)), ("orc_enc_mb_bits", C.POINTER(C.c_uint32)),
                         ("orc_enc_slice_rows", C.POINTER(C.c_int))):
            getattr(L, name).restype = rt
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_enc_mb_width.argtypes = [C.c_void_p]
        L.orc_enc_mb_height.argtypes = [C.c_void_p]
        L.orc_enc_recon.restype = C.POINTER(C.c_uint8)
        L.orc_enc_recon.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.orc_enc_set_band.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_enc_import_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_fdct4x4.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_idct4x4.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_quant.argtypes = [C.c_int] * 4
        L.orc_dequant.argtypes = [C.c_int] * 3
        L.orc_me_rank.argtypes = [C.c_int] * 2
        L.orc_cavlc_block_bits.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.orc_intra4_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_intra4_topright_available.argtypes = [C.c_int] * 4
        L.orc_rc_qp.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int]
        L.orc_enc_picture_qp.argtypes = [C.c_void_p]
        L.orc_escape.restype = C.c_size_t
        L.orc_escape.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_dec_open.restype = C.c_void_p
        L.orc_dec_close.argtypes = [C.c_void_p]
        L.orc_dec_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        for name in ("orc_dec_width", "orc_dec_height", "orc_dec_crop_width", "orc_dec_crop_height"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_dec_plane.restype = C.POINTER(C.c_uint8)
        L.orc_dec_plane.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.orc_dec_error.restype = C.c_char_p
        L.orc_dec_error.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class Frame:
    """Result of encoding one frame with the oracle."""

    def __init__(self, data, nals, is_idr):
        self.data = data          # bytes: all NALs, 4-byte length prefixes
        self.nals = nals          # list of (type, ref_idc, offset, size)
        self.is_idr = is_idr


class Encoder:
    def __init__(self, width, height, qp=26, n_slices=1, keyint=30, fps=(30, 1), sar=(0, 0), threads=1, idr_base=0, rc=False, intra_modes=False, subpel=0, deblock=False):
        """rc: picture QP by spec/CODEC.md section 1a (the CRF approximation), qp = its base; else fixed QP.
        intra_modes: Intra16x16 vertical / horizontal / DC by SAD instead of DC only.
        subpel: 0 integer-sample vectors, 1 half-sample refinement, 2 half then quarter (spec/CODEC.md section 2a)."""
        self.L = lib()
        self.params = OrcParams(width, height, qp, n_slices, keyint, fps[0], fps[1], sar[0], sar[1], threads, idr_base, int(bool(rc)), int(intra_modes), int(bool(deblock)), int(subpel))
        self.h = self.L.orc_enc_open(C.byref(self.params))
        if not self.h:
            raise ValueError("orc_enc_open rejected the parameters")
        self.width, self.height = width, height
        self.mbw = self.L.orc_enc_mb_width(self.h)
        self.mbh = self.L.orc_enc_mb_height(self.h)
        self.nmb = self.mbw * self.mbh
        self._out = np.empty(max(1 << 16, self.nmb * 1700 + 4096), dtype=np.uint8)
        self._nals = (OrcNal * MAX_NALS)()

    def close(self):
        if self.h:
            self.L.orc_enc_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def encode(self, y, u, v):
        y = np.ascontiguousarray(y, dtype=np.uint8)
        u = np.ascontiguousarray(u, dtype=np.uint8)
        v = np.ascontiguousarray(v, dtype=np.uint8)
        assert y.shape == (self.height, self.width)
        assert u.shape == v.shape == (self.height // 2, self.width // 2)
        n, idr = C.c_int(0), C.c_int(0)
        r = self.L.orc_enc_encode(self.h, y.ctypes.data, y.strides[0], u.ctypes.data, u.strides[0], v.ctypes.data,
                                  v.strides[0], self._out.ctypes.data, self._out.size, self._nals, C.byref(n),
                                  C.byref(idr))
        if r < 0:
            raise RuntimeError("oracle encode failed: %d" % r)
        nals = [(self._nals[i].type, self._nals[i].ref_idc, self._nals[i].offset, self._nals[i].size)
                for i in range(n.value)]
        return Frame(self._out[:r].tobytes(), nals, bool(idr.value))

    # --- slice-parallel stand-in (orc.h) ---
    def set_band(self, slice_first, slice_count):
        if self.L.orc_enc_set_band(self.h, slice_first, slice_count) != 0:
            raise ValueError("orc_enc_set_band: bad slice range")

    def import_rows(self, mb_row0, mb_row1, y, u, v):
        y, u, v = (np.ascontiguousarray(a, dtype=np.uint8) for a in (y, u, v))
        n = mb_row1 - mb_row0
        assert y.shape == (n * 16, self.mbw * 16) and u.shape == v.shape == (n * 8, self.mbw * 8)
        if self.L.orc_enc_import_rows(self.h, mb_row0, mb_row1, y.ctypes.data, u.ctypes.data, v.ctypes.data) != 0:
            raise ValueError("orc_enc_import_rows: bad row range")

    # --- introspection of the last frame ---
    def picture_qp(self):
        return self.L.orc_enc_picture_qp(self.h)

    def mv(self):
        return np.ctypeslib.as_array(self.L.orc_enc_mv(self.h), (self.nmb, 2)).copy()

    def mvq(self):
        """the coded vectors in quarter-sample units (4 * mv() without sub-sample refinement)"""
        return np.ctypeslib.as_array(self.L.orc_enc_mvq(self.h), (self.nmb, 2)).copy()

    def sad(self):
        return np.ctypeslib.as_array(self.L.orc_enc_sad(self.h), (self.nmb,)).copy()

    def levels(self):
        return np.ctypeslib.as_array(self.L.orc_enc_levels(self.h), (self.nmb, 27, 16)).copy()

    def mb_bits(self):
        return np.ctypeslib.as_array(self.L.orc_enc_mb_bits(self.h), (self.nmb,)).copy()

    def slice_rows(self):
        n = min(max(1, self.params.n_slices), self.mbh)
        return np.ctypeslib.as_array(self.L.orc_enc_slice_rows(self.h), (n + 1,)).copy()

    def recon_unfiltered(self):
        """the last picture's reconstruction before the in-loop filter (= recon() without deblocking)"""
        self.L.orc_enc_recon_unfiltered.restype = C.POINTER(C.c_uint8)
        self.L.orc_enc_recon_unfiltered.argtypes = [C.c_void_p, C.c_int]
        cw, ch = self.mbw * 16, self.mbh * 16
        return [np.ctypeslib.as_array(self.L.orc_enc_recon_unfiltered(self.h, p), (ch // (2 if p else 1), cw // (2 if p else 1))).copy() for p in range(3)]

    def recon(self):
        out = []
        for p in range(3):
            st = C.c_int(0)
            ptr = self.L.orc_enc_recon(self.h, p, C.byref(st))
            h = self.mbh * 16 // (2 if p else 1)
            out.append(np.ctypeslib.as_array(ptr, (h, st.value)).copy())
        return out


class Decoder:
    def __init__(self):
        self.L = lib()
        self.h = self.L.orc_dec_open()

    def close(self):
        if self.h:
            self.L.orc_dec_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def decode(self, data):
        buf = np.frombuffer(data, dtype=np.uint8)
        r = self.L.orc_dec_decode(self.h, buf.ctypes.data, buf.size)
        if r < 0:
            raise RuntimeError("oracle decoder: " + self.L.orc_dec_error(self.h).decode())
        return r

    def planes(self):
        out = []
        ch = self.L.orc_dec_height(self.h)
        for p in range(3):
            st = C.c_int(0)
            ptr = self.L.orc_dec_plane(self.h, p, C.byref(st))
            out.append(np.ctypeslib.as_array(ptr, (ch // (2 if p else 1), st.value)).copy())
        return out

    def size(self):
        return (self.L.orc_dec_crop_width(self.h), self.L.orc_dec_crop_height(self.h))


def luma_interp(plane, xq, yq, bw, bh):
    """bw x bh luma prediction block whose top-left sample sits at (xq, yq) quarter-sample units of `plane`"""
    plane = np.ascontiguousarray(plane, dtype=np.uint8)
    out = np.empty((bh, bw), dtype=np.uint8)
    L = lib()
    L.orc_luma_interp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.orc_luma_interp.restype = None
    L.orc_luma_interp(plane.ctypes.data, plane.strides[0], plane.shape[1], plane.shape[0], xq, yq, bw, bh, out.ctypes.data)
    return out


def escape(rbsp):
    src = np.frombuffer(bytes(rbsp), dtype=np.uint8)
    out = np.empty(src.size * 3 // 2 + 4, dtype=np.uint8)
    n = lib().orc_escape(src.ctypes.data, src.size, out.ctypes.data)
    return out[:n].tobytes()
