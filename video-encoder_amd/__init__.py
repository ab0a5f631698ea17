"""video-encoder_amd -- MI355X-native macroblock encode loop behind the x264 entry points.

Host-side Python mirror of the two C interfaces of this repository:

* :class:`Engine`      -- include/venc_hip.h, the thin C-ABI of the HIP kernels
                          (frame slots in HBM -> slice RBSPs in pinned memory);
* :class:`X264Encoder` -- include/x264.h, the subset of the x264 API the
                          reference program calls (av_encode.c:384-443, :970,
                          :1076-1078), implemented in C by csrc/x264_shim.c.

Python is plumbing here (ctypes + numpy); all arithmetic runs in the HIP
kernels of csrc/venc_kernels.hip.  There is no CPU fallback: if the shared
libraries are missing or no HIP device is present, construction raises.

The package directory name contains a hyphen, import it with
``importlib.import_module("video-encoder_amd")``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_HIP = os.path.join(CSRC, os.environ.get("VENC_HIP_LIB", "libvenc_hip.so"))  # VENC_HIP_LIB: diagnostic builds
LIB_X264 = os.path.join(CSRC, "libx264_venc.so")
LIB_HOSTCHECK = os.path.join(CSRC, "libvenc_hostcheck.so")
LIB_IO = os.path.join(CSRC, "libvenc_io.so")

MB_COEFS = 27 * 16
MAX_ITEMS = 64


class VencError(RuntimeError):
    pass


def build(verbose=False):
    """Compile every native component in-tree (hipcc --offload-arch=gfx950, gcc)."""
    cmd = ["make", "-C", CSRC] + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)
    tools = os.path.join(os.path.dirname(_HERE), "tools")
    if os.path.exists(os.path.join(tools, "Makefile")):
        subprocess.check_call(["make", "-C", tools] + ([] if verbose else ["-s"]))


# ----------------------------------------------------------------------------- C-ABI structs
class VencConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("width", "height", "qp", "n_slices", "max_items", "in_slots", "out_slots",
                                       "device", "rc_mode", "intra_modes", "deblock", "subpel")]


class VencSlice(C.Structure):
    _fields_ = [("byte_off", C.c_uint32), ("nbytes", C.c_uint32), ("nbits", C.c_uint32), ("pad", C.c_uint32)]


class VencFrame(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("slices", C.POINTER(VencSlice)), ("n_slices", C.c_uint32),
                ("total_bytes", C.c_uint32), ("error", C.c_uint32)]


class X264Nal(C.Structure):
    _fields_ = [("i_ref_idc", C.c_int), ("i_type", C.c_int), ("i_payload", C.c_int),
                ("p_payload", C.POINTER(C.c_uint8))]


class _Vui(C.Structure):
    _fields_ = [("i_sar_width", C.c_int), ("i_sar_height", C.c_int)]


class _Rc(C.Structure):
    _fields_ = [("i_rc_method", C.c_int), ("i_qp_constant", C.c_int), ("f_rf_constant", C.c_float)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)


class _Hip(C.Structure):
    _fields_ = [("i_device", C.c_int), ("i_gop_batch", C.c_int), ("i_gop_base", C.c_int),
                ("i_slice_first", C.c_int), ("i_slice_count", C.c_int), ("pf_exchange", EXCHANGE_FN),
                ("p_exchange_user", C.c_void_p), ("i_rccl_world", C.c_int), ("i_rccl_rank", C.c_int),
                ("i_halo_rows", C.c_int), ("rccl_id", C.c_ubyte * 128), ("i_intra_modes", C.c_int), ("i_subpel", C.c_int), ("i_deblock", C.c_int), ("i_input_depth", C.c_int)]


class X264Param(C.Structure):
    _fields_ = [("i_width", C.c_int), ("i_height", C.c_int), ("i_csp", C.c_int), ("b_annexb", C.c_int),
                ("i_fps_num", C.c_int), ("i_fps_den", C.c_int), ("i_keyint_max", C.c_int), ("i_slice_count", C.c_int),
                ("i_bframe", C.c_int), ("i_frame_reference", C.c_int), ("b_zerolatency", C.c_int), ("vui", _Vui),
                ("rc", _Rc), ("hip", _Hip), ("psz_preset", C.c_char * 16), ("psz_tune", C.c_char * 16),
                ("psz_profile", C.c_char * 16)]


class X264Image(C.Structure):
    _fields_ = [("i_csp", C.c_int), ("i_plane", C.c_int), ("i_stride", C.c_int * 4),
                ("plane", C.POINTER(C.c_uint8) * 4)]


class X264Picture(C.Structure):
    _fields_ = [("i_type", C.c_int), ("i_qpplus1", C.c_int), ("b_keyframe", C.c_int), ("i_pts", C.c_int64),
                ("i_dts", C.c_int64), ("img", X264Image), ("opaque", C.c_void_p)]


X264_CSP_I420 = 1
X264_TYPE_AUTO = 0
X264_RC_CQP, X264_RC_CRF = 0, 1
NAL_SLICE, NAL_SLICE_IDR, NAL_SPS, NAL_PPS, NAL_FILLER = 1, 5, 7, 8, 12

class X264Tally(C.Structure):
    _fields_ = [("frames", C.c_uint64), ("bytes", C.c_uint64), ("keyframes", C.c_uint64), ("fnv1a", C.c_uint64),
                ("pass_frames", C.c_uint64), ("passes", C.c_uint64), ("pass_end_ns", C.c_uint64 * 256)]


SINK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(X264Nal), C.c_int, C.c_int, C.POINTER(X264Picture))

_hip = None
_x264 = None
_io = None


def _load(path):
    if not os.path.exists(path):
        raise VencError("%s is missing: run __graft_entry__.build() (make -C %s). "
                        "The HIP path has no CPU fallback." % (path, CSRC))
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


def bind_hip(L):
    """ctypes signatures of include/venc_hip.h on a loaded library (the product library, or the test-only legacy build)."""
    if True:
        L.venc_last_error.restype = C.c_char_p
        L.venc_open.argtypes = [C.POINTER(VencConfig), C.POINTER(C.c_void_p)]
        L.venc_close.argtypes = [C.c_void_p]
        for n in ("venc_mb_width", "venc_mb_height", "venc_n_slices", "venc_upload_sync", "venc_mark", "venc_sync", "venc_flush"):
            getattr(L, n).argtypes = [C.c_void_p]
        L.venc_frame_bytes.restype = C.c_size_t
        L.venc_frame_bytes.argtypes = [C.c_void_p]
        L.venc_upload.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.venc_set_input.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.venc_input_slot_ptr.restype = C.c_void_p
        L.venc_input_slot_ptr.argtypes = [C.c_void_p, C.c_int]
        L.venc_host_alloc.restype = C.c_void_p
        L.venc_host_alloc.argtypes = [C.c_size_t]
        L.venc_host_free.argtypes = [C.c_void_p]
        L.venc_encode_step.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int,
                                       C.POINTER(C.c_int)]
        L.venc_wait.argtypes = [C.c_void_p, C.c_int]
        L.venc_get_frame.argtypes = [C.c_void_p, C.c_int, C.POINTER(VencFrame)]
        L.venc_fetch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.venc_debug_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.venc_synth_frame.restype = None
        L.venc_synth_frame.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.venc_synth_device.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32]
        L.venc_profile.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.venc_set_overlap.argtypes = [C.c_void_p, C.c_int]
        L.venc_profile_name.restype = C.c_char_p
        L.venc_profile_name.argtypes = [C.c_int]
        L.venc_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.venc_profile_select.argtypes = [C.c_void_p, C.c_uint, C.c_int]
        L.venc_set_band.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.venc_band_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.venc_band_chunk_bytes.restype = C.c_size_t
        L.venc_band_chunk_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.venc_band_export.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
        L.venc_band_import.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int]
        L.venc_search_split.argtypes = [C.c_void_p]
        L.venc_stream_handle.restype = C.c_void_p
        L.venc_stream_handle.argtypes = [C.c_void_p]
        L.venc_numa_bind.argtypes = [C.c_int, C.c_int]
        L.venc_band_xbegin.argtypes = [C.c_void_p]
        L.venc_band_xend.argtypes = [C.c_void_p]
        L.venc_band_stream.restype = C.c_void_p
        L.venc_band_stream.argtypes = [C.c_void_p]
        L.venc_band_split_steps.restype = C.c_longlong
        L.venc_band_split_steps.argtypes = [C.c_void_p]
        L.venc_rccl_unique_id.argtypes = [C.c_void_p]
        L.venc_rccl_init_rank.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.venc_rccl_init_all.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.venc_rccl_world.argtypes = [C.c_void_p]
        L.venc_rccl_rank.argtypes = [C.c_void_p]
        L.venc_band_exchange_rccl.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.venc_rccl_agree.argtypes = [C.c_void_p, C.c_int]
        L.venc_rccl_exchanges.restype = C.c_longlong
        L.venc_rccl_exchanges.argtypes = [C.c_void_p]
        L.venc_rccl_always.restype = None
        L.venc_rccl_always.argtypes = [C.c_void_p, C.c_int]
        L.venc_rccl_poison.restype = None
        L.venc_rccl_poison.argtypes = [C.c_void_p, C.c_int]
        L.venc_rccl_pending.argtypes = []
        L.venc_rccl_comm_count.argtypes = [C.c_void_p]
    return L


def kernel_sources_sha16():
    """Digest of the device sources (csrc/*.hip, *.h): measurements kept under profiles/ carry it, and bench.py quotes a
    kept counter value only while it still describes the kernels that are running."""
    import glob
    import hashlib
    md = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h"))):
        md.update(os.path.basename(f).encode())
        md.update(open(f, "rb").read())
    return md.hexdigest()[:16]


def hip_lib():
    global _hip
    if _hip is None:
        _hip = bind_hip(_load(LIB_HIP))
    return _hip


def x264_lib():
    global _x264
    if _x264 is None:
        hip_lib()
        L = _load(LIB_X264)
        L.x264_param_default_preset.argtypes = [C.POINTER(X264Param), C.c_char_p, C.c_char_p]
        L.x264_param_apply_profile.argtypes = [C.POINTER(X264Param), C.c_char_p]
        L.x264_encoder_open.restype = C.c_void_p
        L.x264_encoder_open.argtypes = [C.POINTER(X264Param)]
        L.x264_picture_alloc.argtypes = [C.POINTER(X264Picture), C.c_int, C.c_int, C.c_int]
        L.x264_picture_clean.restype = None
        L.x264_picture_clean.argtypes = [C.POINTER(X264Picture)]
        L.x264_encoder_encode.argtypes = [C.c_void_p, C.POINTER(C.POINTER(X264Nal)), C.POINTER(C.c_int),
                                          C.POINTER(X264Picture), C.POINTER(X264Picture)]
        L.x264_encoder_delayed_frames.argtypes = [C.c_void_p]
        L.x264_encoder_close.restype = None
        L.x264_encoder_close.argtypes = [C.c_void_p]
        L.x264_venc_encode_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int64, C.c_void_p,
                                                C.c_void_p]
        L.x264_venc_encode_resident_loop.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int64,
                                                     C.c_void_p, C.c_void_p]
        L.x264_venc_engine.restype = C.c_void_p
        L.x264_venc_engine.argtypes = [C.c_void_p]
        _x264 = L
    return _x264


def io_lib():
    """Host-only C library: MP4 writer (include/venc_mp4.h) and frame source (include/venc_source.h).  No GPU needed."""
    global _io
    if _io is None:
        if not os.path.exists(LIB_IO):
            raise VencError("%s is missing: run __graft_entry__.build() (make -C %s)" % (LIB_IO, CSRC))
        L = C.CDLL(LIB_IO)
        L.venc_mp4_open.restype = C.c_void_p
        L.venc_mp4_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.venc_mp4_write_video_sample.argtypes = [C.c_void_p, C.POINTER(X264Nal), C.c_int, C.c_size_t, C.c_int, C.c_int64,
                                                  C.c_int64]
        L.venc_mp4_mux_video.argtypes = [C.c_void_p, C.POINTER(X264Nal), C.c_int, C.c_int, C.POINTER(X264Picture)]
        L.venc_mp4_close.argtypes = [C.c_void_p]
        L.venc_mp4_sample_count.restype = C.c_long
        L.venc_mp4_sample_count.argtypes = [C.c_void_p]
        L.venc_source_open.restype = C.c_void_p
        L.venc_source_open.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.venc_source_close.argtypes = [C.c_void_p]
        L.venc_source_width.argtypes = [C.c_void_p]
        L.venc_source_height.argtypes = [C.c_void_p]
        L.venc_source_rate.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.venc_source_aspect.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.venc_source_read.argtypes = [C.c_void_p, C.POINTER(X264Picture)]
        _io = L
    return _io


class Mp4Writer:
    """include/venc_mp4.h: the reference's enc_mp4_open / enc_mp4_mux_video / MP4Close (av_encode.c:614-834, :1104)."""

    def __init__(self, filename, width, height, time_base=(1, 30), sar=(1, 1)):
        self.L = io_lib()
        self.h = self.L.venc_mp4_open(os.fsencode(filename), width, height, time_base[0], time_base[1], sar[0], sar[1])
        if not self.h:
            raise VencError("venc_mp4_open failed for %s" % filename)

    def mux(self, payload, nals, pts, dts, keyframe):
        """One encoded frame: payload bytes + [(type, ref_idc, offset, size)] as delivered by the encoder."""
        buf = (C.c_uint8 * len(payload)).from_buffer_copy(payload)
        arr = (X264Nal * len(nals))()
        base = C.addressof(buf)
        for i, (ty, ref, off, size) in enumerate(nals):
            arr[i].i_type, arr[i].i_ref_idc, arr[i].i_payload = ty, ref, size
            arr[i].p_payload = C.cast(base + off, C.POINTER(C.c_uint8))
        pic = X264Picture()
        pic.i_pts, pic.i_dts, pic.b_keyframe = pts, dts, int(keyframe)
        if self.L.venc_mp4_mux_video(self.h, arr, len(nals), len(payload), C.byref(pic)) != 0:
            raise VencError("venc_mp4_mux_video failed")

    def flush(self):
        if self.L.venc_mp4_mux_video(self.h, None, 0, 0, None) != 0:
            raise VencError("venc_mp4_mux_video (flush) failed")

    def samples(self):
        return self.L.venc_mp4_sample_count(self.h)

    def close(self):
        if getattr(self, "h", None):
            rc = self.L.venc_mp4_close(self.h)
            self.h = None
            if rc != 0:
                raise VencError("venc_mp4_close failed")

    def __del__(self):
        try:
            self.close()
        except VencError:
            pass


class FrameSource:
    """include/venc_source.h: YUV4MPEG2 or headerless I420 frames into an x264_picture_t (av_encode.c:525-560)."""

    def __init__(self, filename, width=0, height=0):
        self.L = io_lib()
        self.h = self.L.venc_source_open(os.fsencode(filename), width, height)
        if not self.h:
            raise VencError("venc_source_open failed for %s" % filename)
        self.width, self.height = self.L.venc_source_width(self.h), self.L.venc_source_height(self.h)
        a, b = C.c_int(0), C.c_int(0)
        self.L.venc_source_rate(self.h, C.byref(a), C.byref(b))
        self.fps = (a.value, b.value)
        self.L.venc_source_aspect(self.h, C.byref(a), C.byref(b))
        self.sar = (a.value, b.value)

    def read_into(self, pic):
        """pic: X264Picture with allocated planes.  True: frame read; False: end of stream."""
        rc = self.L.venc_source_read(self.h, C.byref(pic))
        if rc < 0:
            raise VencError("venc_source_read failed (truncated or malformed input)")
        return rc == 1

    def close(self):
        if getattr(self, "h", None):
            self.L.venc_source_close(self.h)
            self.h = None

    def __del__(self):
        self.close()


def rccl_unique_id():
    """128-byte RCCL id (rank 0 calls this and hands the bytes to its peers: venc_rccl_unique_id)."""
    buf = (C.c_ubyte * 128)()
    if hip_lib().venc_rccl_unique_id(buf) != 0:
        raise VencError("venc_rccl_unique_id failed: %s" % hip_lib().venc_last_error().decode())
    return bytes(buf)


def synth_frame(width, height, frame_index, seed=0x1234ABCD):
    """Deterministic synthetic YUV420p frame (host implementation of csrc/venc_synth.h)."""
    y = np.empty((height, width), np.uint8)
    u = np.empty((height // 2, width // 2), np.uint8)
    v = np.empty((height // 2, width // 2), np.uint8)
    hip_lib().venc_synth_frame(width, height, frame_index, seed, y.ctypes.data, u.ctypes.data, v.ctypes.data)
    return y, u, v


# ----------------------------------------------------------------------------- Engine (venc_hip.h)
class Engine:
    """One encoder context on one GPU: frame slots in, slice RBSPs out."""

    def __init__(self, width, height, qp=26, n_slices=8, max_items=1, in_slots=4, out_slots=4, device=0, lib=None, rc=False, intra_modes=False, subpel=0, deblock=False):
        """lib: a library bound with bind_hip() instead of the product library (tests load the legacy build this way).
        rc: picture QP by spec/CODEC.md section 1a (qp = the base) instead of fixed QP.
        intra_modes: Intra16x16 vertical / horizontal / DC by SAD instead of DC only.
        subpel: 0 integer-sample vectors, 1 half-sample refinement, 2 half then quarter (spec/CODEC.md section 2a).
        deblock: in-loop deblocking filter inside every slice (spec/CODEC.md section 5a)."""
        self.L = lib if lib is not None else hip_lib()
        self.cfg = VencConfig(width, height, qp, n_slices, max_items, in_slots, out_slots, device, int(bool(rc)), int(intra_modes), int(bool(deblock)), int(subpel))
        h = C.c_void_p()
        rc = self.L.venc_open(C.byref(self.cfg), C.byref(h))
        if rc != 0:
            raise VencError("venc_open failed (%d): %s" % (rc, self.L.venc_last_error().decode()))
        self.h = h
        self.width, self.height = width, height
        self.mbw, self.mbh = self.L.venc_mb_width(h), self.L.venc_mb_height(h)
        self.nmb = self.mbw * self.mbh
        self.n_slices = self.L.venc_n_slices(h)
        self.frame_bytes = self.L.venc_frame_bytes(h)

    @classmethod
    def view(cls, handle):
        """Non-owning view of a venc_ctx* that somebody else opened (the x264 shim's engine)."""
        self = cls.__new__(cls)
        self.L = hip_lib()
        self.h = C.c_void_p(handle)
        self._borrowed = True
        self.mbw, self.mbh = self.L.venc_mb_width(self.h), self.L.venc_mb_height(self.h)
        self.nmb = self.mbw * self.mbh
        self.n_slices = self.L.venc_n_slices(self.h)
        self.frame_bytes = self.L.venc_frame_bytes(self.h)
        return self

    def close(self):
        if getattr(self, "h", None) and not getattr(self, "_borrowed", False):
            self.L.venc_close(self.h)
        self.h = None

    def __del__(self):
        self.close()

    def _ok(self, rc, what):
        if rc < 0:
            raise VencError("%s failed (%d): %s" % (what, rc, self.L.venc_last_error().decode()))
        return rc

    def upload(self, slot, y, u, v):
        y, u, v = (np.ascontiguousarray(a, np.uint8) for a in (y, u, v))
        self._ok(self.L.venc_upload(self.h, slot, y.ctypes.data, y.strides[0], u.ctypes.data, u.strides[0],
                                    v.ctypes.data, v.strides[0]), "venc_upload")
        self._ok(self.L.venc_upload_sync(self.h), "venc_upload_sync")

    def set_input(self, dev_ptr, frame_stride, n_frames):
        self._ok(self.L.venc_set_input(self.h, C.c_void_p(dev_ptr), frame_stride, n_frames), "venc_set_input")

    def input_slot_ptr(self, slot):
        return self.L.venc_input_slot_ptr(self.h, slot)

    def encode_step(self, in_slots, out_slots, is_idr, frame_num, idr_ids=None):
        n = len(in_slots)
        a = (C.c_int * n)(*in_slots)
        b = (C.c_int * n)(*out_slots)
        c = (C.c_int * n)(*(idr_ids if idr_ids is not None else [0] * n))
        self._ok(self.L.venc_encode_step(self.h, n, a, b, int(bool(is_idr)), frame_num, c), "venc_encode_step")

    def mark(self):
        return self._ok(self.L.venc_mark(self.h), "venc_mark")

    def wait(self, ticket):
        self._ok(self.L.venc_wait(self.h, ticket), "venc_wait")

    def sync(self):
        self._ok(self.L.venc_sync(self.h), "venc_sync")

    def get_slices(self, out_slot):
        """Slice RBSPs (bytes) of an output slot."""
        f = VencFrame()
        self._ok(self.L.venc_get_frame(self.h, out_slot, C.byref(f)), "venc_get_frame")
        out = []
        for s in range(f.n_slices):
            sl = f.slices[s]
            out.append(bytes(np.ctypeslib.as_array(f.data, (f.total_bytes,))[sl.byte_off:sl.byte_off + sl.nbytes]))
        return out

    # ---- slice-parallel multi-GPU mode (include/venc_hip.h, "Slice-parallel multi-GPU mode")
    def set_band(self, slice_first=0, slice_count=0):
        self._ok(self.L.venc_set_band(self.h, slice_first, slice_count), "venc_set_band")

    def band_rows(self, slice_first, slice_count):
        r0, r1 = C.c_int(0), C.c_int(0)
        self._ok(self.L.venc_band_rows(self.h, slice_first, slice_count, C.byref(r0), C.byref(r1)), "venc_band_rows")
        return r0.value, r1.value

    def band_chunk_bytes(self, mb_rows, halo_rows=0):
        return self.L.venc_band_chunk_bytes(self.h, mb_rows, halo_rows)

    def band_export(self, n_items, dev_ptr, item_stride, halo_rows=0):
        return self._ok(self.L.venc_band_export(self.h, n_items, C.c_void_p(dev_ptr), item_stride, halo_rows),
                        "venc_band_export")

    def band_import(self, n_items, dev_ptr, item_stride, slice_first, slice_count, halo_rows=0):
        return self._ok(self.L.venc_band_import(self.h, n_items, C.c_void_p(dev_ptr), item_stride, slice_first,
                                                slice_count, halo_rows), "venc_band_import")

    def band_xbegin(self):
        """export / import until band_xend() go to the exchange stream; the next P step splits its search around them"""
        return self._ok(self.L.venc_band_xbegin(self.h), "venc_band_xbegin")

    def band_xend(self):
        return self._ok(self.L.venc_band_xend(self.h), "venc_band_xend")

    def band_stream(self):
        return self.L.venc_band_stream(self.h)

    def band_split_steps(self):
        return int(self.L.venc_band_split_steps(self.h))

    def stream_handle(self):
        return self.L.venc_stream_handle(self.h)

    def debug(self, what, item=0):
        cw, ch = self.mbw * 16, self.mbh * 16
        shapes = {0: ((self.nmb,), np.uint32), 1: ((self.nmb, 27, 16), np.int16), 2: ((self.nmb, 32), np.uint8),
                  3: ((self.nmb,), np.uint32), 4: ((ch, cw), np.uint8), 5: ((ch // 2, cw // 2), np.uint8),
                  6: ((ch // 2, cw // 2), np.uint8), 7: ((self.nmb,), np.uint32),
                  8: ((ch + 25, (cw + 32) & ~15), np.uint32), 9: ((self.nmb,), np.uint8), 10: ((self.nmb,), np.uint8),
                  11: ((8 + 2 * 16384,), np.uint64), 12: ((self.nmb, 32), np.int32), 13: ((self.nmb, 128), np.int32),
                  14: ((self.nmb,), np.uint32), 15: ((3, ch, cw), np.uint8)}
        shape, dt = shapes[what]
        a = np.empty(shape, dt)
        self._ok(self.L.venc_debug_read(self.h, what, item, a.ctypes.data, a.nbytes), "venc_debug_read")
        return a

    def mv_sad(self, item=0):
        m = self.debug(0, item)
        dx = (m & 0xff).astype(np.int8)
        dy = ((m >> 8) & 0xff).astype(np.int8)
        return np.stack([dx, dy], axis=1), (m >> 16).astype(np.uint16)

    def mvq_sad(self, item=0):
        """the coded vectors in quarter samples and their SADs (engines with sub-sample refinement)"""
        m = self.debug(14, item)
        return np.stack([(m & 0xff).astype(np.int8), ((m >> 8) & 0xff).astype(np.int8)], axis=1), (m >> 16).astype(np.uint16)


# ----------------------------------------------------------------------------- X264Encoder (x264.h)
class EncodedFrame:
    def __init__(self, payload, nals, pts, dts, keyframe):
        self.payload = payload      # bytes: all NALs, contiguous
        self.nals = nals            # list of (type, ref_idc, offset, size)
        self.pts, self.dts, self.keyframe = pts, dts, keyframe


class X264Encoder:
    """Mirror of the reference's x264_context_t usage (av_encode.c:369-444, :543-547, :968-975, :1076-1083)."""

    def __init__(self, width, height, qp=26, slices=8, keyint=30, fps=(30, 1), sar=(0, 0), preset="medium",
                 tune="film", profile=None, gop_batch=1, device=0, annexb=False, crf=None, gop_base=0,
                 slice_first=0, slice_count=0, exchange=None, rccl=None, intra_modes=False, subpel=0, deblock=False, input_depth=1,
                 reference_defaults=False, preset_keyint=None):
        """slice_first / slice_count / exchange: slice-parallel multi-GPU mode (include/x264.h, hip extension);
        exchange(engine_view, n_items) is called after every picture that a P picture follows.
        rccl = (world, rank, id_bytes, halo_rows): the engine does the exchange itself in C over RCCL
        (include/venc_hip.h "RCCL exchange"); the band follows from world and rank."""
        self.L = x264_lib()
        p = X264Param()
        if self.L.x264_param_default_preset(C.byref(p), preset.encode() if preset else None,
                                            tune.encode() if tune else None) != 0:
            raise ValueError("x264: failed to set preset %s and tune %s" % (preset, tune))
        p.i_width, p.i_height = width, height
        p.b_annexb = int(annexb)
        p.i_fps_num, p.i_fps_den = fps
        p.vui.i_sar_width, p.vui.i_sar_height = sar
        if crf is not None:
            p.rc.i_rc_method, p.rc.f_rf_constant = X264_RC_CRF, float(crf)
        else:
            p.rc.i_rc_method, p.rc.i_qp_constant = X264_RC_CQP, qp
        p.hip.i_device = device
        p.hip.i_input_depth = int(input_depth)
        if not reference_defaults:
            p.i_keyint_max, p.i_slice_count = keyint, slices
            p.hip.i_gop_batch, p.hip.i_gop_base = gop_batch, gop_base
            p.hip.i_slice_first, p.hip.i_slice_count = slice_first, slice_count
            # Intra16x16 V / H / DC by SAD: the C API's preset default is on; this mirror defaults to the DC-only codec
            # definition that bench.py and the golden fixtures are frozen on
            p.hip.i_intra_modes = int(intra_modes)
            p.hip.i_subpel = int(subpel)  # (likewise: the C API's presets refine to quarter samples)
            p.hip.i_deblock = int(bool(deblock))  # (and filter in the loop)
        if preset_keyint is not None:
            p.i_keyint_max = int(preset_keyint)  # (a caller of libx264 that keeps ITS default key-frame interval, 250)
        # reference_defaults: ONLY what enc_x264_open sets (av_encode.c:384-401) -- size, b_annexb, frame rate, SAR, rate control --;
        # key-frame interval, slices, coding tools and the number of GOPs in lock-step stay what x264_param_default_preset chose
        if rccl is not None:
            world, rank, ident, halo = rccl
            p.hip.i_rccl_world, p.hip.i_rccl_rank, p.hip.i_halo_rows = world, rank, halo
            C.memmove(p.hip.rccl_id, bytes(ident), 128)
        self._exchange_error = None
        if exchange is not None:
            def _xc(user, engine_ptr, n_items):
                try:
                    exchange(Engine.view(engine_ptr), n_items)
                    return 0
                except BaseException as e:  # never unwind through the C frames
                    self._exchange_error = e
                    return -1
            self._exchange_cb = EXCHANGE_FN(_xc)  # keep alive as long as the encoder
            p.hip.pf_exchange = self._exchange_cb
        if self.L.x264_param_apply_profile(C.byref(p), profile.encode() if profile else None) != 0:
            raise ValueError("x264: failed to apply profile %s" % profile)
        self.param = p
        self.h = self.L.x264_encoder_open(C.byref(p))
        if not self.h:
            raise VencError("x264: failed to initialize encoder (no HIP device or bad parameters)")
        self.pic_out = X264Picture()
        self.width, self.height = width, height
        # hip.i_input_depth pictures, filled in turn by encode() (1: the reference's single pic_in)
        self._pics, self._pic_planes, self._n_enc = [], [], 0
        for _ in range(max(1, int(input_depth))):
            pic = X264Picture()
            if self.L.x264_picture_alloc(C.byref(pic), X264_CSP_I420, width, height) != 0:
                raise VencError("x264: could not allocate input picture")
            self._pics.append(pic)
            self._pic_planes.append([np.ctypeslib.as_array(pic.img.plane[0], (height, width)),
                                     np.ctypeslib.as_array(pic.img.plane[1], (height // 2, width // 2)),
                                     np.ctypeslib.as_array(pic.img.plane[2], (height // 2, width // 2))])
        self.pic_in, self._planes = self._pics[0], self._pic_planes[0]

    def close(self):
        if getattr(self, "h", None):
            for pic in self._pics:
                self.L.x264_picture_clean(C.byref(pic))
            self.L.x264_encoder_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _error(self, msg):
        e = VencError(msg)
        if getattr(self, "_exchange_error", None) is not None:  # raised inside the band-exchange callback
            e.__cause__ = self._exchange_error
        return e

    def _collect(self, size, nals, n):
        base = C.addressof(nals[0].p_payload.contents)
        payload = C.string_at(base, size)
        lst = [(nals[i].i_type, nals[i].i_ref_idc, C.addressof(nals[i].p_payload.contents) - base, nals[i].i_payload)
               for i in range(n)]
        return EncodedFrame(payload, lst, self.pic_out.i_pts, self.pic_out.i_dts, bool(self.pic_out.b_keyframe))

    def encode(self, y, u, v, pts):
        """One call of x264_encoder_encode(); returns an EncodedFrame or None (frame buffered)."""
        k = self._n_enc % len(self._pics)
        self._n_enc += 1
        self.pic_in, self._planes = self._pics[k], self._pic_planes[k]
        self.pic_in.i_type = X264_TYPE_AUTO
        self.pic_in.i_pts = pts
        np.copyto(self._planes[0], y)
        np.copyto(self._planes[1], u)
        np.copyto(self._planes[2], v)
        nals, n = C.POINTER(X264Nal)(), C.c_int(0)
        size = self.L.x264_encoder_encode(self.h, C.byref(nals), C.byref(n), C.byref(self.pic_in),
                                          C.byref(self.pic_out))
        if size < 0:
            raise self._error("x264: encoder error")
        return self._collect(size, nals, n.value) if size > 0 else None

    def gop_batch(self):
        """closed GOPs advanced in lock-step: hip.i_gop_batch, or what 0 resolved to, after the cap by device memory"""
        self.L.x264_venc_gop_batch.argtypes = [C.c_void_p]
        return self.L.x264_venc_gop_batch(self.h)

    def delayed_frames(self):
        return self.L.x264_encoder_delayed_frames(self.h)

    def flush(self):
        """The reference's drain loop (av_encode.c:1076-1083)."""
        out = []
        while self.delayed_frames() > 0:
            nals, n = C.POINTER(X264Nal)(), C.c_int(0)
            size = self.L.x264_encoder_encode(self.h, C.byref(nals), C.byref(n), None, C.byref(self.pic_out))
            if size < 0:
                raise self._error("x264: encoder error")
            if size > 0:
                out.append(self._collect(size, nals, n.value))
        return out

    def encode_resident(self, dev_ptr, frame_stride, n_frames, sink=None, first_pts=0, repeats=1):
        """Encode frames already in HBM; sink(payload_bytes, nal_list, pts, keyframe) per frame, or count only.
        repeats > 1: that many passes of the clip as one stream (x264_venc_encode_resident_loop)."""
        raised = []

        def _cb(user, nals, n, size, pic):
            try:
                if sink is not None:
                    base = C.addressof(nals[0].p_payload.contents)
                    lst = [(nals[i].i_type, nals[i].i_ref_idc, C.addressof(nals[i].p_payload.contents) - base,
                            nals[i].i_payload) for i in range(n)]
                    sink(C.string_at(base, size), lst, pic.contents.i_pts, bool(pic.contents.b_keyframe))
                return 0
            except BaseException as e:  # never unwind through the C frames: stop the loop, re-raise below
                raised.append(e)
                return -1
        cb = SINK_FN(_cb)
        r = self.L.x264_venc_encode_resident_loop(self.h, C.c_void_p(dev_ptr), frame_stride, n_frames, repeats, first_pts,
                                                  C.cast(cb, C.c_void_p), None)
        if raised:
            raise raised[0]
        if r < 0:
            raise self._error("x264_venc_encode_resident_loop failed: %s" % hip_lib().venc_last_error().decode())
        return r

    def encode_resident_tally(self, dev_ptr, frame_stride, n_frames, first_pts=0, repeats=1):
        """Same, with the C tally sink (no Python in the loop): returns X264Tally.  repeats > 1: that many passes of
        the resident clip as one stream (x264_venc_encode_resident_loop): consecutive passes overlap."""
        t = X264Tally()
        t.pass_frames = n_frames  # CLOCK_MONOTONIC stamps at the end of every pass (x264_venc_tally.pass_end_ns)
        fn = C.cast(self.L.x264_venc_tally_sink, C.c_void_p)
        r = self.L.x264_venc_encode_resident_loop(self.h, C.c_void_p(dev_ptr), frame_stride, n_frames, repeats, first_pts, fn,
                                                  C.cast(C.pointer(t), C.c_void_p))
        if r != n_frames * repeats:
            raise self._error("x264_venc_encode_resident_loop delivered %d of %d frames: %s" % (
                r, n_frames * repeats, hip_lib().venc_last_error().decode()))
        return t

    # ---- per-kernel HIP-event timing of the engine behind this encoder
    def profile(self, enable=True, capacity=8192):
        if hip_lib().venc_profile(self.L.x264_venc_engine(self.h), int(enable), capacity) != 0:
            raise VencError("venc_profile failed")

    def profile_select(self, names=None, stride=1):
        """Event pairs only around every stride-th launch of the named kernels (None = all): each pair costs the
        stream a few microseconds."""
        H = hip_lib()
        mask = 0xffffffff
        if names is not None:
            all_names = [H.venc_profile_name(i).decode() for i in range(H.venc_profile_kernels())]
            mask = 0
            for n in names:
                mask |= 1 << all_names.index(n)
        if H.venc_profile_select(self.L.x264_venc_engine(self.h), mask, stride) != 0:
            raise VencError("venc_profile_select failed")

    def set_overlap(self, enable):
        if hip_lib().venc_set_overlap(self.L.x264_venc_engine(self.h), int(bool(enable))) != 0:
            raise VencError("venc_set_overlap failed")

    def profile_read(self):
        H = hip_lib()
        n = H.venc_profile_kernels()
        ms, cnt = (C.c_double * n)(), (C.c_int * n)()
        rc = H.venc_profile_read(self.L.x264_venc_engine(self.h), ms, cnt)
        if rc < 0:
            raise VencError("venc_profile_read failed")
        return {H.venc_profile_name(i).decode(): (ms[i], cnt[i]) for i in range(n)}, bool(rc)
