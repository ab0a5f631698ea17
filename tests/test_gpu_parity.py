"""GPU parity tests (-m gpu): every HIP stage and the final bitstream must be
bit-exact against the CPU oracle (oracle/, PARITY UNPINNED w.r.t. libx264 --
the oracle is this repository's own restatement, see oracle/orc.h).

All calls go through the C-ABI: include/venc_hip.h for the stage-level checks,
include/x264.h (the reference's own entry points, av_encode.c:970) for the
frame-in / bitstream-out checks.
"""
import hashlib

import numpy as np
import pytest

from conftest import noise_frames, split_nals, unescape

pytestmark = pytest.mark.gpu


def _clip(venc, w, h, n, seed, kind):
    if kind == "synth":
        return [venc.synth_frame(w, h, t, seed) for t in range(n)]
    return noise_frames(w, h, n, seed)


def _stage_compare(venc, orc, w, h, qp, slices, keyint, n, kind="synth", seed=0x1234ABCD, lib=None):
    enc = orc.Encoder(w, h, qp=qp, n_slices=slices, keyint=keyint, threads=8)
    eng = venc.Engine(w, h, qp=qp, n_slices=slices, max_items=1, in_slots=2, out_slots=2, lib=lib)
    frames = _clip(venc, w, h, n, seed, kind)
    for t, (y, u, v) in enumerate(frames):
        ref = enc.encode(y, u, v)
        idr = t % keyint == 0
        eng.upload(t % 2, y, u, v)
        eng.encode_step([t % 2], [t % 2], idr, (t % keyint) & 15, [(t // keyint) & 1])
        eng.sync()
        tag = "%dx%d qp%d s%d t%d %s" % (w, h, qp, slices, t, "I" if idr else "P")
        mv, sad = eng.mv_sad()
        assert np.array_equal(mv, enc.mv()), "mv " + tag
        if not idr:
            assert np.array_equal(sad, enc.sad()), "sad " + tag
        lv_g, lv_o = eng.debug(1), enc.levels()
        if not np.array_equal(lv_g, lv_o):
            bad = np.argwhere(lv_g != lv_o)
            raise AssertionError("levels %s first mismatch mb,blk,idx=%s gpu=%d orc=%d (%d diffs)" % (
                tag, bad[0], lv_g[tuple(bad[0])], lv_o[tuple(bad[0])], len(bad)))
        rec_o = enc.recon()
        for p in range(3):
            assert np.array_equal(eng.debug(4 + p), rec_o[p]), "recon plane %d %s" % (p, tag)
        # macroblock bit counts (the first MB of a slice also carries the slice header on the GPU)
        bits_g, bits_o = eng.debug(3).astype(np.int64), enc.mb_bits().astype(np.int64)
        first = set(int(r) * enc.mbw for r in enc.slice_rows()[:-1])
        idx = np.array([i for i in range(enc.nmb) if i not in first], dtype=np.int64)
        if not np.array_equal(bits_g[idx], bits_o[idx]):
            bad = idx[np.nonzero(bits_g[idx] != bits_o[idx])[0]]
            raise AssertionError("mb bits %s: first bad mb %d gpu %d orc %d (%d bad)" % (
                tag, bad[0], bits_g[bad[0]], bits_o[bad[0]], len(bad)))
        # slice RBSPs
        sl_g = eng.get_slices(t % 2)
        sl_o = [unescape(b) for (ty, _, b) in split_nals(ref.data) if ty in (1, 5)]
        assert len(sl_g) == len(sl_o), "slice count " + tag
        for s, (a, b) in enumerate(zip(sl_g, sl_o)):
            assert a == b, "slice %d rbsp %s (len gpu %d orc %d)" % (s, tag, len(a), len(b))
    eng.close()


@pytest.mark.parametrize("w,h,qp,slices,keyint,n,kind", [
    (352, 288, 26, 4, 4, 6, "synth"),      # CIF, I + P, multi-slice
    (352, 288, 26, 1, 1, 2, "synth"),      # config 1 shape: I frames only, single slice
    (176, 144, 12, 9, 3, 5, "noise"),      # minimum QP, one slice per MB row
    (176, 144, 51, 2, 3, 4, "noise"),      # maximum QP
    (178, 146, 30, 3, 3, 4, "noise"),      # not a multiple of 16 nor of 4: clamped ingest paths
    (640, 368, 38, 23, 2, 3, "synth"),     # chroma QP table region, many slices
    (1920, 1080, 26, 8, 2, 2, "synth"),    # BASELINE config 2 geometry
    (3840, 2160, 26, 8, 2, 2, "synth"),    # BASELINE config 3 / 5 geometry
    (16, 16, 30, 1, 2, 3, "noise"),        # one macroblock: every window piece but one lies outside the picture
    (144, 32, 24, 2, 3, 4, "noise"),       # one tile row, second tile column partly outside
    (130, 258, 36, 5, 3, 4, "noise"),      # 9 x 17 macroblocks: ragged right and bottom tiles, uneven slices
    (176, 464, 28, 1, 2, 3, "noise"),      # 29 MB rows in one slice: two passes of the intra DC chain (24 + 5 rows)
    (7680, 4320, 32, 8, 2, 2, "synth"),    # 8K: 480 x 270 macroblocks, 34-row slices (two passes of the intra DC chain), 60 tiles per row
    (1280, 720, 30, 45, 3, 3, "synth"),    # one slice per macroblock row: 45 blocks per item in k_scan, the last one writes the table
    (16, 256, 28, 4, 3, 4, "noise"),       # one macroblock wide: single-column wavefront, every search window piece left/right outside
    (352, 288, 51, 3, 3, 4, "synth"),      # maximum QP on smooth content: nearly everything P_Skip, long skip runs across slices
    (1920, 1088, 20, 8, 2, 2, "synth"),    # low QP (dense residual: every syntax unit of most macroblocks listed)
])
def test_stages_match_oracle(venc, orc, w, h, qp, slices, keyint, n, kind):
    _stage_compare(venc, orc, w, h, qp, slices, keyint, n, kind)


@pytest.mark.parametrize("w,h,qp,slices,keyint,n,kind", [
    (352, 288, 26, 4, 4, 4, "synth"),
    (1920, 1088, 20, 8, 2, 2, "synth"),    # dense residual: ~20 entries per macroblock
    (176, 144, 12, 1, 3, 3, "noise"),      # minimum QP, one slice: strings beyond 128 bits among them
])
def test_entropy_segments_beyond_their_lds_part(venc, legacy_lib, orc, w, h, qp, slices, keyint, n, kind):
    """k_entropy_p keeps the lengths of a segment's first ENT_LEN_LDS entries and the first ENT_CLS_LDS of its phase-B list in LDS
    and sends the rest through memory (lengths park in the upper half of the segment's offsets).  The product sizes both for
    ordinary pictures; the test-only library is built with 64 / 16, so every segment here runs mostly on the memory paths."""
    _stage_compare(venc, orc, w, h, qp, slices, keyint, n, kind, lib=legacy_lib)


def _x264_stream(venc, frames, w, h, **kw):
    e = venc.X264Encoder(w, h, **kw)
    out = []
    for t, (y, u, v) in enumerate(frames):
        r = e.encode(y, u, v, t)
        if r is not None:
            out.append(r)
    assert e.delayed_frames() == len(frames) - len(out)
    out += e.flush()
    assert e.delayed_frames() == 0
    e.close()
    return out


@pytest.mark.parametrize("w,h,qp,slices,keyint,n,gop_batch,tune", [
    (352, 288, 26, 8, 30, 30, 1, "film"),          # config 1/2 shape at CIF
    (352, 288, 20, 8, 5, 23, 1, "zerolatency"),    # reference default quality 20 (av_encode.c:104)
    (352, 288, 26, 8, 5, 23, 3, "film"),           # GOP-batched, partial last batch
    (320, 180, 33, 5, 4, 17, 2, "film"),           # height not a multiple of 16 (cropping)
])
def test_x264_api_bitstream_matches_oracle(venc, orc, w, h, qp, slices, keyint, n, gop_batch, tune):
    frames = [venc.synth_frame(w, h, t) for t in range(n)]
    enc = orc.Encoder(w, h, qp=qp, n_slices=slices, keyint=keyint, fps=(30, 1), sar=(1, 1), threads=8)
    dec = orc.Decoder()
    got = _x264_stream(venc, frames, w, h, qp=qp, slices=slices, keyint=keyint, fps=(30, 1), sar=(1, 1),
                       gop_batch=gop_batch, tune=tune)
    assert len(got) == n
    last_dts = None
    for t, (y, u, v) in enumerate(frames):
        ref = enc.encode(y, u, v)
        g = got[t]
        assert g.payload == ref.data, "frame %d bitstream differs (gpu %d bytes, oracle %d)" % (
            t, len(g.payload), len(ref.data))
        # boundary invariants of SURVEY.md section 8b
        assert g.keyframe == ref.is_idr and g.pts == t and g.dts == g.pts
        assert last_dts is None or g.dts > last_dts
        last_dts = g.dts
        off = 0
        for (ty, ref_idc, o, size) in g.nals:
            assert o == off, "NALs must be contiguous"
            assert int.from_bytes(g.payload[o:o + 4], "big") == size - 4
            off += size
        assert off == len(g.payload)
        if g.keyframe:
            assert [n_[0] for n_ in g.nals[:2]] == [7, 8]
            sps = g.payload[g.nals[0][2]:]
            assert sps[5] == 66 and sps[6] == 0xC0
        assert dec.decode(g.payload) == 1
        assert np.array_equal(dec.planes()[0], enc.recon()[0])


@pytest.mark.parametrize("gop_batch,depth", [(1, 2), (4, 2), (4, 3)])
def test_rotating_input_pictures_equal_the_single_picture_stream(venc, gop_batch, depth):
    """hip.i_input_depth = n: the caller fills n pictures in turn and x264_encoder_encode waits for the H2D copy issued
    n - 1 calls earlier instead of its own (include/x264.h).  Every picture is poisoned as soon as the contract lets the
    caller touch it again: the stream must still equal the one-picture stream."""
    w, h, n = 352, 288, 26
    frames = [venc.synth_frame(w, h, t, 7) for t in range(n)]
    a = _x264_stream(venc, frames, w, h, qp=28, slices=4, keyint=6, gop_batch=gop_batch)
    enc = venc.X264Encoder(w, h, qp=28, slices=4, keyint=6, gop_batch=gop_batch, input_depth=depth)
    got = []
    for t, (y, u, v) in enumerate(frames):
        if t >= depth:  # the picture about to be refilled was handed over `depth` calls ago: its copy must be done -- scribble over it first
            for pl in enc._pic_planes[t % depth]:
                pl[...] = 0xA5
        f = enc.encode(y, u, v, t)
        if f is not None:
            got.append(f)
    got += enc.flush()
    enc.close()
    assert [f.payload for f in got] == [f.payload for f in a]


@pytest.mark.parametrize("gop_batch,depth", [(1, 1), (3, 1), (3, 2)])
def test_host_frames_loop_in_c_equals_the_per_call_stream(venc, gop_batch, depth):
    """x264_venc_encode_host_frames -- the reference's loop (av_encode.c:543-547, :970, :1076-1083) in C over frames in pinned host
    memory, what bench.py's ingest-inclusive leg times -- delivers the frames the per-call Python loop delivers, in order, with the
    tally's per-pass delivery stamps set."""
    import ctypes as C
    w, h, n = 176, 144, 17
    frames = [venc.synth_frame(w, h, t, 11) for t in range(n)]
    a = _x264_stream(venc, frames, w, h, qp=29, slices=3, keyint=4, gop_batch=gop_batch)
    H, L = venc.hip_lib(), venc.x264_lib()
    fb = w * h * 3 // 2
    hp = H.venc_host_alloc(n * fb)
    assert hp
    try:
        host = np.ctypeslib.as_array((C.c_uint8 * (n * fb)).from_address(hp))
        for t, (y, u, v) in enumerate(frames):
            host[t * fb:(t + 1) * fb] = np.concatenate([y.ravel(), u.ravel(), v.ravel()])
        enc = venc.X264Encoder(w, h, qp=29, slices=3, keyint=4, gop_batch=gop_batch, input_depth=depth)
        got = []

        def sink(user, nals, n_nals, size, pic):
            got.append(C.string_at(nals[0].p_payload, size))
            return 0
        cb = venc.SINK_FN(sink)
        L.x264_venc_encode_host_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        assert L.x264_venc_encode_host_frames(enc.h, C.c_void_p(hp), fb, w, h, n, 1, C.cast(cb, C.c_void_p), None) == n
        assert got == [f.payload for f in a]
        enc.close()
        # two passes over the first 16 frames (whole GOPs) as one stream, counted by the C tally sink with its per-pass stamps
        enc = venc.X264Encoder(w, h, qp=29, slices=3, keyint=4, gop_batch=gop_batch, input_depth=depth)
        t = venc.X264Tally()
        t.pass_frames = 16
        assert L.x264_venc_encode_host_frames(enc.h, C.c_void_p(hp), fb, w, h, 16, 2, C.cast(L.x264_venc_tally_sink, C.c_void_p), C.cast(C.pointer(t), C.c_void_p)) == 32
        assert t.frames == 32 and t.passes == 2 and 0 < t.pass_end_ns[0] <= t.pass_end_ns[1]
        assert t.bytes == 2 * sum(len(f.payload) for f in a[:16])
        enc.close()
    finally:
        H.venc_host_free(hp)


def test_gop_batched_equals_streaming(venc):
    w, h, n = 352, 288, 40
    frames = [venc.synth_frame(w, h, t, 99) for t in range(n)]
    a = _x264_stream(venc, frames, w, h, qp=28, slices=6, keyint=8, gop_batch=1)
    b = _x264_stream(venc, frames, w, h, qp=28, slices=6, keyint=8, gop_batch=4)
    assert [f.payload for f in a] == [f.payload for f in b]


def test_sixty_four_items_in_lock_step(venc):
    """The engine's maximum: 64 closed GOPs advanced per step (ragged last unit: 70 GOPs = 64 + 6), against one at a time."""
    w, h, keyint, gops = 80, 48, 3, 70
    frames = [venc.synth_frame(w, h, t, 5) for t in range(keyint * gops)]
    a = _x264_stream(venc, frames, w, h, qp=30, slices=3, keyint=keyint, gop_batch=1)
    b = _x264_stream(venc, frames, w, h, qp=30, slices=3, keyint=keyint, gop_batch=64)
    assert len(a) == len(b) == keyint * gops
    assert [f.payload for f in a] == [f.payload for f in b]
    assert [f.pts for f in b] == list(range(keyint * gops))


def test_resident_full_size_roundtrip(venc, orc):
    """BASELINE config 2 geometry at full size: size-independent properties --
    own decoder reproduces a decodable stream, GOP-batched == streaming hash,
    and the device-side synthetic source equals the host one."""
    import torch
    w, h, n, keyint = 1920, 1080, 12, 6
    fb = w * h * 3 // 2
    buf = torch.empty(n * fb, dtype=torch.uint8, device="cuda")
    assert venc.hip_lib().venc_synth_device(buf.data_ptr(), fb, w, h, 0, n, 0x1234ABCD) == 0
    host = buf.cpu().numpy().reshape(n, fb)
    y0, u0, v0 = venc.synth_frame(w, h, 3)
    assert np.array_equal(host[3][:w * h].reshape(h, w), y0)
    assert np.array_equal(host[3][w * h:w * h * 5 // 4].reshape(h // 2, w // 2), u0)
    hashes = []
    for gb in (1, 2):
        e = venc.X264Encoder(w, h, qp=26, slices=8, keyint=keyint, gop_batch=gb)
        dec = orc.Decoder()
        md = hashlib.sha256()
        sizes = []

        def sink(payload, nals, pts, key, md=md, dec=dec, sizes=sizes):
            md.update(payload)
            sizes.append(len(payload))
            assert dec.decode(payload) == 1
            assert key == (pts % keyint == 0)
        assert e.encode_resident(buf.data_ptr(), fb, n, sink) == n
        assert dec.size() == (w, h)
        e.close()
        hashes.append(md.hexdigest())
    assert hashes[0] == hashes[1]


@pytest.mark.gpu
@pytest.mark.parametrize("gop_batch", [3, 1])
def test_resident_loop_equals_separate_calls(venc, gop_batch):
    """x264_venc_encode_resident_loop (what bench.py times): K passes of the resident clip as one stream give the
    frames of K separate calls, in order, although consecutive scheduling units overlap on the GPU."""
    import torch
    w, h, n, keyint, reps = 352, 288, 12, 4, 3
    fb = w * h * 3 // 2
    buf = torch.empty(n * fb, dtype=torch.uint8, device="cuda")
    assert venc.hip_lib().venc_synth_device(buf.data_ptr(), fb, w, h, 0, n, 0x1234ABCD) == 0
    a, b = [], []
    e = venc.X264Encoder(w, h, qp=28, slices=4, keyint=keyint, gop_batch=gop_batch)
    for _ in range(reps):
        assert e.encode_resident(buf.data_ptr(), fb, n, lambda payload, nals, pts, key: a.append((payload, pts % n, key))) == n
    e.close()
    e = venc.X264Encoder(w, h, qp=28, slices=4, keyint=keyint, gop_batch=gop_batch)
    assert e.encode_resident(buf.data_ptr(), fb, n, lambda payload, nals, pts, key: b.append((payload, pts % n, key)),
                             repeats=reps) == n * reps
    assert [p for _, p, _ in b] == [t % n for t in range(n * reps)]
    e.close()
    e = venc.X264Encoder(w, h, qp=28, slices=4, keyint=keyint, gop_batch=gop_batch)
    tally = e.encode_resident_tally(buf.data_ptr(), fb, n, repeats=reps)
    assert tally.frames == n * reps and tally.bytes == sum(len(f[0]) for f in a)
    e.close()
    assert len(a) == len(b) == n * reps
    for t, (x, y) in enumerate(zip(a, b)):
        assert x == y, "frame %d" % t


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,qp,slices", [(352, 288, 26, 4), (176, 144, 51, 2), (178, 146, 12, 3), (176, 464, 30, 1)])
def test_intra_three_kernels_equal_the_wavefront_kernel(venc, legacy_lib, w, h, qp, slices, monkeypatch):
    """I pictures: transform / DC chain / reconstruction as three kernels (venc_intra2.hip) against the one-kernel
    wavefront k_intra (VENC_INTRA=legacy, read when an engine is opened): same reconstruction, levels and macroblock
    records.  QP 51 noise is the case that exposed the v_ashr_pk_u8_i32 selection."""
    frames = noise_frames(w, h, 4, 0x1234ABCD)
    monkeypatch.setenv("VENC_INTRA", "legacy")
    ea = venc.Engine(w, h, qp=qp, n_slices=slices, max_items=1, in_slots=2, out_slots=2, lib=legacy_lib)
    monkeypatch.delenv("VENC_INTRA")
    eb = venc.Engine(w, h, qp=qp, n_slices=slices, max_items=1, in_slots=2, out_slots=2)
    for t, (y, u, v) in enumerate(frames):
        for e in (ea, eb):
            e.upload(t % 2, y, u, v)
            e.encode_step([t % 2], [t % 2], True, 0, [t & 1])
            e.sync()
        for p in range(3):
            assert np.array_equal(ea.debug(4 + p), eb.debug(4 + p)), "plane %d frame %d" % (p, t)
        assert np.array_equal(ea.debug(1), eb.debug(1)), "levels frame %d" % t
        assert np.array_equal(ea.debug(2)[:, :27], eb.debug(2)[:, :27]), "macroblock records frame %d" % t
        assert ea.get_slices(t % 2) == eb.get_slices(t % 2), "slices frame %d" % t
    ea.close()
    eb.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [
    {"VENC_FUSE": "0"},                 # luma + chroma transform as k_inter_fused on the main chain (round 1's shape)
    {"VENC_FUSE": "1"},                 # every luma block coded inside the search kernels (round 2's shape; default: blocks with a level behind them)
    {"VENC_GOP_HINT": "0"},             # a GOP's first P picture starts from the zero vector only (no hint from the previous unit)
    {"VENC_ME_BLOCKS": "3"},            # the search on three blocks per CU (k_me_pool) with the deferred transform
    {"VENC_ME_BLOCKS": "4"},            # ... on four (k_me_pool4), whatever the step size
    {"VENC_NZ_RPB": "4"},               # one follow-up block per search tile: many blocks, every share small
    {"VENC_ENTROPY_BATCH": "1"},        # entropy chain per step
    {"VENC_ENTROPY_BATCH": "2"},
    {"VENC_INTRA_AHEAD": "0"},          # I steps on the main stream
    {"VENC_STAGED": "0"},               # k_pack stores to pinned host memory
    {"VENC_STAGED": "1", "VENC_HOST_THREADS": "1"},   # staged output, no packaging thread
    {"VENC_NO_OVERLAP": "1"},           # one stream
    {"VENC_ME_LDS_PAD": "28000"},       # two search blocks per CU
    {"VENC_CHAINS": "2"},               # round 4: the items of a P step as two search -> fix chains on two streams
    {"VENC_CHAINS": "2", "VENC_INTRA_AHEAD": "0"},   # ... with intra steps on the main stream (the chains resynchronise around them)
    {"VENC_PRIO": "1"},                 # main chain at the highest stream priority, side chain at the lowest
    {"VENC_HOST_THREADS": "2"},         # packaging thread without helpers
    {"VENC_PACK_HEAVY_KB": "1"},        # every step's frames shared between the packaging thread and its helper
])
def test_launcher_variants_change_nothing(venc, orc, monkeypatch, env):
    """Every scheduling variant of the launcher -- where the transforms run, how many steps an entropy
    launch covers, where I steps run, how the bitstream leaves the GPU, how many host threads -- must give the default's
    stream, which equals the oracle's.  Three passes of a clip with short GOPs in lock-step, so that units overlap,
    batches wrap around the ring and intra steps run ahead of their unit."""
    import torch
    w, h, n, keyint, reps = 352, 288, 24, 6, 3
    fb = w * h * 3 // 2
    buf = torch.empty(n * fb, dtype=torch.uint8, device="cuda")
    assert venc.hip_lib().venc_synth_device(buf.data_ptr(), fb, w, h, 0, n, 0x1234ABCD) == 0
    outs = []
    for e_ in (None, env):
        for k, v in (e_ or {}).items():
            monkeypatch.setenv(k, v)
        e = venc.X264Encoder(w, h, qp=27, slices=4, keyint=keyint, gop_batch=2)
        got = []
        assert e.encode_resident(buf.data_ptr(), fb, n, lambda payload, nals, pts, key: got.append(payload), repeats=reps) == n * reps
        e.close()
        outs.append(got)
    assert outs[0] == outs[1]
    assert outs[0][:n] == outs[0][n:2 * n] == outs[0][2 * n:]
    host = buf.cpu().numpy().reshape(n, fb)
    ref = orc.Encoder(w, h, qp=27, n_slices=4, keyint=keyint, threads=8)
    for t in range(n):
        f = host[t]
        assert outs[0][t] == ref.encode(f[:w * h].reshape(h, w), f[w * h:w * h * 5 // 4].reshape(h // 2, w // 2),
                                        f[w * h * 5 // 4:].reshape(h // 2, w // 2)).data, "frame %d vs oracle" % t


@pytest.mark.gpu
def test_placement_knobs_change_nothing(venc, monkeypatch):
    """VENC_ENTROPY_CUS (the entropy chain's stream confined to a subset of the CUs) and VENC_ME_XCD (XCD-aware tile order
    of the motion search), experiments kept behind environment variables, must not change the stream."""
    import torch
    w, h, n, keyint = 352, 288, 8, 4
    fb = w * h * 3 // 2
    buf = torch.empty(n * fb, dtype=torch.uint8, device="cuda")
    assert venc.hip_lib().venc_synth_device(buf.data_ptr(), fb, w, h, 0, n, 0x1234ABCD) == 0
    outs = []
    for cus in (None, "64"):
        if cus:
            monkeypatch.setenv("VENC_ENTROPY_CUS", cus)
            monkeypatch.setenv("VENC_ME_XCD", "1")
        e = venc.X264Encoder(w, h, qp=28, slices=4, keyint=keyint, gop_batch=2)
        got = []
        assert e.encode_resident(buf.data_ptr(), fb, n, lambda payload, nals, pts, key: got.append(payload)) == n
        e.close()
        outs.append(got)
    assert outs[0] == outs[1]


def _me_run(venc, frames, w, h, mode, lib=None):
    import os
    os.environ["VENC_ME"] = mode
    try:
        eng = venc.Engine(w, h, qp=26, n_slices=4, max_items=1, in_slots=2, out_slots=2, lib=lib)
    finally:
        os.environ.pop("VENC_ME", None)
    out = []
    for t, (y, u, v) in enumerate(frames):
        eng.upload(t % 2, y, u, v)
        eng.encode_step([t % 2], [t % 2], t == 0, t & 15, [0])
        eng.sync()
        out.append((eng.debug(0).copy(), eng.get_slices(t % 2)))
    eng.close()
    return out


@pytest.mark.parametrize("kind", ["synth", "flat", "white_noise", "static_noise", "stripes"])
def test_successive_elimination_equals_exhaustive_search(venc, legacy_lib, orc, kind):
    """The pruned search (k_me) must return the same vectors and SADs as the exhaustive kernel
    (k_me_dense) and the oracle, including on content where the bound prunes nothing (flat,
    white noise: dense fallback) and where ties abound (flat, stripes: rank tie-break)."""
    w, h, n = 352, 288, 4
    rng = np.random.default_rng(4)
    if kind == "synth":
        frames = [venc.synth_frame(w, h, t, 321) for t in range(n)]
    elif kind == "flat":
        frames = [(np.full((h, w), 80 + 3 * t, np.uint8), np.full((h // 2, w // 2), 128, np.uint8),
                   np.full((h // 2, w // 2), 120, np.uint8)) for t in range(n)]
    elif kind == "white_noise":
        frames = [tuple(rng.integers(0, 256, s).astype(np.uint8) for s in ((h, w), (h // 2, w // 2), (h // 2, w // 2)))
                  for _ in range(n)]
    elif kind == "static_noise":
        base = tuple(rng.integers(0, 256, s).astype(np.uint8) for s in ((h, w), (h // 2, w // 2), (h // 2, w // 2)))
        frames = [base] * n
    else:
        yy, xx = np.mgrid[0:h, 0:w]
        frames = [((((xx + 2 * t) // 4) % 2) * 200 + 20).astype(np.uint8) for t in range(n)]
        frames = [(f, np.full((h // 2, w // 2), 128, np.uint8), np.full((h // 2, w // 2), 128, np.uint8)) for f in frames]
    a = _me_run(venc, frames, w, h, "pool")      # the product library: k_me_pool (pooled survivors) + k_me_fix
    d = _me_run(venc, frames, w, h, "tile", legacy_lib)   # its predecessor k_me_tile (test-only library)
    b = _me_run(venc, frames, w, h, "dense", legacy_lib)  # exhaustive
    c = _me_run(venc, frames, w, h, "plane", legacy_lib)  # sums from the k_refsum plane, in-kernel dense fallback
    enc = orc.Encoder(w, h, qp=26, n_slices=4, keyint=100, threads=8)
    for t, (y, u, v) in enumerate(frames):
        enc.encode(y, u, v)
        assert np.array_equal(a[t][0], b[t][0]), "%s frame %d: pruned (tile) != exhaustive" % (kind, t)
        assert np.array_equal(c[t][0], b[t][0]), "%s frame %d: pruned (plane) != exhaustive" % (kind, t)
        assert np.array_equal(d[t][0], b[t][0]), "%s frame %d: pruned (k_me_tile) != exhaustive" % (kind, t)
        assert a[t][1] == b[t][1] and c[t][1] == b[t][1] and d[t][1] == b[t][1]
        if t:
            mv = np.stack([(a[t][0] & 0xff).astype(np.int8), ((a[t][0] >> 8) & 0xff).astype(np.int8)], axis=1)
            assert np.array_equal(mv, enc.mv()), "%s frame %d: mv != oracle" % (kind, t)
            assert np.array_equal((a[t][0] >> 16).astype(np.uint16), enc.sad())


def test_reference_block_sums_plane(venc, legacy_lib):
    """k_refsum: P8(x,y) = S8(x,y) | S8(x+8,y) << 16 over the edge-replicated reconstruction."""
    w, h = 352, 288
    import os
    os.environ["VENC_ME"] = "plane"
    try:
        eng = venc.Engine(w, h, qp=30, n_slices=2, max_items=1, in_slots=1, out_slots=1, lib=legacy_lib)
    finally:
        os.environ.pop("VENC_ME", None)
    eng.upload(0, *venc.synth_frame(w, h, 0, 9))
    eng.encode_step([0], [0], True, 0, [0])
    eng.sync()
    rec = eng.debug(4).astype(np.int64)
    p8 = eng.debug(8)
    ch, cw = rec.shape
    pad = np.pad(rec, ((16, 32), (16, 32)), mode="edge")
    c = np.cumsum(np.cumsum(np.pad(pad, ((1, 0), (1, 0))), 0), 1)
    s8 = c[8:, 8:] - c[:-8, 8:] - c[8:, :-8] + c[:-8, :-8]          # s8[Y][X] = block sum at picture (X-16, Y-16)
    H, W = ch + 25, cw + 17
    want = (s8[:H, :W] | (s8[:H, 8:W + 8] << 16)).astype(np.uint32)
    assert np.array_equal(p8[:H, :W], want)
    eng.close()


def test_legacy_and_split_inter_kernels_agree(venc, legacy_lib):
    """k_inter_luma + k_inter_chroma (homogeneous waves) against the first k_inter."""
    import os
    w, h, n = 352, 288, 4
    frames = [venc.synth_frame(w, h, t, 55) for t in range(n)]
    outs = []
    for mode in ("legacy", "split"):
        os.environ["VENC_INTER"] = mode
        try:
            eng = venc.Engine(w, h, qp=24, n_slices=3, max_items=1, in_slots=2, out_slots=2, lib=legacy_lib if mode == "legacy" else None)
        finally:
            os.environ.pop("VENC_INTER", None)
        res = []
        for t, (y, u, v) in enumerate(frames):
            eng.upload(t % 2, y, u, v)
            eng.encode_step([t % 2], [t % 2], t == 0, t & 15, [0])
            eng.sync()
            res.append((eng.debug(1), eng.debug(2)[:, :27], eng.debug(4), eng.debug(5), eng.debug(6), eng.get_slices(t % 2)))
        eng.close()
        outs.append(res)
    for t in range(n):
        for k in range(5):
            assert np.array_equal(outs[0][t][k], outs[1][t][k]), "frame %d output %d" % (t, k)
        assert outs[0][t][5] == outs[1][t][5]


@pytest.mark.parametrize("tune", ["zerolatency", "film"])  # film = the reference's default (av_encode.c:103): ten closed GOPs in lock-step, drained through the flush loop
def test_c_driver_y4m_in_mp4_out(venc, orc, tmp_path, tune):
    """tools/encode_yuv: the reference's call sequence in C, YUV4MPEG2 in (include/venc_source.h), MP4 out
    (include/venc_mp4.h).  The samples in the file equal the oracle's slice NALs, the avcC parameter sets equal the
    oracle's, and the demultiplexed stream decodes to the oracle's reconstruction."""
    import json
    import os
    import struct
    import subprocess
    from test_io import read_track, samples_of, _write_y4m
    w, h, n, keyint, slices = 176, 144, 7, 3, 3
    frames = [venc.synth_frame(w, h, t, 99) for t in range(n)]
    y4m, mp4, es = str(tmp_path / "in.y4m"), str(tmp_path / "out.mp4"), str(tmp_path / "out.h264")
    _write_y4m(y4m, frames, w, h, header_extra="F24:1 Ip A1:1 C420mpeg2")
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "encode_yuv")
    out = subprocess.check_output([tool, "-i", y4m, "-n", "100", "-q", "27", "-S", str(slices), "-k", str(keyint), "-t", tune,
                                   "-o", es, "-m", mp4])
    rep = json.loads(out.decode().strip().splitlines()[-1])
    assert rep["frames"] == n and rep["violations"] == 0 and rep["sync_samples"] == 3
    data = open(mp4, "rb").read()
    t = read_track(data)
    assert t["timescale"] == 24 and t["stts"] == [(n, 1)] and t["stss"] == [1, 4, 7] and (t["width"], t["height"]) == (w, h)
    # the C driver opens the encoder as the reference does: X264_RC_CRF (av_encode.c:400-401) = spec/CODEC.md section 1a
    # ... and with the preset's intra modes (x264_param_default_preset: Intra4x4 tried against Intra16x16 V / H / DC / plane)
    # ... and its sub-sample refinement (every preset but ultrafast: half- then quarter-sample stage)
    enc = orc.Encoder(w, h, qp=27, n_slices=slices, keyint=keyint, fps=(24, 1), sar=(1, 1), threads=4, rc=True, intra_modes=3, subpel=2, deblock=True)
    dec = orc.Decoder()
    got = samples_of(data, t)
    for k, f in enumerate(frames):
        ref = enc.encode(*f)
        first = min(off for (ty, _, off, _) in ref.nals if ty not in (7, 8))
        assert got[k] == ref.data[first:], "sample %d differs from the oracle's slices" % k
        if k == 0:
            sps = [bytes([0x67]) + b for (ty, _, b) in split_nals(ref.data) if ty == 7]
            pps = [bytes([0x68]) + b for (ty, _, b) in split_nals(ref.data) if ty == 8]
            assert t["sps"] == sps and t["pps"] == pps
            assert dec.decode(b"".join(struct.pack(">I", len(x)) + x for x in sps + pps)) == 0
        assert dec.decode(got[k]) == 1
        assert np.array_equal(dec.planes()[0], enc.recon()[0])
    # the reference's own command line (av_encode.c:89-196): long options, "input output", --frame-limit enforced
    mp4b = str(tmp_path / "ref_cli.mp4")
    out = subprocess.check_output([tool, "--silent", "--frame-limit", "5", "--preset", "fast", "--tune", "zerolatency", "--quality", "27",
                                   "--filters", "yadif", y4m, mp4b], stderr=subprocess.STDOUT).decode()
    assert "frame_limit: 5" in out and "preset: fast" in out and "--filters belongs to" in out
    tb = read_track(open(mp4b, "rb").read())
    assert len(tb["stsz"]) == 5 and tb["stss"] == [1] and tb["timescale"] == 24  # default keyint 30: one GOP
    assert subprocess.call([tool, "--preset", "warp9", y4m, mp4b], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL) == 6
    assert subprocess.call([tool, "--nonsense", y4m, mp4b], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL) == 1


def test_baseline_config2_whole_clip_properties(venc, orc):
    """BASELINE config 2 at full size (1920x1080, 300 frames, GOP 30, 8 slices, QP 26), where the oracle is too slow to
    encode along: size-independent properties.  The lock-step schedule does not change the stream (10 GOPs at a time ==
    3 at a time, a digest of all frame digests), every frame decodes with the oracle's decoder, decoded pictures stay
    close to the input (PSNR) and the stream equals the one bench.py measures."""
    import hashlib
    import torch
    w, h, n, keyint = 1920, 1080, 300, 30
    fb = w * h * 3 // 2
    buf = torch.empty(n * fb, dtype=torch.uint8, device="cuda")
    assert venc.hip_lib().venc_synth_device(buf.data_ptr(), fb, w, h, 0, n, 0x1234ABCD) == 0
    digests = []
    for gb in (10, 3):
        e = venc.X264Encoder(w, h, qp=26, slices=8, keyint=keyint, fps=(30, 1), sar=(1, 1), gop_batch=gb)
        md, frames = hashlib.sha256(), []

        def sink(payload, nals, pts, key, md=md, frames=frames):
            md.update(hashlib.sha256(payload).digest())
            frames.append((payload, pts, key))
        assert e.encode_resident(buf.data_ptr(), fb, n, sink) == n
        if gb == 10:
            tally = e.encode_resident_tally(buf.data_ptr(), fb, n)
            assert tally.frames == n and tally.keyframes == n // keyint
            assert tally.bytes == sum(len(f[0]) for f in frames)
        e.close()
        digests.append(md.hexdigest())
        if gb == 10:
            kept = frames
    assert digests[0] == digests[1]
    assert [f[1] for f in kept] == list(range(n)) and [f[2] for f in kept] == [t % keyint == 0 for t in range(n)]
    dec = orc.Decoder()
    host = buf.cpu().numpy().reshape(n, fb)
    worst = 99.0
    for t, (payload, _, _) in enumerate(kept):
        assert dec.decode(payload) == 1, "frame %d" % t
        if t % 29 == 0:
            y = dec.planes()[0][:h, :w].astype(np.float64)
            mse = np.mean((y - host[t][:w * h].reshape(h, w)) ** 2)
            worst = min(worst, 10 * np.log10(255.0 ** 2 / max(mse, 1e-9)))
    assert worst > 33.0, worst


@pytest.mark.parametrize("kind", ["flat", "static_noise", "half_static", "pan"])
def test_p_skip_runs_match_oracle(venc, orc, kind):
    """P_Skip (7.3.4 / 8.4.1.1): whole slices of skipped macroblocks, runs that cross macroblock rows, runs that end a
    slice (trailing mb_skip_run) and runs that end in a coded macroblock -- bitstream and decoded pictures against the
    oracle, and the share of skipped macroblocks is what the content implies."""
    w, h, n, keyint, slices = 208, 160, 5, 5, 3
    rng = np.random.default_rng(12)
    base = tuple(rng.integers(0, 256, sh).astype(np.uint8) for sh in ((h, w), (h // 2, w // 2), (h // 2, w // 2)))
    if kind == "flat":
        frames = [(np.full((h, w), 90, np.uint8), np.full((h // 2, w // 2), 128, np.uint8), np.full((h // 2, w // 2), 128, np.uint8))] * n
    elif kind == "static_noise":
        frames = [base] * n
    elif kind == "half_static":  # left half frozen, right half fresh noise every frame: runs end in coded macroblocks
        frames = []
        for t in range(n):
            f = tuple(p.copy() for p in base)
            f[0][:, w // 2:] = rng.integers(0, 256, (h, w - w // 2))
            f[1][:, w // 4:] = rng.integers(0, 256, (h // 2, w // 2 - w // 4))
            frames.append(f)
    else:  # smooth texture panning by (2, 0) per frame: skipped macroblocks with a non-zero inferred vector
        frames = noise_frames(w, h, n, seed=21)
    got = _x264_stream(venc, frames, w, h, qp=30, slices=slices, keyint=keyint, tune="zerolatency")
    enc = orc.Encoder(w, h, qp=30, n_slices=slices, keyint=keyint)
    dec = orc.Decoder()
    skipped = 0
    for t, f in enumerate(frames):
        ref = enc.encode(*f)
        assert got[t].payload == ref.data, "%s frame %d" % (kind, t)
        assert dec.decode(got[t].payload) == 1
        for p in range(3):
            assert np.array_equal(dec.planes()[p], enc.recon()[p])
        if t:
            bits = enc.mb_bits()
            last = set(int(r) * enc.mbw - 1 for r in enc.slice_rows()[1:])
            skipped += sum(1 for i, b in enumerate(bits) if b == 0 and i not in last)
    nmb_p = (n - 1) * enc.nmb
    if kind in ("flat", "static_noise"):
        assert skipped >= 0.85 * nmb_p  # (the I picture does not reconstruct its source exactly: some residual survives)
        if kind == "flat":
            assert all(len(g.payload) < 64 for g in got[1:])
    elif kind == "half_static":
        assert 0.3 * nmb_p < skipped < 0.6 * nmb_p
    else:
        assert skipped > 0


def test_av_encode_reference_command_line_and_progress_line(venc, orc, tmp_path):
    """tools/av_encode: the reference's command line with its SHORT options (av_encode.c:127 "sdv:a:l:f:"), on a
    terminal-like stdin (the reference's default for --silent is !isatty(stdin), av_encode.c:92), so the progress line
    of av_encode.c:1032-1069 is printed (video half: position, percentage, time left).  --frame-limit is enforced; the
    file's samples equal the oracle's slices."""
    import os
    import pty
    import re
    import subprocess
    from test_io import read_track, samples_of, _write_y4m
    w, h, n = 176, 144, 12
    frames = [venc.synth_frame(w, h, t, 7) for t in range(n)]
    y4m, mp4 = str(tmp_path / "in.y4m"), str(tmp_path / "out.mp4")
    _write_y4m(y4m, frames, w, h, header_extra="F25:1 Ip A1:1 C420jpeg")
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "av_encode")
    env = dict(os.environ, VENC_PROGRESS_INTERVAL="0", VENC_SLICES="3", VENC_KEYINT="5")

    def run_on_terminal(args):
        """stdin = a pseudo-terminal where the box has pty devices; else the driver's stand-in VENC_ASSUME_TTY"""
        try:
            master, slave = pty.openpty()
        except OSError:
            return subprocess.run(args, stdin=subprocess.DEVNULL, capture_output=True, text=True, env=dict(env, VENC_ASSUME_TTY="1"))
        try:
            return subprocess.run(args, stdin=slave, capture_output=True, text=True, env=env)
        finally:
            os.close(master)
            os.close(slave)
    r = run_on_terminal([tool, "-d", "-l", "9", "-v", "0", "--tune", "zerolatency", "--quality", "27.4", y4m, mp4])
    assert r.returncode == 0, r.stderr
    assert "silent: 0" in r.stdout and "frame_limit: 9" in r.stdout and "video_stream_index: 0" in r.stdout and "quality: 27.400000" in r.stdout
    lines = re.findall(r"video: (\d+):(\d\d):(\d\d) \((\d+\.\d)%\) - time left: \d+:\d\d:\d\d", r.stdout)
    assert len(lines) >= 8, r.stdout                       # one per frame at interval 0 (-d: each on its own line)
    pct = [float(x[3]) for x in lines]
    assert pct == sorted(pct) and pct[-1] <= 100.0 and pct[-1] > 50.0
    t = read_track(open(mp4, "rb").read())
    assert len(t["stsz"]) == 9 and t["stss"] == [1, 6] and t["timescale"] == 25
    enc = orc.Encoder(w, h, qp=27, n_slices=3, keyint=5, fps=(25, 1), sar=(1, 1), threads=4, rc=True, intra_modes=3, subpel=2, deblock=True)
    got = samples_of(open(mp4, "rb").read(), t)
    for k in range(9):
        ref = enc.encode(*frames[k])
        first = min(off for (ty, _, off, _) in ref.nals if ty not in (7, 8))
        assert got[k] == ref.data[first:], "sample %d" % k
    # -s: silent, no progress line
    r = run_on_terminal([tool, "-s", "-l", "3", y4m, mp4])
    assert r.returncode == 0 and "video:" not in r.stdout and "silent: 1" in r.stdout


def _varied_clip(w, h, n, seed):
    """Complexity that moves: smooth panning texture, then frames with fresh noise mixed in, then a still picture."""
    rng = np.random.default_rng(seed)
    base = noise_frames(w, h, n, seed=seed)
    out = []
    for t, (y, u, v) in enumerate(base):
        if n // 3 <= t < 2 * n // 3:
            amp = 10 * (t - n // 3 + 1)
            y = np.clip(y.astype(np.int32) + rng.integers(-amp, amp + 1, y.shape), 0, 255).astype(np.uint8)
        elif t >= 2 * n // 3:
            y, u, v = base[2 * n // 3]
        out.append((y, u, v))
    return out


@pytest.mark.parametrize("w,h,crf,slices,keyint,n,gop_batch", [
    (352, 288, 24, 4, 6, 18, 1),       # streaming
    (352, 288, 20, 4, 6, 24, 2),       # two GOPs in lock-step: per-item QPs differ within a launch; partial last unit
    (176, 144, 13, 2, 4, 9, 1),        # IDR QP clipped at 12
    (1920, 1080, 26, 8, 3, 5, 1),      # config 2 geometry
])
def test_crf_rate_control_matches_oracle(venc, orc, w, h, crf, slices, keyint, n, gop_batch):
    """X264_RC_CRF -- what the reference asks for (av_encode.c:400-401) -- as spec/CODEC.md section 1a defines it: IDR
    pictures at base - 3, P pictures at base + delta from the previous picture's SAD sum, decided on the GPU (every block
    of the search derives the QP alike; k_me_fix corrects the sum for the macroblocks it redoes), slice_qp_delta in every
    slice header.  Stream equal to the oracle's, picture QPs really move, own decoder follows."""
    frames = _varied_clip(w, h, n, seed=crf)
    kw = dict(slices=slices, keyint=keyint, fps=(30, 1), sar=(1, 1), crf=crf, gop_batch=gop_batch)
    if gop_batch == 1:
        kw["tune"] = "zerolatency"
    got = _x264_stream(venc, frames, w, h, **kw)
    enc = orc.Encoder(w, h, qp=crf, n_slices=slices, keyint=keyint, fps=(30, 1), sar=(1, 1), threads=8, rc=True)
    dec = orc.Decoder()
    qps = []
    for t, f in enumerate(frames):
        ref = enc.encode(*f)
        qps.append(enc.picture_qp())
        assert got[t].payload == ref.data, "frame %d (oracle QP %d): %d vs %d bytes" % (t, qps[-1], len(got[t].payload), len(ref.data))
        assert dec.decode(got[t].payload) == 1
        assert np.array_equal(dec.planes()[0], enc.recon()[0])
    assert qps[0] == max(12, crf - 3) and qps[1] == crf
    if n >= 9 and w < 1000:
        assert len(set(q for t, q in enumerate(qps) if t % keyint >= 2)) >= 2, qps   # the P pictures' QP follows the content


def test_rate_control_needs_the_whole_picture(venc):
    e = venc.Engine(176, 144, qp=30, n_slices=3, max_items=1, in_slots=2, out_slots=2, rc=True)
    with pytest.raises(venc.VencError):
        e.set_band(0, 2)
    e.close()


def _stripes_and_texture(venc, w, h, n):
    """Content on which the three predictors win in turn: vertical stripes (vertical), horizontal stripes (horizontal),
    smooth texture (DC), changing from region to region."""
    yy, xx = np.mgrid[0:h, 0:w]
    out = []
    for t in range(n):
        y, u, v = venc.synth_frame(w, h, t, 11)
        y = y.copy()
        y[:, : w // 3] = ((((xx[:, : w // 3] + t) // 3) % 2) * 150 + 40).astype(np.uint8)
        y[:, w // 3: 2 * w // 3] = ((((yy[:, w // 3: 2 * w // 3] + 2 * t) // 5) % 2) * 120 + 60).astype(np.uint8)
        out.append((y, u, v))
    return out


@pytest.mark.parametrize("w,h,qp,slices,n", [
    (352, 288, 26, 4, 3),
    (178, 146, 30, 3, 3),        # clamped ingest, ragged macroblocks
    (176, 464, 28, 1, 2),        # 29 macroblock rows in one slice: second chunk reads its top row from memory
    (1920, 1080, 26, 8, 2),
    (16, 16, 24, 1, 2),          # one macroblock: only DC is available
])
def test_intra16x16_vertical_horizontal_dc_match_oracle(venc, orc, w, h, qp, slices, n):
    """Intra16x16 prediction modes (spec/CODEC.md section 3): the engine's wavefront kernel
    k_intra_modes against the oracle, stage by stage -- levels, reconstruction, slice RBSPs (mb_type carries the mode) --
    on content where each of the three predictors wins somewhere; all-I."""
    enc = orc.Encoder(w, h, qp=qp, n_slices=slices, keyint=1, threads=8, intra_modes=True)
    eng = venc.Engine(w, h, qp=qp, n_slices=slices, max_items=1, in_slots=2, out_slots=2, intra_modes=True)
    modes = set()
    for t, (y, u, v) in enumerate(_stripes_and_texture(venc, w, h, n)):
        ref = enc.encode(y, u, v)
        eng.upload(t % 2, y, u, v)
        eng.encode_step([t % 2], [t % 2], True, 0, [t & 1])
        eng.sync()
        lv_g, lv_o = eng.debug(1), enc.levels()
        if not np.array_equal(lv_g, lv_o):
            bad = np.argwhere(lv_g != lv_o)
            raise AssertionError("levels frame %d: first mismatch mb,blk,idx=%s gpu=%d orc=%d (%d diffs)" % (
                t, bad[0], lv_g[tuple(bad[0])], lv_o[tuple(bad[0])], len(bad)))
        rec_o = enc.recon()
        for p in range(3):
            assert np.array_equal(eng.debug(4 + p), rec_o[p]), "recon plane %d frame %d" % (p, t)
        sl_o = [unescape(b) for (ty, _, b) in split_nals(ref.data) if ty in (1, 5)]
        assert eng.get_slices(t % 2) == sl_o, "slices frame %d" % t
        modes |= set(int(x) for x in (eng.debug(2)[:, 25] >> 1) & 3)
    if w >= 176 and h >= 144:
        assert modes == {0, 1, 2}, modes     # every predictor was chosen somewhere
    eng.close()


@pytest.mark.parametrize("w,h,qp,slices", [(176, 144, 26, 2), (352, 288, 30, 3), (178, 146, 22, 1), (1920, 1080, 28, 8)])
def test_intra16x16_plane_prediction_matches_oracle(venc, orc, w, h, qp, slices):
    """Intra16x16 plane prediction as the fourth candidate (intra_modes = 2, spec/CODEC.md section 3): smooth gradients with a
    little texture, where the plane predictor wins in places; levels, reconstruction and slice RBSPs against the oracle; all-I"""
    enc = orc.Encoder(w, h, qp=qp, n_slices=slices, keyint=1, threads=8, intra_modes=2)
    eng = venc.Engine(w, h, qp=qp, n_slices=slices, max_items=1, in_slots=2, out_slots=2, intra_modes=2)
    yy, xx = np.mgrid[0:h, 0:w]
    modes = set()
    for t in range(3):
        tex = venc.synth_frame(w, h, t, 21)
        y = (40 + (1.1 + 0.2 * t) * xx * 200.0 / w + 0.8 * yy * 160.0 / h + 12 * np.sin((xx + 2 * yy + 5 * t) / 11.0)).clip(0, 255).astype(np.uint8)
        y[:, w // 2:] = tex[0][:, w // 2:]      # half gradient, half texture
        u, v = tex[1], tex[2]
        ref = enc.encode(y, u, v)
        eng.upload(t % 2, y, u, v)
        eng.encode_step([t % 2], [t % 2], True, 0, [t & 1])
        eng.sync()
        lv_g, lv_o = eng.debug(1), enc.levels()
        if not np.array_equal(lv_g, lv_o):
            bad = np.argwhere(lv_g != lv_o)
            raise AssertionError("levels frame %d: first mismatch mb,blk,idx=%s gpu=%d orc=%d (%d diffs)" % (
                t, bad[0], lv_g[tuple(bad[0])], lv_o[tuple(bad[0])], len(bad)))
        for p in range(3):
            assert np.array_equal(eng.debug(4 + p), enc.recon()[p]), "recon plane %d frame %d" % (p, t)
        sl_o = [unescape(b) for (ty, _, b) in split_nals(ref.data) if ty in (1, 5)]
        assert eng.get_slices(t % 2) == sl_o, "slices frame %d" % t
        modes |= set(int(x) for x in (eng.debug(2)[:, 25] >> 1) & 3)
    assert 3 in modes, modes
    eng.close()


@pytest.mark.parametrize("w,h,qp,slices,n", [
    (16, 16, 24, 1, 2),          # one macroblock: no neighbour at all
    (64, 48, 30, 2, 3),
    (352, 288, 26, 4, 3),
    (178, 146, 30, 3, 3),        # clamped ingest, ragged macroblocks
    (176, 464, 28, 1, 2),        # 29 macroblock rows in one slice: the second chunk reads its top row (and the four samples beyond) from memory
    (1920, 1080, 26, 8, 2),
    (352, 288, 44, 1, 2),        # coarse quantiser: most blocks empty, cbp bits per 8x8
])
def test_intra4x4_matches_oracle(venc, orc, w, h, qp, slices, n):
    """Intra4x4 tried against the best Intra16x16 predictor (intra_modes = 3, spec/CODEC.md section 3):
    the engine's k_intra_modes<true> -- sixteen blocks per macroblock in decoding order, nine predictors each, rows two macroblocks
    apart because block 5 predicts from the macroblock above-right -- against the oracle, stage by stage: levels, reconstruction,
    slice RBSPs (I_NxN syntax: the modes against their predictions, the Intra column of table 9-4); all-I.  Conformance of the tool
    itself is UNPINNED (tests/test_oracle_intra4.py says what pins the oracle's predictors)."""
    enc = orc.Encoder(w, h, qp=qp, n_slices=slices, keyint=1, threads=8, intra_modes=3)
    eng = venc.Engine(w, h, qp=qp, n_slices=slices, max_items=1, in_slots=2, out_slots=2, intra_modes=3)
    kinds = set()
    for t, (y, u, v) in enumerate(_stripes_and_texture(venc, w, h, n)):
        ref = enc.encode(y, u, v)
        eng.upload(t % 2, y, u, v)
        eng.encode_step([t % 2], [t % 2], True, 0, [t & 1])
        eng.sync()
        flags = eng.debug(2)[:, 25]
        rec_o = enc.recon()
        if not np.array_equal(eng.debug(4), rec_o[0]):
            bad = np.argwhere(eng.debug(4) != rec_o[0])
            raise AssertionError("luma recon frame %d: first mismatch at (y, x) = %s, macroblock %d (flags %d), %d samples differ" % (
                t, bad[0], (bad[0][0] // 16) * ((w + 15) // 16) + bad[0][1] // 16, flags[(bad[0][0] // 16) * ((w + 15) // 16) + bad[0][1] // 16], len(bad)))
        lv_g, lv_o = eng.debug(1), enc.levels()
        if not np.array_equal(lv_g, lv_o):
            bad = np.argwhere(lv_g != lv_o)
            raise AssertionError("levels frame %d: first mismatch mb,blk,idx=%s gpu=%d orc=%d (%d diffs)" % (
                t, bad[0], lv_g[tuple(bad[0])], lv_o[tuple(bad[0])], len(bad)))
        for p in (1, 2):
            assert np.array_equal(eng.debug(4 + p), rec_o[p]), "recon plane %d frame %d" % (p, t)
        sl_o = [unescape(b) for (ty, _, b) in split_nals(ref.data) if ty in (1, 5)]
        assert eng.get_slices(t % 2) == sl_o, "slices frame %d" % t
        kinds |= set(int(x) & 8 for x in flags)
    if w >= 176 and h >= 144 and qp < 40:
        assert kinds == {0, 8}, kinds     # both macroblock types were chosen somewhere
    eng.close()


def test_x264_api_with_every_preset_tool_matches_oracle(venc, orc):
    """What a caller of the C API gets with the preset's defaults (every preset but ultrafast): CRF, Intra4x4 / Intra16x16 with plane,
    quarter-sample vectors, the in-loop filter -- through x264_encoder_encode, GOP-batched, against the oracle frame by frame."""
    w, h, n, keyint = 352, 288, 14, 5
    frames = [venc.synth_frame(w, h, t, 3) for t in range(n)]
    got = _x264_stream(venc, frames, w, h, slices=3, keyint=keyint, fps=(30, 1), sar=(1, 1), crf=24, gop_batch=2, intra_modes=3, subpel=2, deblock=True)
    enc = orc.Encoder(w, h, qp=24, n_slices=3, keyint=keyint, fps=(30, 1), sar=(1, 1), threads=8, rc=True, intra_modes=3, subpel=2, deblock=True)
    dec = orc.Decoder()
    for t, f in enumerate(frames):
        ref = enc.encode(*f)
        assert got[t].payload == ref.data, "frame %d" % t
        assert dec.decode(got[t].payload) == 1 and np.array_equal(dec.planes()[0], enc.recon()[0])


def test_reference_open_sequence_at_1080p_chooses_its_items_and_matches_oracle(venc, orc):
    """The reference's own open sequence (av_encode.c:384-401) at the size the metric is quoted on: preset medium, tune film, X264_RC_CRF,
    nothing else set.  The encoder chooses its GOPs in lock-step itself (32 with eight slices: the wavefront tools' blocks fill the
    chip), delays its output by that many GOPs and drains through the flush loop; the stream equals the oracle's with the same tools."""
    w, h, n = 1920, 1080, 34
    frames = [venc.synth_frame(w, h, t) for t in range(n)]
    e = venc.X264Encoder(w, h, fps=(30, 1), sar=(1, 1), preset="medium", tune="film", crf=20, reference_defaults=True)
    assert e.gop_batch() == 32 and e.param.hip.i_deblock == 1 and e.param.hip.i_intra_modes == 3 and e.param.hip.i_subpel == 2
    keyint, slices = e.param.i_keyint_max, e.param.i_slice_count
    got = []
    for t, (y, u, v) in enumerate(frames):
        r = e.encode(y, u, v, t)
        assert r is None  # (the unit is 32 GOPs: nothing comes back before the flush)
    assert e.delayed_frames() == n
    got = e.flush()
    e.close()
    enc = orc.Encoder(w, h, qp=20, n_slices=slices, keyint=keyint, fps=(30, 1), sar=(1, 1), threads=8, rc=True, intra_modes=3, subpel=2, deblock=True)
    for t, f in enumerate(frames):
        assert got[t].payload == enc.encode(*f).data, "frame %d" % t


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,want", [(1920, 1080, 30), (3840, 2160, 15), (2560, 1440, 17), (1280, 720, 64), (352, 288, 64)])
def test_items_in_lock_step_follow_the_picture_size(venc, w, h, want):
    """hip.i_gop_batch = 0 without the wavefront tools (the configuration bench.py quotes): about 245 000 macroblocks per step, at
    least 15 items, at most the engine's 64 (x264_shim.c shim_auto_gop_batch; include/x264.h: delay and memory); with them at least
    256 / slices.  Nothing is encoded: open, ask, close."""
    e = venc.X264Encoder(w, h, qp=26, slices=8, keyint=30, fps=(30, 1), sar=(1, 1), gop_batch=0)
    got = e.gop_batch()
    e.close()
    assert got == want
    e = venc.X264Encoder(w, h, qp=26, slices=8, keyint=30, fps=(30, 1), sar=(1, 1), gop_batch=0, deblock=True)
    got = e.gop_batch()
    e.close()
    assert got == max(want, 32)


def test_long_key_frame_interval_is_capped_by_device_memory_not_a_failed_open(venc, monkeypatch):
    """the unit is gop_batch x keyint frames and the pools twice that; with libx264's own default interval of 250
    at 2160p the preset's 32 items would ask for 400 GB.  x264_encoder_open lowers the items until the pools fit half of the free
    device memory; the stream still flows: frames in, the same number out through the flush loop, key frames where they belong.
    (The test narrows the budget to 12 GB -- VENC_POOL_BUDGET_MB -- so that it exercises the cap without holding a hundred GB, whose
    release slows whatever runs on the device next.)"""
    monkeypatch.setenv("VENC_POOL_BUDGET_MB", "12000")
    w, h, n = 1920, 1080, 6
    e = venc.X264Encoder(w, h, fps=(30, 1), sar=(1, 1), preset="medium", tune="film", crf=23, reference_defaults=True, preset_keyint=250)
    assert 1 <= e.gop_batch() < 32, e.gop_batch()
    out = []
    for t in range(n):
        r = e.encode(*venc.synth_frame(w, h, t), t)
        if r is not None:  # (one item left: the encoder streams, a frame behind)
            out.append(r)
    out += e.flush()
    e.close()
    assert len(out) == n and out[0].keyframe and not any(f.keyframe for f in out[1:]) and [f.pts for f in out] == list(range(n))


def test_x264_api_with_intra_modes_and_rate_control(venc, orc):
    """What a C caller gets from x264_param_default_preset + X264_RC_CRF (the reference's open sequence,
    av_encode.c:384-401): Intra16x16 V / H / DC and the CRF approximation together, GOP-batched."""
    w, h, n, keyint = 352, 288, 16, 4
    frames = _stripes_and_texture(venc, w, h, n)
    got = _x264_stream(venc, frames, w, h, slices=4, keyint=keyint, fps=(30, 1), sar=(1, 1), crf=23, gop_batch=2, intra_modes=True)
    enc = orc.Encoder(w, h, qp=23, n_slices=4, keyint=keyint, fps=(30, 1), sar=(1, 1), threads=8, rc=True, intra_modes=True)
    dec = orc.Decoder()
    for t, f in enumerate(frames):
        ref = enc.encode(*f)
        assert got[t].payload == ref.data, "frame %d" % t
        assert dec.decode(got[t].payload) == 1 and np.array_equal(dec.planes()[0], enc.recon()[0])


@pytest.mark.gpu
def test_opening_an_engine_puts_the_thread_beside_the_card(venc, monkeypatch):
    """venc_open restricts the opening thread to the CPUs of the card's NUMA node (sysfs) unless VENC_NUMA_BIND=0; venc_numa_bind
    reports the node (-1: one node, no sysfs, or nothing to do) and never widens the set the thread started with."""
    import os
    before = os.sched_getaffinity(0)
    try:
        monkeypatch.setenv("VENC_NUMA_BIND", "0")
        venc.Engine(176, 144, qp=30, n_slices=1, max_items=1, in_slots=2, out_slots=2).close()
        assert os.sched_getaffinity(0) == before
        monkeypatch.setenv("VENC_NUMA_BIND", "1")
        venc.Engine(176, 144, qp=30, n_slices=1, max_items=1, in_slots=2, out_slots=2).close()
        after = os.sched_getaffinity(0)
        assert after and after <= before
        node = venc.hip_lib().venc_numa_bind(0, 0)
        if node >= 0 and os.path.exists("/sys/devices/system/node/node%d/cpulist" % node):
            cpus = set()
            for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
                a, _, b = part.partition("-")
                cpus |= set(range(int(a), int(b or a) + 1))
            assert os.sched_getaffinity(0) <= cpus
    finally:
        os.sched_setaffinity(0, before)


@pytest.mark.gpu
def test_frames_come_back_a_unit_and_a_half_behind_the_input(venc):
    """With G GOPs in lock-step a frame is handed back G x keyint x 1.5 calls after it went in (include/x264.h: a unit is launched when its
    last picture has arrived, half a unit more covers its time on the GPU), in display order, and the flush loop drains the rest."""
    w, h, keyint, g, n = 176, 144, 4, 2, 40
    frames = [venc.synth_frame(w, h, t) for t in range(n)]
    e = venc.X264Encoder(w, h, qp=28, slices=2, keyint=keyint, gop_batch=g)
    unit, first = g * keyint, None
    out = []
    for t, (y, u, v) in enumerate(frames):
        r = e.encode(y, u, v, t)
        if r is not None:
            first = t if first is None else first
            out.append(r)
    assert first == unit + unit // 2
    assert e.delayed_frames() == n - len(out) == unit + unit // 2
    out += e.flush()
    e.close()
    assert [f.pts for f in out] == list(range(n)) and all(f.keyframe == (f.pts % keyint == 0) for f in out)
    ref = venc.X264Encoder(w, h, qp=28, slices=2, keyint=keyint, gop_batch=1)
    want = []
    for t, (y, u, v) in enumerate(frames):
        r = ref.encode(y, u, v, t)
        if r is not None:
            want.append(r)
    want += ref.flush()
    ref.close()
    assert [f.payload for f in out] == [f.payload for f in want]
