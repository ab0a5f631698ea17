"""GPU tests (-m gpu) of the slice-parallel multi-GPU mode (BASELINE config 5):
band-restricted kernels, the band pack/unpack kernel, the x264-API band encoder and the
torch.distributed exchange.  A one-GPU box cannot run RCCL between ranks (one rank per device), so
the multi-rank case is rehearsed with gloo (two processes sharing the card) and the RCCL code path
(device all-gather on the engine's own HIP stream) with a world of one.  Bit-exact bar: the
band-ordered concatenation equals the single-encoder output AND the CPU oracle's (oracle/: this
repository's own restatement, parity with libx264 unpinned)."""
import importlib
import os
import sys

import numpy as np
import pytest

from conftest import noise_frames, split_nals, unescape

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("w,h,n_slices,cuts,halo_rows,items,deblock,xstream", [
    (352, 288, 6, (0, 2, 6), 0, 1, False, False),        # two bands, whole-band exchange
    (352, 288, 6, (0, 1, 3, 6), 16, 1, False, False),    # three bands (one of a single slice), halo exchange
    (1920, 1088, 8, (0, 3, 8), 16, 2, False, False),     # full-HD geometry, odd first band row count, two items in lock-step
    (352, 288, 6, (0, 1, 3, 6), 16, 1, True, False),     # in-loop filter inside slices (idc 2): bands stay independent, halos carry filtered rows
    (1920, 1088, 8, (0, 3, 5, 8), 16, 2, False, True),   # the exchange on the exchange stream: the next search as interior + edge launches
    (1920, 1088, 8, (0, 3, 8), 0, 1, False, True),       # ... whole-band exchange
    (352, 288, 6, (0, 1, 3, 6), 16, 1, True, True),      # ... bands too short to split (one or two tile rows): the step waits instead
])
def test_band_engines_reproduce_the_whole_picture_engine(venc, orc, w, h, n_slices, cuts, halo_rows, items, deblock, xstream):
    """One engine per band on the same GPU, rows swapped with venc_band_export / venc_band_import through a
    device chunk: slices, motion vectors and (whole-band mode) every rank's full reconstruction equal the
    whole-picture engine's, which equals the oracle's.  xstream: export and imports between venc_band_xbegin / venc_band_xend
    (the exchange stream); the engines are NOT synchronised between the imports and the next step, which must then launch the
    tile rows that read no neighbour's row beside the imports and the edge rows behind them."""
    import torch
    n, keyint, qp = 5, 4, 27
    clips = [[venc.synth_frame(w, h, t, 0x1234ABCD + 17 * i) for t in range(n)] for i in range(items)]
    whole = venc.Engine(w, h, qp=qp, n_slices=n_slices, max_items=items, in_slots=2 * items, out_slots=2 * items, deblock=deblock)
    bands = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        e = venc.Engine(w, h, qp=qp, n_slices=n_slices, max_items=items, in_slots=2 * items, out_slots=2 * items, deblock=deblock)
        e.set_band(a, b - a)
        bands.append((e, a, b - a))
    rows = [whole.band_rows(a, c) for _, a, c in bands]
    stride = (whole.band_chunk_bytes(max(r1 - r0 for r0, r1 in rows), halo_rows) + 255) & ~255
    chunks = [torch.zeros(stride * items, dtype=torch.uint8, device="cuda") for _ in bands]
    torch.cuda.synchronize()
    refs = [orc.Encoder(w, h, qp=qp, n_slices=n_slices, keyint=keyint, threads=8, deblock=deblock) for _ in range(items)]
    for t in range(n):
        idr = t % keyint == 0
        slots = [2 * i + (t & 1) for i in range(items)]
        for e in [whole] + [b[0] for b in bands]:
            for i in range(items):
                e.upload(slots[i], *clips[i][t])
            e.encode_step(slots, slots, idr, (t % keyint) & 15, [(t // keyint) & 1] * items)
            e.sync()
        # bitstream: band slices in band order == whole-picture slices == oracle
        for i in range(items):
            want = whole.get_slices(slots[i])
            got = []
            for e, a, c in bands:
                sl = e.get_slices(slots[i])
                assert all(len(x) == 0 for k, x in enumerate(sl) if not a <= k < a + c), "foreign slice not empty"
                got += sl[a:a + c]
            assert got == want, "frame %d item %d: band slices differ from the whole-picture engine" % (t, i)
            ref = refs[i].encode(*clips[i][t])
            assert want == [unescape(b) for (ty, _, b) in split_nals(ref.data) if ty in (1, 5)], "oracle, frame %d" % t
        # exchange (the next picture may be P)
        for k, (e, a, c) in enumerate(bands):
            if xstream:
                e.band_xbegin()
            assert e.band_export(items, chunks[k].data_ptr(), stride, halo_rows) > 0
            if not xstream:
                e.sync()
        if xstream:
            torch.cuda.synchronize()  # (every band's chunk is packed before a peer unpacks it: what the transfer's own ordering does between GPUs)
        for k, (e, a, c) in enumerate(bands):
            peers = [j for j in (k - 1, k + 1) if 0 <= j < len(bands)] if halo_rows else \
                    [j for j in range(len(bands)) if j != k]
            for j in peers:
                e.band_import(items, chunks[j].data_ptr(), stride, bands[j][1], bands[j][2], halo_rows)
            if xstream:
                e.band_xend()   # no synchronisation: the next step's search overlaps the imports
            else:
                e.sync()
        if halo_rows == 0:
            for i in range(items):
                for p in range(3):
                    full = whole.debug(4 + p, i)
                    for e, _, _ in bands:
                        assert np.array_equal(e.debug(4 + p, i), full), "frame %d item %d plane %d" % (t, i, p)
    if xstream:
        # P steps behind an exchange: frames 1 .. n - 1 except the IDRs; bands of three or more tile rows split their search
        n_p = sum(1 for t in range(1, n) if t % keyint != 0)
        for (e, a, c), (r0, r1) in zip(bands, rows):
            splits = e.band_split_steps()
            assert splits == (n_p if (r1 - r0 + 1) // 2 >= 3 else 0), "band %d..%d: %d split steps" % (a, a + c, splits)
    for e in [whole] + [b[0] for b in bands]:
        e.close()


def test_band_arguments_are_checked(venc):
    e = venc.Engine(176, 144, qp=30, n_slices=3, max_items=1, in_slots=2, out_slots=2)
    with pytest.raises(venc.VencError):
        e.set_band(2, 2)
    with pytest.raises(venc.VencError):
        e.band_rows(0, 4)
    with pytest.raises(venc.VencError):
        e.band_export(1, 0, 4096, 0)       # null chunk
    assert e.band_chunk_bytes(3, 0) == 3 * 16 * 176 * 3 // 2
    assert e.band_chunk_bytes(3, 16) == 2 * 16 * 176 * 3 // 2
    assert e.band_chunk_bytes(2, 16) == 2 * 16 * 176 * 3 // 2  # halos cover the band: whole band
    assert e.band_chunk_bytes(1, 8) == 0                        # halo must be a multiple of 16
    e.set_band(0, 0)
    e.close()
    # a band without an exchange hook cannot code P pictures correctly: refused at open
    with pytest.raises(venc.VencError):
        venc.X264Encoder(176, 144, qp=30, slices=3, slice_first=1, slice_count=1)


def test_rccl_exchange_path_on_the_engine_stream_world_of_one(venc):
    """The device path of BandExchange (all_gather_into_tensor under the engine's HIP stream wrapped as a
    torch ExternalStream) with backend nccl = RCCL; one rank is all a one-GPU box can host."""
    import torch
    import torch.distributed as dist
    d = importlib.import_module("video-encoder_amd.dist")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29800 + os.getpid() % 100))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        w, h = 352, 288
        frames = [venc.synth_frame(w, h, t) for t in range(4)]
        xc = d.BandExchange(d.EnginePort(0), 4, 1, 0, halo_rows=16)
        xc.always = True  # run export + gather even though there is no peer
        got = _stream(venc, frames, w, h, qp=26, slices=4, keyint=4, tune="zerolatency", slice_first=0, slice_count=4,
                      exchange=xc)
        want = _stream(venc, frames, w, h, qp=26, slices=4, keyint=4, tune="zerolatency")
        assert got == want and xc.exchanges == 3
    finally:
        dist.destroy_process_group()


def _stream(venc, frames, w, h, **kw):
    e = venc.X264Encoder(w, h, **kw)
    out = []
    for t, (y, u, v) in enumerate(frames):
        r = e.encode(y, u, v, t)
        if r is not None:
            out.append(r.payload)
    out += [r.payload for r in e.flush()]
    e.close()
    return out


def _rank(rank, world, port, q, halo_rows, gop_batch):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    venc = importlib.import_module("video-encoder_amd")
    d = importlib.import_module("video-encoder_amd.dist")
    d.init_from_env("gloo")
    w, h, n_frames, keyint, n_slices = 352, 288, 10, 4, 5
    frames = [venc.synth_frame(w, h, t) for t in range(n_frames)]
    s0, cnt = d.slice_band(n_slices, world, rank)
    xc = d.BandExchange(d.EnginePort(0), n_slices, world, rank, halo_rows=halo_rows)
    kw = dict(qp=26, slices=n_slices, keyint=keyint, fps=(30, 1))
    if gop_batch == 1:
        kw["tune"] = "zerolatency"
    mine = _stream(venc, frames, w, h, gop_batch=gop_batch, slice_first=s0, slice_count=cnt, exchange=xc, **kw)
    parts = d.gather_frames(mine)
    if rank == 0:
        import orc_py
        whole = d.interleave_bands(parts)
        single = _stream(venc, frames, w, h, gop_batch=gop_batch, **kw)
        ref = orc_py.Encoder(w, h, qp=26, n_slices=n_slices, keyint=keyint, fps=(30, 1), threads=4)
        oracle = [ref.encode(*f).data for f in frames]
        q.put((whole == single, whole == oracle, xc.exchanges, len(whole)))
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("halo_rows,gop_batch", [(0, 1), (16, 3)])
def test_two_rank_gloo_band_encoders_reproduce_the_single_stream(halo_rows, gop_batch):
    """Two processes share the GPU, each runs the x264-API encoder on its band (slice_first / slice_count /
    pf_exchange) with BandExchange over gloo: the interleaved payloads equal the single encoder's and the
    CPU oracle's, frame by frame -- streaming (gop_batch 1) and with three GOPs in lock-step."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() + 11 * gop_batch) % 90
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q, halo_rows, gop_batch)) for r in range(2)]
    for p in procs:
        p.start()
    same_single, same_oracle, exchanges, n = q.get(timeout=500)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert n == 10 and same_single and same_oracle
    assert exchanges > 0


@pytest.mark.parametrize("halo_rows,gop_batch", [(16, 1), (0, 1), (16, 3)])
def test_c_level_rccl_exchange_world_of_one(venc, halo_rows, gop_batch):
    """The exchange done by the engine itself, in C on RCCL (include/venc_hip.h "RCCL exchange": venc_rccl_unique_id,
    venc_rccl_init_rank, venc_band_exchange_rccl -- librccl bound at run time), through the x264 API's hip.i_rccl_*
    parameters; no Python on the per-picture path.  One rank is all a one-GPU box can host (RCCL refuses two ranks on
    one device), so the collective runs with itself as its only member: pack kernel, grouped send / receive (halo) or
    all-gather (whole band) on the engine's stream, unpack loop.  Stream identical to the plain encoder's; the exchange
    count is one per picture that a P picture follows; venc_rccl_agree returns the maximum flag."""
    H = venc.hip_lib()
    w, h, n, keyint = 352, 288, 8, 4
    frames = [venc.synth_frame(w, h, t) for t in range(n)]
    kw = dict(qp=26, slices=4, keyint=keyint, gop_batch=gop_batch)
    if gop_batch == 1:
        kw["tune"] = "zerolatency"
    e = venc.X264Encoder(w, h, rccl=(1, 0, venc.rccl_unique_id(), halo_rows), **kw)
    eng = e.L.x264_venc_engine(e.h)
    assert H.venc_rccl_world(eng) == 1 and H.venc_rccl_rank(eng) == 0
    H.venc_rccl_always(eng, 1)
    got = []
    for t, (y, u, v) in enumerate(frames):
        r = e.encode(y, u, v, t)
        if r is not None:
            got.append(r.payload)
    got += [r.payload for r in e.flush()]
    units = n // keyint if gop_batch == 1 else 1  # GOPs in lock-step share their exchanges: one per step
    assert H.venc_rccl_exchanges(eng) == units * (keyint - 1)
    assert H.venc_rccl_agree(eng, 0) == 0 and H.venc_rccl_agree(eng, 5) == 5
    e.close()
    assert got == _stream(venc, frames, w, h, **kw)


def test_c_level_rccl_arguments_are_checked(venc):
    H = venc.hip_lib()
    e = venc.Engine(176, 144, qp=30, n_slices=3, max_items=1, in_slots=2, out_slots=2)
    assert H.venc_band_exchange_rccl(e.h, 1, 16) != 0          # no communicator yet
    ident = venc.rccl_unique_id()
    import ctypes as C
    buf = (C.c_ubyte * 128).from_buffer_copy(ident)
    assert H.venc_rccl_init_rank(e.h, buf, 4, 0) != 0          # more ranks than slices: refused before RCCL is called
    assert b"slices" in H.venc_last_error()
    e.close()


def test_c_level_rccl_single_process_init_all(venc, orc):
    """A C host that drives every GPU from one process (the reference's shape: one program, one thread) uses
    venc_rccl_init_all (= ncclCommInitAll over the contexts' devices) and brackets the per-context exchanges of a picture
    with venc_rccl_group_begin / venc_rccl_group_end.  On the one-GPU box: one context, which then codes the whole picture;
    slices equal the oracle's."""
    import ctypes as C
    H = venc.hip_lib()
    w, h, n = 352, 288, 4
    e = venc.Engine(w, h, qp=28, n_slices=4, max_items=1, in_slots=2, out_slots=2)
    arr = (C.c_void_p * 1)(e.h)
    assert H.venc_rccl_init_all(arr, 1) == 0, H.venc_last_error()
    assert H.venc_rccl_world(e.h) == 1 and e.band_rows(0, 4) == (0, 18)
    H.venc_rccl_always(e.h, 1)
    H.venc_rccl_poison(e.h, 1)
    ref = orc.Encoder(w, h, qp=28, n_slices=4, keyint=n, threads=4)
    for t in range(n):
        f = venc.synth_frame(w, h, t, 3)
        e.upload(t & 1, *f)
        e.encode_step([t & 1], [t & 1], t == 0, t & 15, [0])
        assert H.venc_rccl_group_begin() == 0
        assert H.venc_band_exchange_rccl(e.h, 1, 16) == 0, H.venc_last_error()
        # inside the group RCCL has only recorded the transfer: the unpack half must wait for venc_rccl_group_end
        assert H.venc_rccl_pending() == 1
        assert H.venc_rccl_group_end() == 0
        assert H.venc_rccl_pending() == 0
        e.sync()
        want = [unescape(b) for (ty, _, b) in split_nals(ref.encode(*f).data) if ty in (1, 5)]
        assert e.get_slices(t & 1) == want, "frame %d" % t
    assert H.venc_rccl_exchanges(e.h) == n
    e.close()
