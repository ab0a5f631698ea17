"""CPU tests of the oracle (oracle/): structural checks of its tables,
known-answer tests of each stage, the encode -> decode self-consistency loop and
the edge cases of the domain.  PARITY UNPINNED: the reference ships no golden
vectors for this path; what is pinned here is the
oracle against hand-computable answers, its own decoder and the committed
fixtures in tests/golden/ (test_golden.py)."""
import numpy as np
import pytest

from conftest import noise_frames, split_nals, unescape


def _roundtrip(orc, w, h, qp, slices, keyint, frames, threads=1):
    enc = orc.Encoder(w, h, qp=qp, n_slices=slices, keyint=keyint, threads=threads)
    dec = orc.Decoder()
    sizes = []
    for t, (y, u, v) in enumerate(frames):
        fr = enc.encode(y, u, v)
        assert fr.is_idr == (t % keyint == 0)
        assert dec.decode(fr.data) == 1
        for p, (a, b) in enumerate(zip(enc.recon(), dec.planes())):
            assert np.array_equal(a, b), "decoder drift: frame %d plane %d" % (t, p)
        sizes.append(len(fr.data))
    assert dec.size() == (w, h)
    return enc, sizes


@pytest.mark.parametrize("qp", [12, 20, 26, 35, 44, 51])
@pytest.mark.parametrize("slices", [1, 3, 9])
def test_encode_decode_roundtrip(orc, qp, slices):
    _roundtrip(orc, 176, 144, qp, slices, 4, noise_frames(176, 144, 6))


def test_cropped_and_ragged_sizes(orc):
    for (w, h) in ((178, 146), (16, 16), (30, 18), (320, 180)):
        _roundtrip(orc, w, h, 28, 2, 3, noise_frames(w, h, 4, seed=w))


def test_threads_do_not_change_the_stream(orc):
    fr = noise_frames(176, 144, 5)
    a = orc.Encoder(176, 144, qp=26, n_slices=4, keyint=3, threads=1)
    b = orc.Encoder(176, 144, qp=26, n_slices=4, keyint=3, threads=4)
    for y, u, v in fr:
        assert a.encode(y, u, v).data == b.encode(y, u, v).data


def test_quality_is_monotone_in_qp(orc):
    fr = noise_frames(176, 144, 3)
    psnr, size = [], []
    for qp in (14, 26, 38, 50):
        enc, sizes = _roundtrip(orc, 176, 144, qp, 1, 3, fr)
        rec = enc.recon()[0][:144, :176].astype(np.float64)
        psnr.append(10 * np.log10(255 ** 2 / np.mean((rec - fr[-1][0]) ** 2)))
        size.append(sum(sizes))
    assert psnr == sorted(psnr, reverse=True) and psnr[0] > 44 and psnr[-1] > 17
    assert size == sorted(size, reverse=True)


def test_motion_search_finds_a_global_pan(orc):
    """Known answer: a textured frame panned by (+5, -3) must yield mv = (-5, +3)... everywhere inside."""
    rng = np.random.default_rng(3)
    big = rng.integers(0, 256, (200, 240)).astype(np.uint8)
    big = (big.astype(np.int32) + np.roll(big, 1, 0) + np.roll(big, 1, 1) + np.roll(big, 2, 0)) // 4
    big = big.astype(np.uint8)
    w, h = 176, 144

    def crop(ox, oy):
        y = big[20 + oy:20 + oy + h, 20 + ox:20 + ox + w]
        c = np.full((h // 2, w // 2), 128, np.uint8)
        return y, c, c
    enc = orc.Encoder(w, h, qp=12, n_slices=1, keyint=2)
    enc.encode(*crop(0, 0))
    enc.encode(*crop(5, -3))       # content moved left/down: block at x matches reference at x+5, y-3
    mv = enc.mv().reshape(enc.mbh, enc.mbw, 2)
    inner = mv[1:-1, 1:-1]
    assert (inner[..., 0] == 5).all() and (inner[..., 1] == -3).all()
    # zero-motion frame: every vector is (0,0) thanks to the L1 tie-break rank
    enc2 = orc.Encoder(w, h, qp=12, n_slices=1, keyint=2)
    flat = (np.full((h, w), 90, np.uint8), np.full((h // 2, w // 2), 128, np.uint8), np.full((h // 2, w // 2), 128, np.uint8))
    enc2.encode(*flat)
    enc2.encode(*flat)
    assert not enc2.mv().any() and not enc2.sad().any()


def test_me_rank_is_a_total_order(orc):
    L = orc.lib()
    ranks = {(dx, dy): L.orc_me_rank(dx, dy) for dx in range(-16, 17) for dy in range(-16, 17)}
    assert sorted(ranks.values()) == list(range(1089))
    assert ranks[(0, 0)] == 0
    assert ranks[(0, -1)] < ranks[(-1, 0)] < ranks[(1, 0)] < ranks[(0, 1)] < ranks[(-1, -1)]
    assert max(ranks.values()) < 2048 and (65280 << 11) | 2047 < 2 ** 32


def test_flat_frame_known_bitstream_shape(orc):
    """A mid-grey frame: DC prediction is exact, every macroblock codes nothing but its header."""
    w, h = 64, 48
    y = np.full((h, w), 128, np.uint8)
    c = np.full((h // 2, w // 2), 128, np.uint8)
    enc = orc.Encoder(w, h, qp=26, n_slices=1, keyint=2)
    fr = enc.encode(y, c, c)
    assert not enc.levels().any()
    # I16x16 DC, cbp 0: mb_type ue(3)=00100, chroma pred ue(0)=1, qp_delta se(0)=1, luma DC coeff_token(0)=1 -> 8 bits
    assert (enc.mb_bits() == 8).all()
    assert [n[0] for n in fr.nals] == [7, 8, 5]
    p = enc.encode(y, c, c)
    # zero vector, no residual: all 12 macroblocks are P_Skip; the slice data is one mb_skip_run = ue(12) = 0001101,
    # booked on the last macroblock
    bits = enc.mb_bits()
    assert (bits[:-1] == 0).all() and bits[-1] == 7 and [n[0] for n in p.nals] == [1]
    nal = [b for (ty, _, b) in split_nals(p.data) if ty == 1][0]
    assert len(nal) <= 4  # slice header + 7 bits + stop bit
    assert np.array_equal(enc.recon()[0], np.full((48, 64), 128, np.uint8))


def test_nal_framing_and_sps_bytes(orc):
    enc = orc.Encoder(1920, 1080, qp=26, n_slices=8, keyint=30, fps=(30, 1), sar=(1, 1))
    y = np.zeros((1080, 1920), np.uint8)
    c = np.zeros((540, 960), np.uint8)
    fr = enc.encode(y, c, c)
    nals = split_nals(fr.data)
    assert [t for t, _, _ in nals] == [7, 8] + [5] * 8
    sps = fr.data[fr.nals[0][2]:]
    assert sps[4] == 0x67 and sps[5] == 66 and sps[6] == 0xC0 and sps[7] == 40   # level 4.0 for 1080p30
    rows = enc.slice_rows()
    assert list(np.diff(rows)) == [9, 9, 9, 9, 8, 8, 8, 8]        # SURVEY.md section 8e band sizes
    enc4k = orc.Encoder(3840, 2160, qp=26, n_slices=8, keyint=30)
    assert list(np.diff(enc4k.slice_rows())) == [17, 17, 17, 17, 17, 17, 17, 16]
    # no start-code emulation anywhere in any NAL
    for _, _, body in nals:
        assert b"\x00\x00\x00" not in body and b"\x00\x00\x01" not in body and b"\x00\x00\x02" not in body


def test_emulation_prevention(orc):
    assert orc.escape(b"\x00\x00\x00\x00\x00") == b"\x00\x00\x03\x00\x00\x03\x00"
    assert orc.escape(b"\x00\x00\x01") == b"\x00\x00\x03\x01"
    assert orc.escape(b"\x00\x00\x04\x00\x00\x03") == b"\x00\x00\x04\x00\x00\x03\x03"
    assert orc.escape(b"\x01\x02") == b"\x01\x02"
    rng = np.random.default_rng(5)
    for _ in range(50):
        raw = bytes(rng.choice([0, 0, 0, 1, 2, 3, 4, 255], size=200).astype(np.uint8))
        assert unescape(orc.escape(raw)) == raw


def test_rejects_bad_parameters(orc):
    for kw in (dict(qp=11), dict(qp=52)):
        with pytest.raises(ValueError):
            orc.Encoder(176, 144, **kw)
    with pytest.raises(ValueError):
        orc.Encoder(175, 144)
    e = orc.Encoder(176, 144, n_slices=100)      # clamped to one slice per MB row
    assert len(e.slice_rows()) == 10
