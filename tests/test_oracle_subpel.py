"""Sub-sample refinement of the oracle (spec/CODEC.md section 2a) -- CPU only.

PARITY UNPINNED: the reference holds no fixture for this path (libx264 is absent); what is checked here is (a) the
interpolation against an independent numpy restatement of H.264 8.4.2.2.1 (whole-plane separable filtering and the
standard's table of quarter positions by name, where the C code walks a doubled grid by parity), (b) encoder -> decoder
round trips, (c) that the refinement does what it is for (fewer bytes, higher PSNR on sub-sample motion) and leaves the
integer path alone."""
import numpy as np
import pytest

from oracle import orc_py
from subpel_clip import subpel_clip


def _numpy_planes(p):
    """G, b, h, j of 8.4.2.2.1 for every integer position of plane p (edge-replicated), as int arrays"""
    H, W = p.shape
    pad = 8
    q = np.pad(p.astype(np.int64), pad, mode="edge")
    taps = np.array([1, -5, 20, 20, -5, 1], dtype=np.int64)

    def filt(a, axis):  # sum over k of taps[k] * a[pos - 2 + k]
        out = np.zeros_like(a)
        for k, t in enumerate(taps):
            out += t * np.roll(a, 2 - k, axis=axis)
        return out

    b1 = filt(q, 1)
    h1 = filt(q, 0)
    j1 = filt(b1, 0)
    clip = lambda v: np.clip(v, 0, 255)
    crop = lambda a: a[pad - 4:pad + H + 4, pad - 4:pad + W + 4]  # a margin of 4 integer positions around the plane
    return crop(q), crop(clip((b1 + 16) >> 5)), crop(clip((h1 + 16) >> 5)), crop(clip((j1 + 512) >> 10))


def _numpy_qpel(planes, x, y, fx, fy):
    """sample at integer position (x, y) (may lie up to 3 outside the plane) + (fx, fy) / 4, by the standard's names"""
    G_, b_, h_, j_ = planes
    at = lambda a, xx, yy: int(a[yy + 4, xx + 4])
    G, H, M = at(G_, x, y), at(G_, x + 1, y), at(G_, x, y + 1)
    b, h, j = at(b_, x, y), at(h_, x, y), at(j_, x, y)
    m, s = at(h_, x + 1, y), at(b_, x, y + 1)
    avg = lambda p, q: (p + q + 1) >> 1
    table = {(0, 0): G, (1, 0): avg(G, b), (2, 0): b, (3, 0): avg(H, b),
             (0, 1): avg(G, h), (1, 1): avg(b, h), (2, 1): avg(b, j), (3, 1): avg(b, m),
             (0, 2): h, (1, 2): avg(h, j), (2, 2): j, (3, 2): avg(j, m),
             (0, 3): avg(M, h), (1, 3): avg(h, s), (2, 3): avg(j, s), (3, 3): avg(m, s)}
    return table[(fx, fy)]


def test_interpolation_matches_a_numpy_restatement_of_8_4_2_2_1():
    rng = np.random.default_rng(5)
    p = rng.integers(0, 256, size=(24, 40), dtype=np.uint8)
    p[:6, :8] = 255  # saturated corner: the clips of b, h, j matter
    p[-5:, -7:] = 0
    planes = _numpy_planes(p)
    H, W = p.shape
    for fy in range(4):
        for fx in range(4):
            # blocks that straddle every border (coordinates clamp) and one inside
            for (x0, y0, bw, bh) in ((-3, -3, 8, 8), (W - 5, H - 5, 8, 8), (10, 7, 16, 16), (-2, H - 6, 4, 8)):
                got = orc_py.luma_interp(p, 4 * x0 + fx, 4 * y0 + fy, bw, bh)
                for yy in range(bh):
                    for xx in range(bw):
                        # positions further than 3 outside are not covered by the numpy margin: clamp the integer part the
                        # way the standard does (Clip3 applies to each tap, so a far-outside sample equals the border one)
                        xi, yi = x0 + xx, y0 + yy
                        if -3 <= xi < W + 2 and -3 <= yi < H + 2:
                            assert got[yy, xx] == _numpy_qpel(planes, xi, yi, fx, fy), (fx, fy, x0, y0, xx, yy)


def test_six_tap_filter_known_answers():
    """constant planes stay constant; a linear ramp is reproduced exactly at half positions (taps sum to 32, symmetric:
    the half sample of a ramp x -> 4x is 4x + 2); the impulse response is the tap vector"""
    c = np.full((16, 16), 77, dtype=np.uint8)
    for fy in range(4):
        for fx in range(4):
            assert (orc_py.luma_interp(c, 4 * 4 + fx, 4 * 4 + fy, 4, 4) == 77).all()
    ramp = np.tile((4 * np.arange(32)).astype(np.uint8), (16, 1))
    assert (orc_py.luma_interp(ramp, 4 * 8 + 2, 4 * 4, 8, 4) == (4 * np.arange(8, 16) + 2)[None, :]).all()
    assert (orc_py.luma_interp(ramp.T.copy(), 4 * 4, 4 * 8 + 2, 4, 8) == (4 * np.arange(8, 16) + 2)[:, None]).all()
    imp = np.zeros((16, 32), dtype=np.uint8)
    imp[8, 16] = 32  # b around the impulse: (32 * tap + 16) >> 5 = tap, negatives clip to 0
    row = orc_py.luma_interp(imp, 4 * 13 + 2, 4 * 8, 6, 1)[0]
    assert list(row) == [1, 0, 20, 20, 0, 1]


@pytest.mark.parametrize("subpel", [1, 2])
def test_refined_streams_round_trip_through_the_decoder(subpel):
    w, h = 96, 80
    clip = subpel_clip(w, h, 5, factor=4, sx=5, sy=-3)
    e = orc_py.Encoder(w, h, qp=24, n_slices=2, keyint=30, subpel=subpel)
    d = orc_py.Decoder()
    frac = 0
    for (y, u, v) in clip:
        f = e.encode(y, u, v)
        d.decode(f.data)
        for a, b in zip(e.recon(), d.planes()):
            assert np.array_equal(a, b)
        if not f.is_idr:
            q = e.mvq()
            frac += int(((q & 3) != 0).any(axis=1).sum())
            assert np.abs(q).max() <= 67
            if subpel == 1:
                assert not (q & 1).any()  # half-sample vectors only
    assert frac > 0


def test_refinement_pays_on_sub_sample_motion_and_leaves_the_integer_path_alone():
    w, h = 128, 96
    half = subpel_clip(w, h, 5, factor=2, sx=3, sy=1)     # 1.5 / 0.5 samples per frame
    quarter = subpel_clip(w, h, 5, factor=4, sx=5, sy=3)  # 1.25 / 0.75 samples per frame

    def run(clip, subpel):
        e = orc_py.Encoder(w, h, qp=26, n_slices=1, keyint=30, subpel=subpel)
        n = 0
        for (y, u, v) in clip:
            f = e.encode(y, u, v)
            n += len(f.data)
            if not f.is_idr and subpel == 0:
                assert np.array_equal(e.mvq(), 4 * e.mv())
        rec = e.recon()[0][:h, :w].astype(np.float64)
        psnr = 10 * np.log10(255.0 ** 2 / np.mean((rec - clip[-1][0].astype(np.float64)) ** 2))
        return n, psnr

    b0, p0 = run(half, 0)
    b1, p1 = run(half, 1)
    assert b1 < 0.8 * b0 and p1 > p0 + 0.5
    q0, _ = run(quarter, 0)
    q1, _ = run(quarter, 1)
    q2, pq2 = run(quarter, 2)
    assert q2 < q1 < q0


def test_refinement_never_raises_a_macroblocks_sad():
    w, h = 64, 48
    clip = subpel_clip(w, h, 3, factor=2, sx=1, sy=1)
    a = orc_py.Encoder(w, h, qp=26, keyint=30, subpel=0)
    b = orc_py.Encoder(w, h, qp=26, keyint=30, subpel=2)
    for t, (y, u, v) in enumerate(clip):
        a.encode(y, u, v)
        b.encode(y, u, v)
        if t == 1:  # same reference (the I picture): the refined SAD is at most the full search's
            assert np.array_equal(a.mv(), b.mv())
            assert (b.sad() <= a.sad()).all()
