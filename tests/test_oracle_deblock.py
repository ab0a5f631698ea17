"""In-loop deblocking of the oracle (spec/CODEC.md section 5a, H.264 8.7) -- CPU only.

PARITY AND CONFORMANCE UNPINNED: the reference holds no fixture for this path and no other H.264 decoder exists here.
Checked: (a) the C filter against a second restatement of 8.7 written here in Python from the standard's text (edge lists
per plane, strengths through a callback, filters on integer arrays) -- the two share the tables' VALUES, whose structure
is checked separately; (b) encoder -> decoder round trips; (c) that the filter does what it is for."""
import numpy as np
import pytest

from conftest import noise_frames
from oracle import orc_py

ALPHA = [0] * 16 + [4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162,
                    182, 203, 226, 255, 255]
BETA = [0] * 16 + [2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18]
TC0 = [(0, 0, 0)] * 17 + [(0, 0, 1)] * 4 + [(0, 1, 1)] * 2 + [(1, 1, 1)] * 4 + [(1, 1, 2)] * 4 + [(1, 2, 3)] * 2 + [(2, 2, 3), (2, 2, 4), (2, 3, 4), (2, 3, 4),
       (3, 3, 5), (3, 4, 6), (3, 4, 6), (4, 5, 7), (4, 5, 8), (4, 6, 9), (5, 7, 10), (6, 8, 11), (6, 8, 13), (7, 10, 14), (8, 11, 16), (9, 12, 18),
       (10, 13, 20), (11, 15, 23), (13, 17, 25)]
QPC = list(range(30)) + [29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39]
BLK_RASTER = [0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15]  # luma4x4BlkIdx -> 4 * by + bx


def test_table_structure():
    """Tables 8-16 / 8-17 as the standard prints them: 52 entries, zero below indexA 16 (tC0: 17), non-decreasing, alpha and
    beta end at 255 / 18, tC0 ends at 13 / 17 / 25, tC0 grows with bS"""
    assert len(ALPHA) == len(BETA) == len(TC0) == 52
    assert ALPHA[:16] == [0] * 16 and ALPHA[16] == 4 and ALPHA[51] == 255 and ALPHA == sorted(ALPHA)
    assert BETA[:16] == [0] * 16 and BETA[16] == 2 and BETA[51] == 18 and BETA == sorted(BETA)
    assert all(t == (0, 0, 0) for t in TC0[:17]) and TC0[51] == (13, 17, 25)
    for k in range(3):
        col = [t[k] for t in TC0]
        assert col == sorted(col)
    assert all(t[0] <= t[1] <= t[2] for t in TC0)


def _filter_line(s, bs, qpav, chroma):
    """s: the eight samples p3 p2 p1 p0 q0 q1 q2 q3 (chroma: p1 p0 q0 q1 are s[2:6]) as ints; returns the filtered list"""
    p3, p2, p1, p0, q0, q1, q2, q3 = s
    a, b = ALPHA[qpav], BETA[qpav]
    if not (bs and abs(p0 - q0) < a and abs(p1 - p0) < b and abs(q1 - q0) < b):
        return s
    clip3 = lambda lo, hi, v: max(lo, min(hi, v))
    out = list(s)
    if bs < 4:
        tc0 = TC0[qpav][bs - 1]
        ap, aq = abs(p2 - p0), abs(q2 - q0)
        tc = tc0 + 1 if chroma else tc0 + (1 if ap < b else 0) + (1 if aq < b else 0)
        delta = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3)
        out[3], out[4] = clip3(0, 255, p0 + delta), clip3(0, 255, q0 - delta)
        if not chroma:
            if ap < b:
                out[2] = p1 + clip3(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - 2 * p1) >> 1)
            if aq < b:
                out[5] = q1 + clip3(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - 2 * q1) >> 1)
        return out
    if chroma:
        out[3], out[4] = (2 * p1 + p0 + q1 + 2) >> 2, (2 * q1 + q0 + p1 + 2) >> 2
        return out
    small = abs(p0 - q0) < (a >> 2) + 2
    if small and abs(p2 - p0) < b:
        out[3], out[2], out[1] = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3, (p2 + p1 + p0 + q0 + 2) >> 2, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3
    else:
        out[3] = (2 * p1 + p0 + q1 + 2) >> 2
    if small and abs(q2 - q0) < b:
        out[4], out[5], out[6] = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3, (p0 + q0 + q1 + q2 + 2) >> 2, (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3
    else:
        out[4] = (2 * q1 + q0 + p1 + 2) >> 2
    return out


def python_deblock(planes, mbw, mbh, intra, nnz, mv, slice_of, qp):
    """8.7 for this codec's streams, macroblock by macroblock in raster order; planes: three int arrays, modified in place"""
    def strength(mbp, bp, mbq, bq, on_mb_edge):
        if intra[mbp] or intra[mbq]:
            return 4 if on_mb_edge else 3
        if nnz[mbp][bp] or nnz[mbq][bq]:
            return 2
        if mbp != mbq and (abs(int(mv[mbp][0]) - int(mv[mbq][0])) >= 4 or abs(int(mv[mbp][1]) - int(mv[mbq][1])) >= 4):
            return 1
        return 0

    for mb in range(mbw * mbh):
        mx, my = mb % mbw, mb // mbw
        for vertical in (True, False):
            nb = mb - 1 if vertical else mb - mbw
            have_nb = (mx > 0 if vertical else my > 0) and slice_of[nb] == slice_of[mb]
            for plane, size, step_edges, chroma in ((0, 16, (0, 4, 8, 12), False), (1, 8, (0, 4), True), (2, 8, (0, 4), True)):
                P = planes[plane]
                for pos in step_edges:
                    if pos == 0 and not have_nb:
                        continue
                    mbp = nb if pos == 0 else mb
                    qpav = (QPC[qp] + QPC[qp] + 1) >> 1 if chroma else qp
                    for k in range(size):  # sample along the edge
                        lk, lpos = (2 * k, 2 * pos) if chroma else (k, pos)  # the luma position that decides the strength
                        seg, e = lk >> 2, lpos >> 2
                        bq = 4 * seg + e if vertical else 4 * e + seg
                        bp = (4 * seg + 3 if vertical else 12 + seg) if pos == 0 else (4 * seg + e - 1 if vertical else 4 * (e - 1) + seg)
                        bs = strength(mbp, bp, mb, bq, pos == 0)
                        if not bs:
                            continue
                        if vertical:
                            yy, xx = my * size + k, mx * size + pos
                            idx = [(yy, xx + d) for d in range(-4, 4)]
                        else:
                            yy, xx = my * size + pos, mx * size + k
                            idx = [(yy + d, xx) for d in range(-4, 4)]
                        if chroma:  # only p1 p0 q0 q1 exist for the filter; pad the tuple so that the same routine serves
                            vals = [0, 0] + [int(P[i]) for i in idx[2:6]] + [0, 0]
                        else:
                            vals = [int(P[i]) for i in idx]
                        new = _filter_line(vals, bs, qpav, chroma)
                        for i, v in zip(idx[2:6] if chroma else idx[1:7], new[2:6] if chroma else new[1:7]):
                            P[i] = v


def _mb_state(enc, n_slices):
    lv = enc.levels()
    nnz = [[0] * 16 for _ in range(enc.nmb)]
    for mb in range(enc.nmb):
        for blk in range(16):
            nnz[mb][BLK_RASTER[blk]] = int(np.count_nonzero(lv[mb, blk]))
    rows = enc.slice_rows()
    slice_of = [0] * enc.nmb
    for s in range(len(rows) - 1):
        for mb in range(rows[s] * enc.mbw, rows[s + 1] * enc.mbw):
            slice_of[mb] = s
    return nnz, slice_of


@pytest.mark.parametrize("w,h,qp,slices", [(64, 48, 30, 1), (80, 64, 38, 2), (48, 48, 22, 3)])
def test_c_filter_matches_the_python_restatement(w, h, qp, slices):
    frames = noise_frames(w, h, 4, seed=11)
    enc = orc_py.Encoder(w, h, qp=qp, n_slices=slices, keyint=30, deblock=True, subpel=1)
    changed = 0
    for t, (y, u, v) in enumerate(frames):
        f = enc.encode(y, u, v)
        nnz, slice_of = _mb_state(enc, slices)
        intra = [1 if f.is_idr else 0] * enc.nmb
        if f.is_idr:  # Intra16x16: the strengths never look at coefficients
            nnz = [[0] * 16 for _ in range(enc.nmb)]
        planes = [p.astype(np.int64) for p in enc.recon_unfiltered()]
        python_deblock(planes, enc.mbw, enc.mbh, intra, nnz, enc.mvq(), slice_of, qp)
        for k, (a, b) in enumerate(zip(planes, enc.recon())):
            bad = np.argwhere(a != b)
            assert len(bad) == 0, "frame %d plane %d: first mismatch at %s: python %d C %d (%d differ)" % (t, k, bad[0], a[tuple(bad[0])], b[tuple(bad[0])], len(bad))
        changed += int(sum(np.count_nonzero(a != b) for a, b in zip(enc.recon_unfiltered(), enc.recon())))
    assert changed > 0


@pytest.mark.parametrize("subpel,rc,intra_modes", [(0, False, False), (2, True, True)])
def test_deblocked_streams_round_trip_through_the_decoder(subpel, rc, intra_modes):
    w, h = 112, 80
    frames = noise_frames(w, h, 7, seed=3)
    e = orc_py.Encoder(w, h, qp=33, n_slices=2, keyint=4, deblock=True, subpel=subpel, rc=rc, intra_modes=intra_modes)
    d = orc_py.Decoder()
    for (y, u, v) in frames:
        f = e.encode(y, u, v)
        assert d.decode(f.data) == 1
        for a, b in zip(e.recon(), d.planes()):
            assert np.array_equal(a, b)


def test_the_filter_smooths_block_edges():
    """smooth content, coarse quantiser: the step across 4x4 block edges (relative to the steps inside blocks) shrinks, and the
    pictures come closer to the source"""
    w, h = 128, 96
    yy, xx = np.mgrid[0:h, 0:w]
    frames = []
    for t in range(4):
        y = (128 + 70 * np.sin((xx + 2 * t) / 23.0) + 40 * np.cos((yy + t) / 17.0)).astype(np.uint8)
        c = np.full((h // 2, w // 2), 128, np.uint8)
        frames.append((y, (c + (xx[::2, ::2] // 8)).astype(np.uint8), c))

    def run(deblock):
        e = orc_py.Encoder(w, h, qp=42, n_slices=2, keyint=30, deblock=deblock)
        out = [e.encode(*f).data for f in frames]
        return out, e.recon()[0].astype(np.int64)

    off, rec_off = run(False)
    on, rec_on = run(True)
    assert off != on

    def blockiness(r):
        d = np.abs(np.diff(r, axis=1))
        return d[:, 3::4].mean() / max(np.delete(d, np.s_[3::4], axis=1).mean(), 1e-9)

    assert blockiness(rec_on) < blockiness(rec_off)
    src = frames[-1][0].astype(np.int64)
    assert np.mean((rec_on - src) ** 2) < np.mean((rec_off - src) ** 2)
