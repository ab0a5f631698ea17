"""CPU checks of the per-lane device functions (csrc/venc_device.h compiled by
g++ into libvenc_hostcheck.so -- a TEST artefact, never loaded by the product)
against the CPU oracle.  This is how the lane-level logic of the HIP kernels is
verified in a container without a GPU; the kernels themselves are checked by
tests/test_gpu_parity.py on the GPU box."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def hc(venc):
    venc.build()
    L = C.CDLL(venc.LIB_HOSTCHECK)
    L.hc_synth_pixel.restype = C.c_ubyte
    L.hc_fdct4.argtypes = [C.c_void_p, C.c_void_p]
    L.hc_idct4.argtypes = [C.c_void_p, C.c_void_p]
    L.hc_cavlc_block.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.hc_cavlc_block_mem.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.hc_ue_bits.argtypes = [C.c_uint, C.c_void_p]
    L.hc_se_bits.argtypes = [C.c_int, C.c_void_p]
    L.hc_table.argtypes = [C.c_int, C.c_void_p]
    L.hc_synth_pixel.argtypes = [C.c_int] * 6 + [C.c_uint]
    return L


def test_transforms_match_oracle(hc, orc):
    rng = np.random.default_rng(0)
    L = orc.lib()
    for _ in range(300):
        x = rng.integers(-255, 256, 16).astype(np.int32)
        w = np.zeros(16, np.int32)
        hc.hc_fdct4(x.ctypes.data, w.ctypes.data)
        x16, w16 = x.astype(np.int16), np.zeros(16, np.int16)
        L.orc_fdct4x4(x16.ctypes.data, w16.ctypes.data)
        assert np.array_equal(w, w16)
        d = rng.integers(-20000, 20000, 16).astype(np.int32)
        r, r16 = np.zeros(16, np.int32), np.zeros(16, np.int16)
        hc.hc_idct4(d.ctypes.data, r.ctypes.data)
        L.orc_idct4x4(d.ctypes.data, r16.ctypes.data)
        assert np.array_equal(r, r16)


def test_known_answers(hc):
    # constant block: only the DC coefficient, equal to 16 x value
    x = np.full(16, 7, np.int32)
    w = np.zeros(16, np.int32)
    hc.hc_fdct4(x.ctypes.data, w.ctypes.data)
    assert w[0] == 112 and not w[1:].any()
    # inverse of a pure DC of 64*k reconstructs the constant k
    d = np.zeros(16, np.int32)
    d[0] = 64 * 5
    r = np.zeros(16, np.int32)
    hc.hc_idct4(d.ctypes.data, r.ctypes.data)
    assert (r == 5).all()
    assert [hc.hc_zigzag(i) for i in range(16)] == [0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15]


def test_quantiser_matches_oracle(hc, orc):
    L = orc.lib()
    rng = np.random.default_rng(1)
    for qp in range(12, 52):
        assert hc.hc_chroma_qp(qp) == [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21,
                                       22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36,
                                       37, 37, 37, 38, 38, 38, 39, 39, 39, 39][qp]
        for w in list(rng.integers(-9180, 9181, 40)) + [0, 1, -1, 9180, -9180]:
            for r in (0, 1, 5, 10, 15):
                for intra in (0, 1):
                    assert hc.hc_quant(int(w), qp, r, intra) == L.orc_quant(int(w), qp, r, intra)
        for lv in (-2063, -7, -1, 0, 1, 3, 100):
            for r in range(16):
                assert hc.hc_dequant(lv, qp, r) == L.orc_dequant(lv, qp, r)


def _rand_block(rng, max_coef, density, big):
    c = np.zeros(16, np.int32)
    n = max_coef
    mask = rng.random(n) < density
    mag = rng.integers(1, 2100 if big else 4, n)
    sign = rng.integers(0, 2, n) * 2 - 1
    c[:n] = np.where(mask, mag * sign, 0)
    return c


def test_cavlc_block_matches_oracle(hc, orc):
    L = orc.lib()
    rng = np.random.default_rng(2)
    out_h, out_o = np.zeros(128, np.uint8), np.zeros(128, np.uint8)
    cases = 0
    for max_coef, first in ((16, 0), (15, 1), (4, 0)):
        for nC in ((-1,) if max_coef == 4 else (0, 1, 2, 3, 4, 7, 8, 16)):
            for density in (0.0, 0.1, 0.3, 0.6, 1.0):
                for big in (False, True):
                    for _ in range(25):
                        blk = _rand_block(rng, max_coef, density, big)
                        c_dev = np.zeros(16, np.int32)
                        c_dev[first:first + max_coef] = blk[:max_coef]
                        bits_h = hc.hc_cavlc_block(c_dev.ctypes.data, first, max_coef, nC, out_h.ctypes.data, 128)
                        c16 = blk[:max_coef].astype(np.int16)
                        bits_o = L.orc_cavlc_block_bits(c16.ctypes.data, max_coef, nC, out_o.ctypes.data, 128)
                        assert bits_h == bits_o and bits_h > 0
                        nb = (bits_h + 7) // 8
                        assert bytes(out_h[:nb]) == bytes(out_o[:nb]), (max_coef, nC, blk)
                        # the data-driven variant the kernel uses (levels in memory, tables in one array)
                        c_mem = c_dev.astype(np.int16)
                        bits_m = hc.hc_cavlc_block_mem(c_mem.ctypes.data, first, max_coef, nC, out_h.ctypes.data, 128)
                        assert bits_m == bits_o and bytes(out_h[:nb]) == bytes(out_o[:nb]), (max_coef, nC, blk)
                        cases += 1
    assert cases > 2000


def test_exp_golomb(hc):
    buf = np.zeros(16, np.uint8)
    known = {0: "1", 1: "010", 2: "011", 3: "00100", 6: "00111", 7: "0001000", 32399: None}
    for v, s in known.items():
        bits = hc.hc_ue_bits(v, buf.ctypes.data)
        got = "".join("{:08b}".format(b) for b in buf[:(bits + 7) // 8])[:bits]
        if s is not None:
            assert got == s
        assert bits == 2 * int(np.floor(np.log2(v + 1))) + 1
    for v, s in {0: "1", 1: "010", -1: "011", 2: "00100", -2: "00101"}.items():
        bits = hc.hc_se_bits(v, buf.ctypes.data)
        assert "".join("{:08b}".format(b) for b in buf[:(bits + 7) // 8])[:bits] == s


def _check_prefix_code(entries, complete_minus=None):
    """entries: list of (len, bits) with len > 0.  Prefix-free; Kraft sum = 1 - 2^-complete_minus (or 1)."""
    from fractions import Fraction
    k = sum(Fraction(1, 1 << l) for l, _ in entries)
    for i, (l1, b1) in enumerate(entries):
        assert b1 < (1 << l1)
        for j, (l2, b2) in enumerate(entries):
            if i != j and l2 >= l1:
                assert (b2 >> (l2 - l1)) != b1, "prefix clash"
    expect = Fraction(1) - (Fraction(1, 1 << complete_minus) if complete_minus else 0)
    assert k == expect, (k, expect)


def test_product_vlc_tables_structure(hc):
    buf = (C.c_ushort * 272)()

    def tab(which):
        n = hc.hc_table(which, buf)
        return [(buf[i] >> 8, buf[i] & 0xff) for i in range(n)]
    ct = tab(0)
    for t, missing in enumerate((15, 13, 10, None)):
        ent = [e for e in ct[68 * t:68 * t + 68] if e[0]]
        assert len(ent) == 62
        if missing:
            _check_prefix_code(ent, missing)      # only the all-zero codeword is unused
        else:
            assert all(l == 6 for l, _ in ent) and len({b for _, b in ent}) == 62
    _check_prefix_code([e for e in tab(1) if e[0]])
    tz = tab(2)
    for t in range(15):
        ent = [e for e in tz[16 * t:16 * t + 16] if e[0]]
        assert len(ent) == 16 - t
        _check_prefix_code(ent, 9 if t == 0 else None)
    tzc = tab(3)
    for t in range(3):
        _check_prefix_code([e for e in tzc[4 * t:4 * t + 4] if e[0]])
    rb = tab(4)
    for t in range(7):
        ent = [e for e in rb[16 * t:16 * t + 16] if e[0]]
        assert len(ent) == (t + 2 if t < 6 else 15)
        _check_prefix_code(ent, 11 if t == 6 else None)
    assert sorted(hc.hc_cbp_code(c) for c in range(48)) == list(range(48))
    assert hc.hc_cbp_code(0) == 0 and hc.hc_cbp_code(47) == 12 and hc.hc_cbp_code(16) == 1


def test_synth_source_is_deterministic_and_in_range(hc, venc):
    w, h = 64, 48
    y, u, v = venc.synth_frame(w, h, 5, 0x1234ABCD)
    y2, _, _ = venc.synth_frame(w, h, 5, 0x1234ABCD)
    assert np.array_equal(y, y2)
    for (x, yy) in ((0, 0), (17, 31), (63, 47)):
        assert y[yy, x] == hc.hc_synth_pixel(0, x, yy, 5, w, h, 0x1234ABCD)
    assert u[3, 9] == hc.hc_synth_pixel(1, 9, 3, 5, w, h, 0x1234ABCD)
    assert v[3, 9] == hc.hc_synth_pixel(2, 9, 3, 5, w, h, 0x1234ABCD)
    y3, _, _ = venc.synth_frame(w, h, 6, 0x1234ABCD)
    assert not np.array_equal(y, y3)
    assert 8 <= y.min() and y.max() <= 245


def test_zero_block_test_is_exactly_the_quantiser(hc, orc):
    """The transform kernels skip quantisation, dequantisation and the inverse transform of a wave whose blocks all pass
    blk_all_zero (venc_inter2.hip); restated for the host as hc_block_all_zero.  It must hold exactly when every level is
    zero under the ORACLE's inter quantiser -- on both sides of every threshold, every QP, every coefficient class."""
    L = orc.lib()
    hc.hc_block_all_zero.argtypes = [C.c_void_p, C.c_int, C.c_int]
    rng = np.random.default_rng(3)
    for qp in range(12, 52):
        # the largest |w| that still quantises to zero, per class (positions 0, 5, 1), found with the oracle's quantiser
        edge = {}
        for r in (0, 5, 1):
            t = 0
            while L.orc_quant(t + 1, qp, r, 0) == 0:
                t += 1
            edge[r] = t
            assert L.orc_quant(-t, qp, r, 0) == 0 and L.orc_quant(-(t + 1), qp, r, 0) != 0
        cls_of = [0, 2, 0, 2, 2, 1, 2, 1, 0, 2, 0, 2, 2, 1, 2, 1]
        rep = {0: 0, 1: 5, 2: 1}
        for first in (0, 1):
            for trial in range(60):
                w = np.zeros(16, np.int32)
                for r in range(16):
                    e = edge[rep[cls_of[r]]]
                    w[r] = rng.integers(-e, e + 1)              # every coefficient inside its dead zone
                assert hc.hc_block_all_zero(w.ctypes.data, qp, first) == 1
                r = int(rng.integers(first, 16))
                e = edge[rep[cls_of[r]]]
                w[r] = (e + 1) * (1 if trial & 1 else -1)       # one coefficient just outside
                assert hc.hc_block_all_zero(w.ctypes.data, qp, first) == 0
                assert L.orc_quant(int(w[r]), qp, r, 0) != 0
            if first == 1:                                      # a large DC is ignored when the DC is coded separately
                w = np.zeros(16, np.int32)
                w[0] = 9000
                assert hc.hc_block_all_zero(w.ctypes.data, qp, 1) == 1 and hc.hc_block_all_zero(w.ctypes.data, qp, 0) == 0


def test_rate_control_rule_matches_the_oracle(hc, orc):
    """spec/CODEC.md section 1a: the picture QP under rate control, product (venc_device.h, evaluated on the GPU by every
    block of the search) against the oracle, over the whole argument space that matters; and the device-side construction of
    a QP's quantiser constants against the host's."""
    L = orc.lib()
    hc.hc_rc_qp.argtypes = [C.c_int, C.c_int, C.c_int, C.c_ulonglong, C.c_int]
    rng = np.random.default_rng(5)
    for base in (12, 20, 26, 40, 51):
        for nmb in (1, 396, 8160, 32400):
            sums = [0, 1, nmb - 1, nmb, 3 * nmb, 511 * nmb, 512 * nmb, 513 * nmb, 65280 * nmb] + [int(x) for x in rng.integers(0, 65281 * nmb, 200)]
            for s in sums:
                for idr, hist in ((1, 0), (0, 0), (0, 1)):
                    assert hc.hc_rc_qp(base, idr, hist, s, nmb) == L.orc_rc_qp(base, idr, hist, s, nmb), (base, nmb, s, idr, hist)
    assert [L.orc_rc_qp(20, 0, 1, c, 1) for c in (256, 512, 1024, 2048)] == [17, 20, 22, 24]   # 2.4 QP per octave, floor
    assert L.orc_rc_qp(20, 1, 0, 0, 1) == 17 and L.orc_rc_qp(13, 1, 0, 0, 1) == 12 and L.orc_rc_qp(50, 0, 1, 65280, 1) == 51
    for qp in range(12, 52):
        assert hc.hc_make_qparams_matches(qp) == 1, qp


# ---------------------------------------------------------------- round 3: venc_lines.h
def test_deblock_line_filters_match_the_python_restatement(hc):
    """k_deblock_ls's two line filters (csrc/venc_lines.h) against the second restatement of H.264 8.7.2.3 / 8.7.2.4 in
    tests/test_oracle_deblock.py: the branching one for every strength 0 .. 4, the branch-free normal filter (the fast path of
    P pictures) for strengths 0 .. 3 -- luma and chroma, a spread of QPs, random samples and smooth ones (so that the alpha / beta
    tests pass often)."""
    from test_oracle_deblock import ALPHA, BETA, TC0, _filter_line
    hc.hc_dbk_line.argtypes = [C.c_void_p] + [C.c_int] * 5
    hc.hc_dbk_line_normal.argtypes = [C.c_void_p] + [C.c_int] * 5
    rng = np.random.default_rng(11)
    checked = filtered = 0
    for qp in (12, 20, 26, 30, 34, 40, 46, 51):
        for chroma in (0, 1):
            for bs in range(5):
                for trial in range(400):
                    if trial % 2:
                        s = rng.integers(0, 256, 8)
                    else:  # a step across the edge on a gentle slope: what the filter exists for
                        base = int(rng.integers(8, 248))
                        s = np.clip(base + rng.integers(-3, 4, 8) + np.array([0, 0, 0, 0, 1, 1, 1, 1]) * int(rng.integers(-12, 13)), 0, 255)
                    want = _filter_line([int(v) for v in s], bs, qp, bool(chroma))
                    tc0 = TC0[qp][bs - 1] if 1 <= bs <= 3 else 0
                    a = np.array(s, dtype=np.int32)
                    hc.hc_dbk_line(a.ctypes.data, bs, ALPHA[qp], BETA[qp], tc0, chroma)
                    assert list(a) == want, (qp, chroma, bs, list(s))
                    if bs < 4:
                        b = np.array(s, dtype=np.int32)
                        hc.hc_dbk_line_normal(b.ctypes.data, bs, ALPHA[qp], BETA[qp], tc0, chroma)
                        assert list(b) == want, ("branch-free", qp, chroma, bs, list(s))
                    checked += 1
                    filtered += list(s) != want
    assert filtered > checked // 20  # the filters did run


def test_follow_up_blocks_find_every_listed_block(hc):
    """k_me_fix's follow-up blocks see an item's listed 4x4 blocks as one concatenated list (csrc/venc_lines.h venc_nz_locate): every
    entry k must map to its (tile, wave) segment and its place inside -- empty segments, full ones (64), clusters, ragged counts
    per thread-segment (rpt = 4, 8, 32)."""
    hc.hc_nz_locate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint, C.c_void_p]
    hc.hc_nz_locate.restype = C.c_int
    rng = np.random.default_rng(5)
    for rpt, regions in ((4, 1000), (8, 2040), (32, 8160), (8, 37)):
        for kind in ("sparse", "clustered", "full", "empty"):
            cnt = np.zeros(256 * rpt, dtype=np.uint32)
            if kind == "sparse":
                cnt[:regions] = rng.integers(0, 8, regions) * (rng.random(regions) < 0.4)
            elif kind == "clustered":
                lo = int(rng.integers(0, max(1, regions - 40)))
                cnt[lo:lo + 40] = 64
                cnt[:regions] = np.minimum(64, cnt[:regions] + (rng.random(regions) < 0.05) * rng.integers(1, 64, regions))
            elif kind == "full":
                cnt[:regions] = 64
            per_thread = cnt.reshape(256, rpt).sum(axis=1)
            part = np.concatenate([[0], np.cumsum(per_thread)]).astype(np.uint32)
            cnt32 = (cnt.reshape(-1, 4) * np.array([1, 1 << 8, 1 << 16, 1 << 24], dtype=np.uint64)).sum(axis=1).astype(np.uint32)
            total = int(part[256])
            want = [(r, i) for r in range(256 * rpt) for i in range(int(cnt[r]))]
            assert len(want) == total
            idx = C.c_uint(0)
            ks = range(total) if total <= 3000 else sorted(set([0, total - 1] + [int(v) for v in rng.integers(0, total, 3000)]))
            for k in ks:
                r = hc.hc_nz_locate(part.ctypes.data, cnt32.ctypes.data, rpt, k, C.byref(idx))
                assert (r, idx.value) == want[k], (rpt, kind, k)


# ---------------------------------------------------------------- round 4: the P pictures' entries (venc_cavlc_p.hip)
def test_register_head_bit_sink_equals_the_plain_sink(hc):
    """venc_bitsink4 (four words in registers, the rest spilled at their word index; what a lane of k_cavlc_p2 codes an entry
    into) against venc_bitsink, word for word: a prefix of 0 .. 32 bits that moves every later code across the word boundaries,
    two Exp-Golomb codes and one residual block of any density, up to strings beyond 128 bits."""
    hc.hc_cavlc_block_mem4.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_void_p]
    rng = np.random.default_rng(4)
    words = np.zeros(64, np.uint32)
    spilled = 0
    for max_coef, first in ((16, 0), (15, 1), (4, 0)):
        for nC in ((-1,) if max_coef == 4 else (0, 2, 4, 8)):
            for density in (0.0, 0.2, 0.7, 1.0):
                for big in (False, True):
                    for plen in (0, 1, 7, 31, 32):
                        blk = _rand_block(rng, max_coef, density, big)
                        c = np.zeros(16, np.int16)
                        c[first:first + max_coef] = blk[:max_coef]
                        words[:] = 0
                        pcode = int(rng.integers(0, 1 << plen)) if plen else 0
                        bits = hc.hc_cavlc_block_mem4(c.ctypes.data, first, max_coef, nC, pcode, plen, words.ctypes.data)
                        assert bits > 0, (bits, max_coef, nC, blk)
                        nw = (bits + 31) // 32
                        assert np.array_equal(words[:nw], words[32:32 + nw]), (max_coef, nC, plen, blk)
                        assert not words[nw:32].any()  # nothing beyond the string (the pack kernel ORs whole words)
                        spilled += nw > 4
    assert spilled > 20


def test_search_key_orders_like_the_rank_table():
    """k_me_pool4 keeps a macroblock's running best as ONE dword, SAD << 13 | (|dx| + |dy|) << 7 | cy << 1 | (dx > 0)
    (venc_me4.hip me4_key), where the other search kernels and the oracle order by (SAD, rank), rank = index of the
    displacement in the list sorted by (|dx| + |dy|, dy, dx) (venc_hip.hip, the table behind venc_bufs.me_rank).  Python
    model of both: the tie-break field must sort all 33 x 33 displacements exactly as the rank does, be unique, fit 13 bits,
    and give the displacement back (me4_key_disp)."""
    R = 16
    disps = sorted(((abs(dx) + abs(dy), dy, dx) for dy in range(-R, R + 1) for dx in range(-R, R + 1)))
    rank = {(dx, dy): i for i, (_, dy, dx) in enumerate(disps)}

    def key(sad, cx, cy):
        l1 = abs(cx - 16) + abs(cy - 16)
        return (sad << 13) | (l1 << 7) | (cy << 1) | (1 if cx > 16 else 0)

    def disp(k):
        cy, l1 = (k >> 1) & 63, (k >> 7) & 63
        a = l1 - abs(cy - 16)
        return (16 + a if k & 1 else 16 - a), cy

    ties = {}
    for (dx, dy), r in rank.items():
        t = key(0, dx + 16, dy + 16)
        assert t < (1 << 13) and t not in ties
        ties[t] = r
        assert disp(key(12345, dx + 16, dy + 16)) == (dx + 16, dy + 16)
    assert [ties[t] for t in sorted(ties)] == list(range(33 * 33))
    assert key(255 * 256, 32, 32) < (1 << 32) and key(7, 0, 0) < key(8, 16, 16)  # the SAD dominates; the largest key fits
