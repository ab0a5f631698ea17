"""Structural checks of the ORACLE's constant tables (oracle/orc_tables.h) -- the test that header promises.

The encoder and the decoder of the oracle share orc_tables.h, so an encode -> decode round trip cannot see a typo
in a table.  What can: (1) the code-theoretic structure every H.264 VLC table must have (prefix-free, Kraft sum
complete up to the codewords the standard leaves unused), (2) the arithmetic identities of the quantiser tables,
(3) agreement, entry by entry, with the PRODUCT's tables (csrc/venc_tables.h), which were written independently in a
different packed form, and (4) known answers published outside this repository: the three worked CAVLC examples of
I. Richardson, "H.264 and MPEG-4 Video Compression" (Wiley 2003), section 6.4.11.

PARITY UNPINNED all the same: the reference (arkanis/video-encoder) holds no fixture for this path (SURVEY.md
section 8c); these checks pin the tables to the published standard, not to libx264's output.
"""
import ctypes as C
import os
import re
from fractions import Fraction

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tables():
    """{name: nested list} of every `static const ... name[..] = {...};` in orc_tables.h (parsed as text)."""
    src = open(os.path.join(ROOT, "oracle", "orc_tables.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"static const \w+ (\w+)((?:\[[^\]]*\])+)\s*=\s*(\{.*?\});", src, flags=re.S):
        body = m.group(3).replace("{", "[").replace("}", "]")
        out[m.group(1)] = eval(body, {"__builtins__": {}})  # digits, commas and brackets only
    return out


T = _tables()


def _prefix_code(entries, unused=Fraction(0)):
    """entries: [(len, bits)], len > 0.  Prefix-free, and Kraft sum = 1 - unused."""
    k = sum(Fraction(1, 1 << l) for l, _ in entries)
    for i, (l1, b1) in enumerate(entries):
        assert b1 < (1 << l1), "code value does not fit its length"
        for j, (l2, b2) in enumerate(entries):
            if i != j and l2 >= l1:
                assert (b2 >> (l2 - l1)) != b1, "prefix clash %s %s" % ((l1, b1), (l2, b2))
    assert k == 1 - unused, (k, unused)


def test_every_table_was_found():
    for name in ("orc_zigzag4x4", "orc_blk_x", "orc_blk_y", "orc_quant_mf", "orc_dequant_v", "orc_chroma_qp",
                 "orc_golomb_to_inter_cbp", "orc_coeff_token_len", "orc_coeff_token_bits",
                 "orc_chroma_dc_coeff_token_len", "orc_chroma_dc_coeff_token_bits", "orc_total_zeros_len",
                 "orc_total_zeros_bits", "orc_chroma_dc_total_zeros_len", "orc_chroma_dc_total_zeros_bits", "orc_run_len",
                 "orc_run_bits"):
        assert name in T, name


def test_scan_and_block_order():
    zz = T["orc_zigzag4x4"]
    assert sorted(zz) == list(range(16))
    # a zig-zag walks to a neighbour (king move) every step and visits anti-diagonals in order
    for a, b in zip(zz, zz[1:]):
        assert max(abs(a // 4 - b // 4), abs(a % 4 - b % 4)) == 1
    diag = [r // 4 + r % 4 for r in zz]
    assert diag == sorted(diag)
    pos = {(T["orc_blk_x"][i], T["orc_blk_y"][i]) for i in range(16)}
    assert pos == {(x, y) for x in range(4) for y in range(4)}
    for i in range(16):  # figure 6-10: bits 0/2 of the index select x, bits 1/3 select y
        assert T["orc_blk_x"][i] == (i & 1) | ((i >> 1) & 2) and T["orc_blk_y"][i] == ((i >> 1) & 1) | ((i >> 2) & 2)


def test_quantiser_tables_are_mutual_inverses():
    """8.5.9 / 8.5.12: MF * V * (norm of the basis function pair) = 2^21 up to rounding, for every qp % 6 and class
    (norms 16, 25, 20 for the classes (0,0) / (1,1) / mixed of the 4x4 core transform)."""
    mf, v = T["orc_quant_mf"], T["orc_dequant_v"]
    for q in range(6):
        for cls, norm in enumerate((16, 25, 20)):
            prod = mf[q][cls] * v[q][cls] * norm
            assert abs(prod - (1 << 21)) <= (1 << 21) // 400, (q, cls, prod)
    # step size doubles every 6: V is increasing in qp % 6 and V[5] < 2 * V[0]
    for cls in range(3):
        col = [v[q][cls] for q in range(6)]
        assert col == sorted(col) and col[5] < 2 * col[0]


def test_chroma_qp_map():
    t = T["orc_chroma_qp"]
    assert t[:30] == list(range(30)) and t[51] == 39 and all(0 <= b - a <= 1 for a, b in zip(t, t[1:]))
    assert [t[i] for i in (30, 34, 39, 43, 47)] == [29, 32, 35, 37, 38]  # table 8-15 spot values


def test_inter_cbp_map_is_a_permutation_with_the_standard_anchors():
    g = T["orc_golomb_to_inter_cbp"]
    assert sorted(g) == list(range(48))
    # table 9-4, column Inter: codeNum 0 -> cbp 0, 1 -> 16, 2 -> 1, 11 -> 15, 12 -> 47
    assert (g[0], g[1], g[2], g[11], g[12]) == (0, 16, 1, 15, 47)


def test_vlc_tables_are_complete_prefix_codes():
    ln, bt = T["orc_coeff_token_len"], T["orc_coeff_token_bits"]
    for t, unused in enumerate((Fraction(1, 1 << 15), Fraction(1, 1 << 13), Fraction(1, 1 << 10), None)):
        ent = [(ln[t][i], bt[t][i]) for i in range(68) if ln[t][i]]
        # TrailingOnes <= min(3, TotalCoeff): 62 legal combinations
        assert len(ent) == 62 and all(ln[t][4 * tc + t1] == 0 for tc in range(3) for t1 in range(tc + 1, 4))
        if unused is not None:
            _prefix_code(ent, unused)  # only the all-zero codeword of the longest length is unused
        else:
            assert all(l == 6 for l, _ in ent) and len({b for _, b in ent}) == 62
    cl, cb = T["orc_chroma_dc_coeff_token_len"], T["orc_chroma_dc_coeff_token_bits"]
    _prefix_code([(cl[i], cb[i]) for i in range(20) if cl[i]])
    tl, tb = T["orc_total_zeros_len"], T["orc_total_zeros_bits"]
    for k in range(15):  # tzVlcIndex k + 1: total_zeros 0 .. 15 - k
        assert len(tl[k]) == len(tb[k]) == 16 - k if k else len(tl[k]) == 16
        _prefix_code(list(zip(tl[k], tb[k])), Fraction(1, 1 << 9) if k == 0 else Fraction(0))
    dl, db = T["orc_chroma_dc_total_zeros_len"], T["orc_chroma_dc_total_zeros_bits"]
    for k in range(3):
        _prefix_code([(dl[k][i], db[k][i]) for i in range(4 - k)])
    rl, rb = T["orc_run_len"], T["orc_run_bits"]
    for k in range(7):
        assert len(rl[k]) == (k + 2 if k < 6 else 15)
        _prefix_code(list(zip(rl[k], rb[k])), Fraction(1, 1 << 11) if k == 6 else Fraction(0))


@pytest.fixture(scope="module")
def hc(venc):
    L = C.CDLL(venc.LIB_HOSTCHECK)
    L.hc_table.argtypes = [C.c_int, C.POINTER(C.c_ushort)]
    L.hc_cbp_code.argtypes = [C.c_int]
    L.hc_chroma_qp.argtypes = [C.c_int]
    L.hc_zigzag.argtypes = [C.c_int]
    L.hc_cavlc_block.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    return L


def test_oracle_tables_equal_the_product_tables(hc):
    """Two independently written copies of the standard's tables (oracle: len[] + bits[] arrays; product: packed
    len << 8 | bits, cbp -> codeNum instead of codeNum -> cbp, zig-zag as a 64-bit constant): a typo in either shows."""
    buf = (C.c_ushort * 272)()

    def tab(which):
        n = hc.hc_table(which, buf)
        return [(buf[i] >> 8, buf[i] & 0xff) for i in range(n)]
    ct = tab(0)
    for t in range(4):
        for i in range(68):
            assert ct[68 * t + i] == (T["orc_coeff_token_len"][t][i], T["orc_coeff_token_bits"][t][i] if T["orc_coeff_token_len"][t][i] else 0) \
                or (ct[68 * t + i][0] == 0 and T["orc_coeff_token_len"][t][i] == 0), (t, i)
    cdc = tab(1)
    for i in range(20):
        if T["orc_chroma_dc_coeff_token_len"][i]:
            assert cdc[i] == (T["orc_chroma_dc_coeff_token_len"][i], T["orc_chroma_dc_coeff_token_bits"][i]), i
        else:
            assert cdc[i][0] == 0
    tz = tab(2)
    for k in range(15):
        for i, (l, b) in enumerate(zip(T["orc_total_zeros_len"][k], T["orc_total_zeros_bits"][k])):
            assert tz[16 * k + i] == (l, b), (k, i)
    tzc = tab(3)
    for k in range(3):
        for i in range(4 - k):
            assert tzc[4 * k + i] == (T["orc_chroma_dc_total_zeros_len"][k][i], T["orc_chroma_dc_total_zeros_bits"][k][i])
    rb = tab(4)
    for k in range(7):
        for i, (l, b) in enumerate(zip(T["orc_run_len"][k], T["orc_run_bits"][k])):
            assert rb[16 * k + i] == (l, b), (k, i)
    for code, cbp in enumerate(T["orc_golomb_to_inter_cbp"]):
        assert hc.hc_cbp_code(cbp) == code
    assert [hc.hc_chroma_qp(q) for q in range(52)] == T["orc_chroma_qp"]
    assert [hc.hc_zigzag(i) for i in range(16)] == T["orc_zigzag4x4"]


# Richardson, "H.264 and MPEG-4 Video Compression", 6.4.11, examples 1-3 (4x4 blocks, nC in 0..1): levels in zig-zag
# order -> transmitted bitstream.  Published known answers, independent of this repository.
KNOWN_CAVLC = [
    ([0, 3, 0, 1, -1, -1, 0, 1], "000010001110010111101101"),
    ([-2, 4, 3, -3, 0, 0, -1], "000000011010001001000010111001100"),
    ([0, 0, 0, 1, 0, 1, 0, 0, 0, -1], "0001110001110010"),
]


@pytest.mark.parametrize("levels,want", KNOWN_CAVLC)
def test_cavlc_published_known_answers_oracle(orc, levels, want):
    c = np.zeros(16, np.int16)
    c[:len(levels)] = levels
    out = np.zeros(64, np.uint8)
    n = orc.lib().orc_cavlc_block_bits(c.ctypes.data, 16, 0, out.ctypes.data, 64)
    assert "".join("{:08b}".format(b) for b in out)[:n] == want


@pytest.mark.parametrize("levels,want", KNOWN_CAVLC)
def test_cavlc_published_known_answers_product_device_function(hc, levels, want):
    """The product's per-lane CAVLC coder (csrc/venc_device.h, compiled for the host by hostcheck.cpp)."""
    c = np.zeros(16, np.int32)
    c[:len(levels)] = levels
    out = np.zeros(64, np.uint8)
    n = hc.hc_cavlc_block(c.ctypes.data, 0, 16, 0, out.ctypes.data, 64)
    assert "".join("{:08b}".format(b) for b in out)[:n] == want


def test_plane_prediction_reproduces_a_plane():
    """Intra16x16 plane prediction (8.3.3.4) of a picture that IS a plane: away from the first row and column of macroblocks the
    predictor is chosen and leaves (almost) nothing to code; the decoder agrees (PARITY UNPINNED: no external vector)"""
    import numpy as np
    from oracle import orc_py
    w, h = 96, 80
    yy, xx = np.mgrid[0:h, 0:w]
    y = (20 + 1.5 * xx + 0.75 * yy).astype(np.uint8)
    c = np.full((h // 2, w // 2), 128, np.uint8)
    a = orc_py.Encoder(w, h, qp=20, n_slices=1, keyint=1, intra_modes=1)
    b = orc_py.Encoder(w, h, qp=20, n_slices=1, keyint=1, intra_modes=2)
    d = orc_py.Decoder()
    fa, fb = a.encode(y, c, c), b.encode(y, c, c)
    assert len(fb.data) < len(fa.data)
    lv = b.levels()
    inner = [my * b.mbw + mx for my in range(1, b.mbh) for mx in range(1, b.mbw)]
    assert sum(int(np.count_nonzero(lv[m, :17])) for m in inner) <= len(inner)   # at most a stray level per macroblock
    assert d.decode(fb.data) == 1
    for p, q in zip(b.recon(), d.planes()):
        assert np.array_equal(p, q)
    assert np.abs(b.recon()[0][:h, :w].astype(int) - y.astype(int)).max() <= 6
