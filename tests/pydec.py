"""pydec.py -- an INDEPENDENT H.264 bitstream decoder for the tests (numpy only).

Why it exists: the oracle encoder and the oracle decoder (oracle/orc_enc.c, orc_dec.c) share
orc_tables.h, orc_interp.h, orc_deblock.h and orc_intra4.h, so an encode -> decode round trip cannot see a misreading of the
standard that sits in one of those headers.  This decoder shares NOTHING with oracle/ -- no import, no table, no helper.  It is
written from ITU-T H.264 (03/2005 numbering) as the author knows it: clause 7.3 syntax, 8.3 intra prediction, 8.4 inter
prediction, 8.5 transform, 8.7 deblocking, 9.1 Exp-Golomb, 9.2 CAVLC; the VLC tables below are typed in from the standard's
Tables 9-4, 9-5, 9-7 .. 9-10 in the length / code-word form, the filter tables from 8-16 / 8-17.  It is a general decoder of
the subset the streams use, not a mirror of the encoder: Constrained Baseline, frame macroblocks, CAVLC, one reference picture,
I slices with Intra16x16 (four modes) and Intra4x4 (nine modes), P slices with P_L0_16x16 and P_Skip, quarter-sample motion
compensation, disable_deblocking_filter_idc 0 / 1 / 2, slice_qp_delta, frame cropping.  Anything else raises NotImplementedError.

PARITY UNPINNED all the same: no published conformance stream exists in this image, so this is a second, independent reading
of the standard by the same project -- it removes the shared-header blind spot, it does not replace a conformance suite.
tests/test_pydec.py feeds it the oracle encoder's streams and compares its pictures with the encoder's reconstruction.
"""
import numpy as np


# ----------------------------------------------------------------------------------------------------------- bits
class Bits:
    def __init__(self, data):
        self.b = np.unpackbits(np.frombuffer(bytes(data), dtype=np.uint8))
        self.n = len(self.b)
        self.p = 0
        # rbsp_trailing_bits: the last 1 bit is the stop bit
        nz = np.nonzero(self.b)[0]
        self.end = int(nz[-1]) if len(nz) else 0

    def u(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | int(self.b[self.p])
            self.p += 1
        return v

    def ue(self):
        z = 0
        while self.b[self.p] == 0:
            z += 1
            self.p += 1
        self.p += 1
        return (1 << z) - 1 + (self.u(z) if z else 0)

    def se(self):
        k = self.ue()
        return (k + 1) >> 1 if k & 1 else -(k >> 1)

    def more_rbsp_data(self):
        return self.p < self.end

    def peek(self, n):
        v = 0
        for i in range(n):
            v = (v << 1) | (int(self.b[self.p + i]) if self.p + i < self.n else 0)
        return v


def split_nals(data):
    """[(nal_ref_idc, nal_unit_type, rbsp bytes)] of a sequence of NAL units, each with a 4-byte big-endian length prefix"""
    out, o = [], 0
    while o < len(data):
        n = int.from_bytes(data[o:o + 4], "big")
        nal = data[o + 4:o + 4 + n]
        o += 4 + n
        body, z, rb = nal[1:], 0, bytearray()
        for c in body:  # emulation_prevention_three_byte
            if z >= 2 and c == 3:
                z = 0
                continue
            rb.append(c)
            z = z + 1 if c == 0 else 0
        out.append(((nal[0] >> 5) & 3, nal[0] & 31, bytes(rb)))
    assert o == len(data)
    return out


# ----------------------------------------------------------------------------------------------------------- tables
# Table 9-5, coeff_token: [nC class][TotalCoeff][TrailingOnes] -> (length, code word); typed in as length and value
_CT_LEN = [
    [[1], [6, 2], [8, 6, 3], [9, 8, 7, 5], [10, 9, 8, 6], [11, 10, 9, 7], [13, 11, 10, 8], [13, 13, 11, 9], [13, 13, 13, 10],
     [14, 14, 13, 11], [14, 14, 14, 13], [15, 15, 14, 14], [15, 15, 15, 14], [16, 15, 15, 15], [16, 16, 16, 15], [16, 16, 16, 16],
     [16, 16, 16, 16]],
    [[2], [6, 2], [6, 5, 3], [7, 6, 6, 4], [8, 6, 6, 4], [8, 7, 7, 5], [9, 8, 8, 6], [11, 9, 9, 6], [11, 11, 11, 7], [12, 11, 11, 9],
     [12, 12, 12, 11], [12, 12, 12, 11], [13, 13, 13, 12], [13, 13, 13, 13], [13, 14, 13, 13], [14, 14, 14, 13], [14, 14, 14, 14]],
    [[4], [6, 4], [6, 5, 4], [6, 5, 5, 4], [7, 5, 5, 4], [7, 5, 5, 4], [7, 6, 6, 4], [7, 6, 6, 4], [8, 7, 7, 5], [8, 8, 7, 6],
     [9, 8, 8, 7], [9, 9, 8, 8], [9, 9, 9, 8], [10, 9, 9, 9], [10, 10, 10, 10], [10, 10, 10, 10], [10, 10, 10, 10]],
]
_CT_BITS = [
    [[1], [5, 1], [7, 4, 1], [7, 6, 5, 3], [7, 6, 5, 3], [7, 6, 5, 4], [15, 6, 5, 4], [11, 14, 5, 4], [8, 10, 13, 4], [15, 14, 9, 4],
     [11, 10, 13, 12], [15, 14, 9, 12], [11, 10, 13, 8], [15, 1, 9, 12], [11, 14, 13, 8], [7, 10, 9, 12], [4, 6, 5, 8]],
    [[3], [11, 2], [7, 7, 3], [7, 10, 9, 5], [7, 6, 5, 4], [4, 6, 5, 6], [7, 6, 5, 8], [15, 6, 5, 4], [11, 14, 13, 4], [15, 10, 9, 4],
     [11, 14, 13, 12], [8, 10, 9, 8], [15, 14, 13, 12], [11, 10, 9, 12], [7, 11, 6, 8], [9, 8, 10, 1], [7, 6, 5, 4]],
    [[15], [15, 14], [11, 15, 13], [8, 12, 14, 12], [15, 10, 11, 11], [11, 8, 9, 10], [9, 14, 13, 9], [8, 10, 9, 8], [15, 14, 13, 13],
     [11, 14, 10, 12], [15, 10, 13, 12], [11, 14, 9, 12], [8, 10, 13, 8], [13, 7, 9, 12], [9, 12, 11, 10], [5, 8, 7, 6], [1, 4, 3, 2]],
]
# nC = -1 (chroma DC, 4:2:0)
_CDC_LEN = [[2], [6, 1], [6, 6, 3], [6, 7, 7, 6], [6, 8, 8, 7]]
_CDC_BITS = [[1], [7, 1], [4, 6, 1], [3, 3, 2, 5], [2, 3, 2, 0]]
# Tables 9-7 / 9-8, total_zeros for 4x4 blocks: [tzVlcIndex - 1][total_zeros]
_TZ_LEN = [
    [1, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 9], [3, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 6, 6, 6, 6], [4, 3, 3, 3, 4, 4, 3, 3, 4, 5, 5, 6, 5, 6],
    [5, 3, 4, 4, 3, 3, 3, 4, 3, 4, 5, 5, 5], [4, 4, 4, 3, 3, 3, 3, 3, 4, 5, 4, 5], [6, 5, 3, 3, 3, 3, 3, 3, 4, 3, 6], [6, 5, 3, 3, 3, 2, 3, 4, 3, 6],
    [6, 4, 5, 3, 2, 2, 3, 3, 6], [6, 6, 4, 2, 2, 3, 2, 5], [5, 5, 3, 2, 2, 2, 4], [4, 4, 3, 3, 1, 3], [4, 4, 2, 1, 3], [3, 3, 1, 2], [2, 2, 1], [1, 1]]
_TZ_BITS = [
    [1, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 1], [7, 6, 5, 4, 3, 5, 4, 3, 2, 3, 2, 3, 2, 1, 0], [5, 7, 6, 5, 4, 3, 4, 3, 2, 3, 2, 1, 1, 0],
    [3, 7, 5, 4, 6, 5, 4, 3, 3, 2, 2, 1, 0], [5, 4, 3, 7, 6, 5, 4, 3, 2, 1, 1, 0], [1, 1, 7, 6, 5, 4, 3, 2, 1, 1, 0], [1, 1, 5, 4, 3, 3, 2, 1, 1, 0],
    [1, 1, 1, 3, 3, 2, 2, 1, 0], [1, 0, 1, 3, 2, 1, 1, 1], [1, 0, 1, 3, 2, 1, 1], [0, 1, 1, 2, 1, 3], [0, 1, 1, 1, 1], [0, 1, 1, 1], [0, 1, 1], [0, 1]]
# Table 9-9 (a), total_zeros for chroma DC 2x2
_TZC_LEN = [[1, 2, 3, 3], [1, 2, 2], [1, 1]]
_TZC_BITS = [[1, 1, 1, 0], [1, 1, 0], [1, 0]]
# Table 9-10, run_before: [min(zerosLeft, 7) - 1][run_before]
_RUN_LEN = [[1, 1], [1, 2, 2], [2, 2, 2, 2], [2, 2, 2, 3, 3], [2, 2, 3, 3, 3, 3], [2, 3, 3, 3, 3, 3, 3], [3, 3, 3, 3, 3, 3, 3, 4, 5, 6, 7, 8, 9, 10, 11]]
_RUN_BITS = [[1, 0], [1, 1, 0], [3, 2, 1, 0], [3, 2, 1, 1, 0], [3, 2, 3, 2, 1, 0], [3, 0, 1, 3, 2, 5, 4], [7, 6, 5, 4, 3, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1]]
# Table 9-4, codeNum -> coded_block_pattern (chroma_format_idc 1): Intra_4x4 column, Inter column
_CBP_INTRA = [47, 31, 15, 0, 23, 27, 29, 30, 7, 11, 13, 14, 39, 43, 45, 46, 16, 3, 5, 10, 12, 19, 21, 26, 28, 35, 37, 42, 44, 1, 2, 4, 8, 17, 18, 20, 24, 6, 9, 22,
              25, 32, 33, 34, 36, 40, 38, 41]
_CBP_INTER = [0, 16, 1, 2, 4, 8, 32, 3, 5, 10, 12, 15, 47, 7, 11, 13, 14, 6, 9, 31, 35, 37, 42, 44, 33, 34, 36, 40, 39, 43, 45, 46, 17, 18, 20, 24, 19, 21, 26, 28,
              23, 27, 29, 30, 22, 25, 38, 41]
# Table 8-15: QPc as a function of qPI
_QPC = list(range(30)) + [29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39]
# 8.5.9: LevelScale(m, i, j) = 16 x normAdjust4x4: v for positions (even, even), (odd, odd), others
_V = [(10, 16, 13), (11, 18, 14), (13, 20, 16), (14, 23, 18), (16, 25, 20), (18, 29, 23)]
_ZIGZAG = [0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15]  # 4x4 frame scan: index -> raster position
# Table 8-16 alpha' and beta' by indexA / indexB; Table 8-17 tC0' by indexA and bS
_ALPHA = [0] * 16 + [4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255]
_BETA = [0] * 16 + [2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18]
_TC0 = [(0, 0, 0)] * 17 + [(0, 0, 1)] * 4 + [(0, 1, 1)] * 2 + [(1, 1, 1)] * 4 + [(1, 1, 2)] * 4 + [(1, 2, 3)] * 2 + [(2, 2, 3), (2, 2, 4), (2, 3, 4), (2, 3, 4), (3, 3, 5),
        (3, 4, 6), (3, 4, 6), (4, 5, 7), (4, 5, 8), (4, 6, 9), (5, 7, 10), (6, 8, 11), (6, 8, 13), (7, 10, 14), (8, 11, 16), (9, 12, 18), (10, 13, 20), (11, 15, 23), (13, 17, 25)]
assert len(_ALPHA) == 52 and len(_BETA) == 52 and len(_TC0) == 52 and len(_QPC) == 52


def _vlc_map(lens, bits):
    """{(length, code): index} of one VLC table row set; checks that the code is prefix-free"""
    m = {}
    for i, (ln, c) in enumerate(zip(lens, bits)):
        assert c < (1 << ln)
        m[(ln, c)] = i
    words = sorted("{:0{}b}".format(c, ln) for (ln, c) in m)
    for a, b in zip(words, words[1:]):
        assert not b.startswith(a), ("not prefix-free", a, b)
    return m


def _coeff_token_maps():
    maps = []
    for cls in range(3):
        lens, bits, idx = [], [], []
        for tc in range(17):
            for t1 in range(len(_CT_LEN[cls][tc])):
                lens.append(_CT_LEN[cls][tc][t1]), bits.append(_CT_BITS[cls][tc][t1]), idx.append((tc, t1))
        mm = _vlc_map(lens, bits)
        maps.append({k: idx[v] for k, v in mm.items()})
    # 8 <= nC: 6-bit fixed length: xxxxyy = TotalCoeff - 1, TrailingOnes; 0000 11 = no coefficients
    flc = {(6, 3): (0, 0)}
    for tc in range(1, 17):
        for t1 in range(min(tc, 3) + 1):
            flc[(6, ((tc - 1) << 2) | t1)] = (tc, t1)
    maps.append(flc)
    lens, bits, idx = [], [], []
    for tc in range(5):
        for t1 in range(len(_CDC_LEN[tc])):
            lens.append(_CDC_LEN[tc][t1]), bits.append(_CDC_BITS[tc][t1]), idx.append((tc, t1))
    mm = _vlc_map(lens, bits)
    maps.append({k: idx[v] for k, v in mm.items()})
    return maps


_CT_MAPS = _coeff_token_maps()
_TZ_MAPS = [_vlc_map(a, b) for a, b in zip(_TZ_LEN, _TZ_BITS)]
_TZC_MAPS = [_vlc_map(a, b) for a, b in zip(_TZC_LEN, _TZC_BITS)]
_RUN_MAPS = [_vlc_map(a, b) for a, b in zip(_RUN_LEN, _RUN_BITS)]


def _read_vlc(br, m, maxlen=16):
    code = 0
    for ln in range(1, maxlen + 1):
        code = (code << 1) | int(br.b[br.p + ln - 1])
        if (ln, code) in m:
            br.p += ln
            return m[(ln, code)]
    raise ValueError("no VLC code word matches at bit %d" % br.p)


def residual_block_cavlc(br, nC, max_num_coeff):
    """9.2: the block's coefficients in scan order (length max_num_coeff) and TotalCoeff"""
    cls = 4 if nC < 0 else 0 if nC < 2 else 1 if nC < 4 else 2 if nC < 8 else 3
    total, t1s = _read_vlc(br, _CT_MAPS[cls])
    coef = [0] * max_num_coeff
    if total == 0:
        return coef, 0
    suffix_len = 1 if (total > 10 and t1s < 3) else 0
    level = []
    for i in range(total):
        if i < t1s:
            level.append(1 - 2 * br.u(1))
            continue
        prefix = 0
        while br.u(1) == 0:
            prefix += 1
        code = min(15, prefix) << suffix_len
        if suffix_len > 0 or prefix >= 14:
            size = suffix_len
            if prefix == 14 and suffix_len == 0:
                size = 4
            if prefix >= 15:
                size = prefix - 3
            if size:
                code += br.u(size)
        if prefix >= 15 and suffix_len == 0:
            code += 15
        if prefix >= 16:
            code += (1 << (prefix - 3)) - 4096
        if i == t1s and t1s < 3:
            code += 2
        lv = (code + 2) >> 1 if code % 2 == 0 else (-code - 1) >> 1
        level.append(lv)
        if suffix_len == 0:
            suffix_len = 1
        if abs(lv) > (3 << (suffix_len - 1)) and suffix_len < 6:
            suffix_len += 1
    zeros_left = 0
    if total < max_num_coeff:
        zeros_left = _read_vlc(br, _TZC_MAPS[total - 1] if max_num_coeff == 4 else _TZ_MAPS[total - 1])
    run = [0] * total
    for i in range(total - 1):
        if zeros_left > 0:
            run[i] = _read_vlc(br, _RUN_MAPS[min(zeros_left, 7) - 1])
            zeros_left -= run[i]
    run[total - 1] = zeros_left
    pos = -1
    for i in range(total - 1, -1, -1):
        pos += run[i] + 1
        coef[pos] = level[i]
    return coef, total


# ----------------------------------------------------------------------------------------------------------- transforms
def _idct4(d):
    """8.5.12: d = 4x4 list of lists (already scaled); returns the residual (with the final (x + 32) >> 6)"""
    f = [[0] * 4 for _ in range(4)]
    for i in range(4):
        e0, e1 = d[i][0] + d[i][2], d[i][0] - d[i][2]
        e2, e3 = (d[i][1] >> 1) - d[i][3], d[i][1] + (d[i][3] >> 1)
        f[i] = [e0 + e3, e1 + e2, e1 - e2, e0 - e3]
    r = [[0] * 4 for _ in range(4)]
    for j in range(4):
        g0, g1 = f[0][j] + f[2][j], f[0][j] - f[2][j]
        g2, g3 = (f[1][j] >> 1) - f[3][j], f[1][j] + (f[3][j] >> 1)
        r[0][j], r[1][j], r[2][j], r[3][j] = (g0 + g3 + 32) >> 6, (g1 + g2 + 32) >> 6, (g1 - g2 + 32) >> 6, (g0 - g3 + 32) >> 6
    return r


def _level_scale(qp, pos):
    i, j = pos >> 2, pos & 3
    v = _V[qp % 6]
    return v[0] if (i % 2 == 0 and j % 2 == 0) else v[1] if (i % 2 == 1 and j % 2 == 1) else v[2]


def _scale_block(coef16, qp, dc=None):
    """coef16 in scan order -> scaled 4x4 d (8.5.12.1); dc: an already scaled DC that replaces position 0"""
    d = [[0] * 4 for _ in range(4)]
    for k in range(16):
        pos = _ZIGZAG[k]
        if k == 0 and dc is not None:
            d[0][0] = dc
        else:
            # qP >= 24: (c * LevelScale) << (qP / 6 - 4) with LevelScale = 16 v; below: rounding shift -- both equal (c * v) << (qP / 6)
            d[pos >> 2][pos & 3] = (coef16[k] * _level_scale(qp, pos)) << (qp // 6)
    return d


def _clip1(x):
    return 0 if x < 0 else 255 if x > 255 else x


# ----------------------------------------------------------------------------------------------------------- decoder
I16, I4, PL0, PSKIP = 1, 2, 3, 4


class Decoder:
    def __init__(self):
        self.sps = self.pps = None
        self.ref = None   # (Y, U, V) of the previous decoded picture, after the loop filter
        self.out = None

    # -- parameter sets
    def _sps(self, rb):
        br = Bits(rb)
        s = {"profile": br.u(8)}
        br.u(8)
        s["level"] = br.u(8)
        br.ue()
        if s["profile"] in (100, 110, 122, 244, 44, 83, 86, 118, 128):
            raise NotImplementedError("high profiles")
        s["log2_max_frame_num"] = br.ue() + 4
        s["poc_type"] = br.ue()
        if s["poc_type"] == 0:
            s["log2_max_poc_lsb"] = br.ue() + 4
        elif s["poc_type"] == 1:
            raise NotImplementedError("pic_order_cnt_type 1")
        br.ue()  # max_num_ref_frames
        br.u(1)
        s["mbw"] = br.ue() + 1
        s["mbh"] = br.ue() + 1
        if br.u(1) != 1:
            raise NotImplementedError("field coding")
        br.u(1)
        s["crop"] = (0, 0, 0, 0)
        if br.u(1):
            s["crop"] = (br.ue(), br.ue(), br.ue(), br.ue())  # left, right, top, bottom in chroma sample units
        self.sps = s

    def _pps(self, rb):
        br = Bits(rb)
        p = {}
        br.ue(), br.ue()
        if br.u(1):
            raise NotImplementedError("CABAC")
        p["poc_present"] = br.u(1)
        if br.ue() != 0:
            raise NotImplementedError("slice groups")
        br.ue(), br.ue()
        if br.u(1) or br.u(2):
            raise NotImplementedError("weighted prediction")
        p["init_qp"] = 26 + br.se()
        br.se()
        p["chroma_qp_offset"] = br.se()
        p["dbk_present"] = br.u(1)
        if br.u(1):
            raise NotImplementedError("constrained_intra_pred")
        if br.u(1):
            raise NotImplementedError("redundant pictures")
        self.pps = p

    # -- a picture
    def size(self):
        s = self.sps
        return s["mbw"] * 16 - 2 * (s["crop"][0] + s["crop"][1]), s["mbh"] * 16 - 2 * (s["crop"][2] + s["crop"][3])

    def decode(self, data):
        """all NAL units of ONE access unit (4-byte length prefixes); returns the decoded picture (coded size) as (Y, U, V)"""
        first = True
        for ref_idc, typ, rb in split_nals(data):
            if typ == 7:
                self._sps(rb)
            elif typ == 8:
                self._pps(rb)
            elif typ in (1, 5):
                if first:
                    self._begin_picture()
                    first = False
                self._slice(ref_idc, typ, rb)
            elif typ in (6, 9, 12):
                pass
            else:
                raise NotImplementedError("nal_unit_type %d" % typ)
        if first:
            return None
        assert (self.mbt > 0).all(), "macroblocks missing from the picture"
        self._deblock()
        self.ref = (self.fy.copy(), self.fu.copy(), self.fv.copy())
        return self.ref

    def _begin_picture(self):
        s = self.sps
        mbw, mbh = s["mbw"], s["mbh"]
        self.mbw, self.mbh = mbw, mbh
        self.y = np.zeros((mbh * 16, mbw * 16), np.int32)
        self.cu = np.zeros((mbh * 8, mbw * 8), np.int32)
        self.cv = np.zeros((mbh * 8, mbw * 8), np.int32)
        n = mbw * mbh
        self.mbt = np.zeros(n, np.int32)        # macroblock kind
        self.slice_of = np.full(n, -1, np.int32)
        self.mv = np.zeros((n, 2), np.int32)    # quarter samples
        self.qp = np.zeros(n, np.int32)
        self.tc_y = np.zeros((mbh * 4, mbw * 4), np.int32)   # TotalCoeff per 4x4 luma block (for nC)
        self.tc_u = np.zeros((mbh * 2, mbw * 2), np.int32)
        self.tc_v = np.zeros((mbh * 2, mbw * 2), np.int32)
        self.nzc_y = np.zeros((mbh * 4, mbw * 4), bool)      # block holds non-zero transform coefficients (for bS)
        self.i4mode = np.full((mbh * 4, mbw * 4), -1, np.int32)
        self.dbk = np.zeros(n, np.int32)        # disable_deblocking_filter_idc of the macroblock's slice
        self.dbk_ab = np.zeros((n, 2), np.int32)
        self.n_slices = 0

    def _slice(self, ref_idc, typ, rb):
        br = Bits(rb)
        s, p = self.sps, self.pps
        first_mb = br.ue()
        st = br.ue() % 5
        if st not in (0, 2):
            raise NotImplementedError("slice_type %d" % st)
        br.ue()
        br.u(s["log2_max_frame_num"])
        if typ == 5:
            br.ue()
        if s["poc_type"] == 0:
            br.u(s["log2_max_poc_lsb"])
            if p["poc_present"]:
                br.se()
        if st == 0:
            if br.u(1):
                br.ue()
            if br.u(1):
                raise NotImplementedError("ref_pic_list_modification")
        if ref_idc:
            if typ == 5:
                br.u(2)
            elif br.u(1):
                raise NotImplementedError("adaptive_ref_pic_marking")
        qp = p["init_qp"] + br.se()
        idc, a_off, b_off = 0, 0, 0
        if p["dbk_present"]:
            idc = br.ue()
            if idc != 1:
                a_off, b_off = 2 * br.se(), 2 * br.se()
        sid = self.n_slices
        self.n_slices += 1
        mb = first_mb
        n = self.mbw * self.mbh
        if st == 2:
            while True:
                self._begin_mb(mb, sid, qp, idc, a_off, b_off)
                qp = self._mb_intra(br, mb, qp)
                mb += 1
                if not br.more_rbsp_data():
                    break
        else:
            while True:
                run = br.ue()
                for _ in range(run):
                    self._begin_mb(mb, sid, qp, idc, a_off, b_off)
                    self._mb_pskip(mb)
                    mb += 1
                if not br.more_rbsp_data():
                    break
                self._begin_mb(mb, sid, qp, idc, a_off, b_off)
                qp = self._mb_p(br, mb, qp)
                mb += 1
                if not br.more_rbsp_data():
                    break
        assert mb <= n

    def _begin_mb(self, mb, sid, qp, idc, a_off, b_off):
        assert self.slice_of[mb] < 0, "macroblock %d coded twice" % mb
        self.slice_of[mb] = sid
        self.qp[mb] = qp
        self.dbk[mb] = idc
        self.dbk_ab[mb] = (a_off, b_off)

    # -- neighbour availability (6.4): another macroblock is available iff it precedes this one in its slice
    def _mb_avail(self, mb, other_x, other_y):
        if other_x < 0 or other_y < 0 or other_x >= self.mbw or other_y >= self.mbh:
            return False
        o = other_y * self.mbw + other_x
        return o < mb and self.slice_of[o] == self.slice_of[mb]

    def _nC(self, mb, plane_tc, bx, by, per_mb):
        """9.2.1: from the TotalCoeff of the blocks to the left (A) and above (B); bx, by in 4x4 blocks of the plane"""
        mx, my = mb % self.mbw, mb // self.mbw
        availA = bx % per_mb > 0 or self._mb_avail(mb, mx - 1, my)
        availB = by % per_mb > 0 or self._mb_avail(mb, mx, my - 1)
        nA = int(plane_tc[by, bx - 1]) if availA else 0
        nB = int(plane_tc[by - 1, bx]) if availB else 0
        if availA and availB:
            return (nA + nB + 1) >> 1
        return nA if availA else nB if availB else 0

    # -- residual of one macroblock (7.3.5.3): fills the planes' residual into res arrays
    def _residual(self, br, mb, qp, cbp, i16, i16_ac):
        """returns (luma residual blocks [16] of 4x4 lists or None, chroma residual [2][4])"""
        mx, my = mb % self.mbw, mb // self.mbw
        qpc = _QPC[max(0, min(51, qp + self.pps["chroma_qp_offset"]))]
        luma = [None] * 16
        dcs = None
        if i16:
            c, _ = residual_block_cavlc(br, self._nC(mb, self.tc_y, mx * 4, my * 4, 4), 16)
            # 8.5.10: inverse 4x4 Hadamard of the DC levels (zig-zag scanned matrix), then scaling
            m = [[0] * 4 for _ in range(4)]
            for k in range(16):
                m[_ZIGZAG[k] >> 2][_ZIGZAG[k] & 3] = c[k]
            A = [[1, 1, 1, 1], [1, 1, -1, -1], [1, -1, -1, 1], [1, -1, 1, -1]]
            t = [[sum(A[i][k] * m[k][j] for k in range(4)) for j in range(4)] for i in range(4)]
            f = [[sum(t[i][k] * A[k][j] for k in range(4)) for j in range(4)] for i in range(4)]
            ls = 16 * _V[qp % 6][0]
            if qp >= 36:
                dcs = [[(f[i][j] * ls) << (qp // 6 - 6) for j in range(4)] for i in range(4)]
            else:
                dcs = [[(f[i][j] * ls + (1 << (5 - qp // 6))) >> (6 - qp // 6) for j in range(4)] for i in range(4)]
        for blk in range(16):
            bx = (blk & 1) | ((blk >> 1) & 2)
            by = ((blk >> 1) & 1) | ((blk >> 2) & 2)
            gx, gy = mx * 4 + bx, my * 4 + by
            coded = (cbp >> (blk >> 2)) & 1 if not i16 else i16_ac
            total = 0
            coef = [0] * 16
            if coded:
                nC = self._nC(mb, self.tc_y, gx, gy, 4)
                if i16:
                    c15, total = residual_block_cavlc(br, nC, 15)
                    coef = [0] + c15
                else:
                    coef, total = residual_block_cavlc(br, nC, 16)
            self.tc_y[gy, gx] = total
            self.nzc_y[gy, gx] = total > 0 or (i16 and dcs is not None and dcs[by][bx] != 0)
            if i16:
                luma[blk] = _idct4(_scale_block(coef, qp, dc=dcs[by][bx]))
            elif total:
                luma[blk] = _idct4(_scale_block(coef, qp))
        chroma = [[None] * 4, [None] * 4]
        cdc = [[0] * 4, [0] * 4]
        cbpc = cbp >> 4
        if cbpc:
            for pl in range(2):
                c, _ = residual_block_cavlc(br, -1, 4)
                # 8.5.11: f = [[1, 1], [1, -1]] c [[1, 1], [1, -1]]; dcC = ((f * LevelScale(QPc % 6, 0, 0)) << (QPc / 6)) >> 5
                f = [c[0] + c[1] + c[2] + c[3], c[0] - c[1] + c[2] - c[3], c[0] + c[1] - c[2] - c[3], c[0] - c[1] - c[2] + c[3]]
                cdc[pl] = [((v * 16 * _V[qpc % 6][0]) << (qpc // 6)) >> 5 for v in f]
        for pl, tcp in ((0, self.tc_u), (1, self.tc_v)):
            for k in range(4):
                gx, gy = mx * 2 + (k & 1), my * 2 + (k >> 1)
                total = 0
                coef = [0] * 16
                if cbpc == 2:
                    c15, total = residual_block_cavlc(br, self._nC(mb, tcp, gx, gy, 2), 15)
                    coef = [0] + c15
                tcp[gy, gx] = total
                if total or cdc[pl][k]:
                    chroma[pl][k] = _idct4(_scale_block(coef, qpc, dc=cdc[pl][k]))
        return luma, chroma

    def _add(self, plane, x0, y0, pred, res):
        for i in range(4):
            for j in range(4):
                plane[y0 + i, x0 + j] = _clip1(int(pred[i][j]) + (res[i][j] if res is not None else 0))

    # -- intra
    def _chroma_dc_pred(self, mb, plane):
        mx, my = mb % self.mbw, mb // self.mbw
        L, T = self._mb_avail(mb, mx - 1, my), self._mb_avail(mb, mx, my - 1)
        x0, y0 = mx * 8, my * 8
        pred = np.zeros((8, 8), np.int32)
        for k in range(4):
            xo, yo = 4 * (k & 1), 4 * (k >> 1)
            st = int(plane[y0 - 1, x0 + xo:x0 + xo + 4].sum()) if T else 0
            sl = int(plane[y0 + yo:y0 + yo + 4, x0 - 1].sum()) if L else 0
            if (xo, yo) in ((0, 0), (4, 4)):
                v = (st + sl + 4) >> 3 if (T and L) else (sl + 2) >> 2 if L else (st + 2) >> 2 if T else 128
            elif (xo, yo) == (4, 0):
                v = (st + 2) >> 2 if T else (sl + 2) >> 2 if L else 128
            else:
                v = (sl + 2) >> 2 if L else (st + 2) >> 2 if T else 128
            pred[yo:yo + 4, xo:xo + 4] = v
        return pred

    def _i16_pred(self, mb, mode):
        mx, my = mb % self.mbw, mb // self.mbw
        x0, y0 = mx * 16, my * 16
        L, T = self._mb_avail(mb, mx - 1, my), self._mb_avail(mb, mx, my - 1)
        Y = self.y
        if mode == 0:
            assert T
            return np.tile(Y[y0 - 1, x0:x0 + 16], (16, 1))
        if mode == 1:
            assert L
            return np.tile(Y[y0:y0 + 16, x0 - 1].reshape(16, 1), (1, 16))
        if mode == 2:
            st = int(Y[y0 - 1, x0:x0 + 16].sum()) if T else 0
            sl = int(Y[y0:y0 + 16, x0 - 1].sum()) if L else 0
            v = (st + sl + 16) >> 5 if (T and L) else (sl + 8) >> 4 if L else (st + 8) >> 4 if T else 128
            return np.full((16, 16), v, np.int32)
        assert L and T and self._mb_avail(mb, mx - 1, my - 1)
        p = lambda x, y: int(Y[y0 + y, x0 + x])
        H = sum((k + 1) * (p(8 + k, -1) - p(6 - k, -1)) for k in range(8))
        V = sum((k + 1) * (p(-1, 8 + k) - p(-1, 6 - k)) for k in range(8))
        a = 16 * (p(-1, 15) + p(15, -1))
        b, c = (5 * H + 32) >> 6, (5 * V + 32) >> 6
        return np.array([[_clip1((a + b * (x - 7) + c * (y - 7) + 16) >> 5) for x in range(16)] for y in range(16)], np.int32)

    def _sample_avail(self, mb, blk, x, y):
        """luma sample (x, y) of the picture usable for the Intra4x4 prediction of block blk of macroblock mb (6.4.10 / 8.3.1.2)"""
        if x < 0 or y < 0 or x >= self.mbw * 16 or y >= self.mbh * 16:
            return False
        omx, omy = x >> 4, y >> 4
        o = omy * self.mbw + omx
        if o == mb:
            bx, by = (x >> 2) & 3, (y >> 2) & 3
            oblk = (bx & 1) | ((by & 1) << 1) | ((bx >> 1) << 2) | ((by >> 1) << 3)
            return oblk < blk
        return o < mb and self.slice_of[o] == self.slice_of[mb]

    def _i4_pred(self, mb, blk, mode):
        mx, my = mb % self.mbw, mb // self.mbw
        bx = (blk & 1) | ((blk >> 1) & 2)
        by = ((blk >> 1) & 1) | ((blk >> 2) & 2)
        x0, y0 = mx * 16 + bx * 4, my * 16 + by * 4
        Y = self.y
        aT = self._sample_avail(mb, blk, x0, y0 - 1)
        aL = self._sample_avail(mb, blk, x0 - 1, y0)
        aTL = self._sample_avail(mb, blk, x0 - 1, y0 - 1)
        aTR = self._sample_avail(mb, blk, x0 + 4, y0 - 1)
        p = {}
        if aT:
            for x in range(4):
                p[(x, -1)] = int(Y[y0 - 1, x0 + x])
            for x in range(4, 8):  # not available: substituted by p[3, -1]
                p[(x, -1)] = int(Y[y0 - 1, x0 + x]) if aTR else p[(3, -1)]
        if aL:
            for y in range(4):
                p[(-1, y)] = int(Y[y0 + y, x0 - 1])
        if aTL:
            p[(-1, -1)] = int(Y[y0 - 1, x0 - 1])
        pr = [[0] * 4 for _ in range(4)]
        if mode == 0:
            assert aT
            for y in range(4):
                for x in range(4):
                    pr[y][x] = p[(x, -1)]
        elif mode == 1:
            assert aL
            for y in range(4):
                for x in range(4):
                    pr[y][x] = p[(-1, y)]
        elif mode == 2:
            st = sum(p[(x, -1)] for x in range(4)) if aT else 0
            sl = sum(p[(-1, y)] for y in range(4)) if aL else 0
            v = (st + sl + 4) >> 3 if (aT and aL) else (sl + 2) >> 2 if aL else (st + 2) >> 2 if aT else 128
            pr = [[v] * 4 for _ in range(4)]
        elif mode == 3:  # diagonal down left
            assert aT
            for y in range(4):
                for x in range(4):
                    if x == 3 and y == 3:
                        pr[y][x] = (p[(6, -1)] + 3 * p[(7, -1)] + 2) >> 2
                    else:
                        pr[y][x] = (p[(x + y, -1)] + 2 * p[(x + y + 1, -1)] + p[(x + y + 2, -1)] + 2) >> 2
        elif mode == 4:  # diagonal down right
            assert aT and aL and aTL
            for y in range(4):
                for x in range(4):
                    if x > y:
                        pr[y][x] = (p[(x - y - 2, -1)] + 2 * p[(x - y - 1, -1)] + p[(x - y, -1)] + 2) >> 2
                    elif x < y:
                        pr[y][x] = (p[(-1, y - x - 2)] + 2 * p[(-1, y - x - 1)] + p[(-1, y - x)] + 2) >> 2
                    else:
                        pr[y][x] = (p[(0, -1)] + 2 * p[(-1, -1)] + p[(-1, 0)] + 2) >> 2
        elif mode == 5:  # vertical right
            assert aT and aL and aTL
            for y in range(4):
                for x in range(4):
                    z = 2 * x - y
                    if z >= 0 and z % 2 == 0:
                        pr[y][x] = (p[(x - (y >> 1) - 1, -1)] + p[(x - (y >> 1), -1)] + 1) >> 1
                    elif z >= 0:
                        pr[y][x] = (p[(x - (y >> 1) - 2, -1)] + 2 * p[(x - (y >> 1) - 1, -1)] + p[(x - (y >> 1), -1)] + 2) >> 2
                    elif z == -1:
                        pr[y][x] = (p[(-1, 0)] + 2 * p[(-1, -1)] + p[(0, -1)] + 2) >> 2
                    else:
                        pr[y][x] = (p[(-1, y - 1)] + 2 * p[(-1, y - 2)] + p[(-1, y - 3)] + 2) >> 2
        elif mode == 6:  # horizontal down
            assert aT and aL and aTL
            for y in range(4):
                for x in range(4):
                    z = 2 * y - x
                    if z >= 0 and z % 2 == 0:
                        pr[y][x] = (p[(-1, y - (x >> 1) - 1)] + p[(-1, y - (x >> 1))] + 1) >> 1
                    elif z >= 0:
                        pr[y][x] = (p[(-1, y - (x >> 1) - 2)] + 2 * p[(-1, y - (x >> 1) - 1)] + p[(-1, y - (x >> 1))] + 2) >> 2
                    elif z == -1:
                        pr[y][x] = (p[(-1, 0)] + 2 * p[(-1, -1)] + p[(0, -1)] + 2) >> 2
                    else:
                        pr[y][x] = (p[(x - 1, -1)] + 2 * p[(x - 2, -1)] + p[(x - 3, -1)] + 2) >> 2
        elif mode == 7:  # vertical left
            assert aT
            for y in range(4):
                for x in range(4):
                    if y % 2 == 0:
                        pr[y][x] = (p[(x + (y >> 1), -1)] + p[(x + (y >> 1) + 1, -1)] + 1) >> 1
                    else:
                        pr[y][x] = (p[(x + (y >> 1), -1)] + 2 * p[(x + (y >> 1) + 1, -1)] + p[(x + (y >> 1) + 2, -1)] + 2) >> 2
        elif mode == 8:  # horizontal up
            assert aL
            for y in range(4):
                for x in range(4):
                    z = x + 2 * y
                    if z > 5:
                        pr[y][x] = p[(-1, 3)]
                    elif z == 5:
                        pr[y][x] = (p[(-1, 2)] + 3 * p[(-1, 3)] + 2) >> 2
                    elif z % 2 == 0:
                        pr[y][x] = (p[(-1, y + (x >> 1))] + p[(-1, y + (x >> 1) + 1)] + 1) >> 1
                    else:
                        pr[y][x] = (p[(-1, y + (x >> 1))] + 2 * p[(-1, y + (x >> 1) + 1)] + p[(-1, y + (x >> 1) + 2)] + 2) >> 2
        else:
            raise ValueError("Intra4x4PredMode %d" % mode)
        return pr

    def _mb_intra(self, br, mb, qp, mb_type=None):
        mx, my = mb % self.mbw, mb // self.mbw
        t = br.ue() if mb_type is None else mb_type
        if t == 25:
            raise NotImplementedError("I_PCM")
        if t == 0:
            # I_NxN: prev_intra4x4_pred_mode_flag / rem_intra4x4_pred_mode of the 16 blocks (8.3.1.1)
            self.mbt[mb] = I4
            modes = []
            for blk in range(16):
                bx = (blk & 1) | ((blk >> 1) & 2)
                by = ((blk >> 1) & 1) | ((blk >> 2) & 2)
                gx, gy = mx * 4 + bx, my * 4 + by
                availA = bx > 0 or self._mb_avail(mb, mx - 1, my)
                availB = by > 0 or self._mb_avail(mb, mx, my - 1)
                if not availA or not availB:
                    pred = 2
                else:
                    mA = int(self.i4mode[gy, gx - 1])
                    mB = int(self.i4mode[gy - 1, gx])
                    pred = min(2 if mA < 0 else mA, 2 if mB < 0 else mB)  # a neighbour that is not Intra4x4: DC
                if br.u(1):
                    m = pred
                else:
                    rem = br.u(3)
                    m = rem if rem < pred else rem + 1
                self.i4mode[gy, gx] = m
                modes.append(m)
            cmode = br.ue()
            cbp = _CBP_INTRA[br.ue()]
            if cbp:
                qp = (qp + br.se() + 52) % 52
            self.qp[mb] = qp
            # luma: residual parsed first (syntax order), prediction + reconstruction block by block
            luma, chroma = self._residual(br, mb, qp, cbp, False, 0)
            for blk in range(16):
                bx = (blk & 1) | ((blk >> 1) & 2)
                by = ((blk >> 1) & 1) | ((blk >> 2) & 2)
                self._add(self.y, mx * 16 + bx * 4, my * 16 + by * 4, self._i4_pred(mb, blk, modes[blk]), luma[blk])
        else:
            self.mbt[mb] = I16
            t -= 1
            mode, cbpc, ac = t % 4, (t // 4) % 3, 1 if t >= 12 else 0
            cmode = br.ue()
            qp = (qp + br.se() + 52) % 52
            self.qp[mb] = qp
            pred = self._i16_pred(mb, mode)
            luma, chroma = self._residual(br, mb, qp, (15 if ac else 0) | (cbpc << 4), True, ac)
            for blk in range(16):
                bx = (blk & 1) | ((blk >> 1) & 2)
                by = ((blk >> 1) & 1) | ((blk >> 2) & 2)
                self._add(self.y, mx * 16 + bx * 4, my * 16 + by * 4, pred[by * 4:by * 4 + 4, bx * 4:bx * 4 + 4], luma[blk])
        if cmode != 0:
            raise NotImplementedError("intra_chroma_pred_mode %d" % cmode)
        for pl, plane in ((0, self.cu), (1, self.cv)):
            pred = self._chroma_dc_pred(mb, plane)
            for k in range(4):
                xo, yo = 4 * (k & 1), 4 * (k >> 1)
                self._add(plane, mx * 8 + xo, my * 8 + yo, pred[yo:yo + 4, xo:xo + 4], chroma[pl][k])
        return qp

    # -- inter
    def _mvp(self, mb):
        """8.4.1.3: median prediction for a 16x16 partition, refIdx 0; returns (mvp, mvA, mvB, availA, availB)"""
        mx, my = mb % self.mbw, mb // self.mbw

        def nb(x, y):
            if not self._mb_avail(mb, x, y):
                return None
            o = y * self.mbw + x
            if self.mbt[o] in (I16, I4):
                return (-1, (0, 0))  # available, intra: refIdx -1
            return (0, (int(self.mv[o][0]), int(self.mv[o][1])))
        A, B, Cn = nb(mx - 1, my), nb(mx, my - 1), nb(mx + 1, my - 1)
        if Cn is None:
            Cn = nb(mx - 1, my - 1)
        if B is None and Cn is None and A is not None:
            mvp = A[1] if A[0] == 0 else (0, 0)
            return mvp, A, B
        cands = [c if c is not None else (-1, (0, 0)) for c in (A, B, Cn)]
        same = [c for c in cands if c[0] == 0]
        if len(same) == 1:
            return same[0][1], A, B
        med = lambda a, b, c: sorted((a, b, c))[1]
        return (med(cands[0][1][0], cands[1][1][0], cands[2][1][0]), med(cands[0][1][1], cands[1][1][1], cands[2][1][1])), A, B

    def _interp_luma(self, ref, x0, y0, mvx, mvy, bw, bh):
        """8.4.2.2.1: bw x bh prediction at full-sample position (x0, y0) + mv (quarter samples), reference clamped at the edges"""
        H, W = ref.shape
        xi, yi, xf, yf = x0 + (mvx >> 2), y0 + (mvy >> 2), mvx & 3, mvy & 3
        ys = np.clip(np.arange(yi - 2, yi + bh + 4), 0, H - 1)
        xs = np.clip(np.arange(xi - 2, xi + bw + 4), 0, W - 1)
        R = ref[np.ix_(ys, xs)].astype(np.int64)  # R[2 + y, 2 + x] = sample (xi + x, yi + y); rows / columns -2 .. size + 3
        tap = lambda a, b, c, d, e, f: a - 5 * b + 20 * c + 20 * d - 5 * e + f
        G = R[2:2 + bh + 1, 2:2 + bw + 1]
        # horizontal half samples b1 for rows -2 .. bh + 2 (needed for j), columns 0 .. bw
        b1 = tap(R[:, 0:bw + 1], R[:, 1:bw + 2], R[:, 2:bw + 3], R[:, 3:bw + 4], R[:, 4:bw + 5], R[:, 5:bw + 6])
        b = np.clip((b1[2:2 + bh + 1] + 16) >> 5, 0, 255)
        h1 = tap(R[0:bh + 1, 2:2 + bw + 1], R[1:bh + 2, 2:2 + bw + 1], R[2:bh + 3, 2:2 + bw + 1], R[3:bh + 4, 2:2 + bw + 1], R[4:bh + 5, 2:2 + bw + 1],
                 R[5:bh + 6, 2:2 + bw + 1])
        h = np.clip((h1 + 16) >> 5, 0, 255)
        j1 = tap(b1[0:bh + 1], b1[1:bh + 2], b1[2:bh + 3], b1[3:bh + 4], b1[4:bh + 5], b1[5:bh + 6])
        j = np.clip((j1 + 512) >> 10, 0, 255)
        c = lambda a: a[:bh, :bw]
        Gx1, Gy1 = G[:bh, 1:bw + 1], G[1:bh + 1, :bw]
        s, m = b[1:bh + 1, :bw], h[:bh, 1:bw + 1]
        avg = lambda p, q: (p + q + 1) >> 1
        tbl = {(0, 0): c(G), (0, 1): avg(c(G), c(h)), (0, 2): c(h), (0, 3): avg(Gy1, c(h)),
               (1, 0): avg(c(G), c(b)), (1, 1): avg(c(b), c(h)), (1, 2): avg(c(h), c(j)), (1, 3): avg(c(h), s),
               (2, 0): c(b), (2, 1): avg(c(b), c(j)), (2, 2): c(j), (2, 3): avg(c(j), s),
               (3, 0): avg(c(b), Gx1), (3, 1): avg(c(b), m), (3, 2): avg(c(j), m), (3, 3): avg(m, s)}
        return tbl[(xf, yf)].astype(np.int32)

    def _interp_chroma(self, ref, x0, y0, mvx, mvy, bw, bh):
        """8.4.2.2.2: mv in eighth chroma samples (= the luma vector's quarter-sample units for 4:2:0)"""
        H, W = ref.shape
        xi, yi, xf, yf = x0 + (mvx >> 3), y0 + (mvy >> 3), mvx & 7, mvy & 7
        ys = np.clip(np.arange(yi, yi + bh + 1), 0, H - 1)
        xs = np.clip(np.arange(xi, xi + bw + 1), 0, W - 1)
        R = ref[np.ix_(ys, xs)].astype(np.int64)
        A, B, Cc, D = R[:bh, :bw], R[:bh, 1:], R[1:, :bw], R[1:, 1:]
        return (((8 - xf) * (8 - yf) * A + xf * (8 - yf) * B + (8 - xf) * yf * Cc + xf * yf * D + 32) >> 6).astype(np.int32)

    def _inter_pred(self, mb):
        if self.ref is None:
            raise ValueError("P picture without a reference picture")
        mx, my = mb % self.mbw, mb // self.mbw
        mvx, mvy = int(self.mv[mb][0]), int(self.mv[mb][1])
        return (self._interp_luma(self.ref[0], mx * 16, my * 16, mvx, mvy, 16, 16),
                self._interp_chroma(self.ref[1], mx * 8, my * 8, mvx, mvy, 8, 8), self._interp_chroma(self.ref[2], mx * 8, my * 8, mvx, mvy, 8, 8))

    def _mb_pskip(self, mb):
        self.mbt[mb] = PSKIP
        mvp, A, B = self._mvp(mb)
        # 8.4.1.1: zero when A or B is not available or has a zero vector with refIdx 0
        zero = A is None or B is None or (A[0] == 0 and A[1] == (0, 0)) or (B[0] == 0 and B[1] == (0, 0))
        self.mv[mb] = (0, 0) if zero else mvp
        py, pu, pv = self._inter_pred(mb)
        mx, my = mb % self.mbw, mb // self.mbw
        self.y[my * 16:my * 16 + 16, mx * 16:mx * 16 + 16] = py
        self.cu[my * 8:my * 8 + 8, mx * 8:mx * 8 + 8] = pu
        self.cv[my * 8:my * 8 + 8, mx * 8:mx * 8 + 8] = pv

    def _mb_p(self, br, mb, qp):
        t = br.ue()
        if t >= 5:
            return self._mb_intra(br, mb, qp, mb_type=t - 5)
        if t != 0:
            raise NotImplementedError("P mb_type %d" % t)
        self.mbt[mb] = PL0
        mvp, _, _ = self._mvp(mb)
        self.mv[mb] = (mvp[0] + br.se(), mvp[1] + br.se())
        cbp = _CBP_INTER[br.ue()]
        if cbp:
            qp = (qp + br.se() + 52) % 52
        self.qp[mb] = qp
        luma, chroma = self._residual(br, mb, qp, cbp, False, 0)
        py, pu, pv = self._inter_pred(mb)
        mx, my = mb % self.mbw, mb // self.mbw
        for blk in range(16):
            bx = (blk & 1) | ((blk >> 1) & 2)
            by = ((blk >> 1) & 1) | ((blk >> 2) & 2)
            self._add(self.y, mx * 16 + bx * 4, my * 16 + by * 4, py[by * 4:by * 4 + 4, bx * 4:bx * 4 + 4], luma[blk])
        for pl, plane, pp in ((0, self.cu, pu), (1, self.cv, pv)):
            for k in range(4):
                xo, yo = 4 * (k & 1), 4 * (k >> 1)
                self._add(plane, mx * 8 + xo, my * 8 + yo, pp[yo:yo + 4, xo:xo + 4], chroma[pl][k])
        return qp

    # -- 8.7 deblocking filter
    def _bs(self, mbp, mbq, bpx, bpy, bqx, bqy, mb_edge):
        """boundary strength of the edge segment between 4x4 luma blocks p (bpx, bpy) and q (in 4x4 units of the picture)"""
        ip, iq = self.mbt[mbp] in (I16, I4), self.mbt[mbq] in (I16, I4)
        if ip or iq:
            return 4 if mb_edge else 3
        if self.nzc_y[bpy, bpx] or self.nzc_y[bqy, bqx]:
            return 2
        if mbp != mbq:
            dx, dy = abs(int(self.mv[mbp][0]) - int(self.mv[mbq][0])), abs(int(self.mv[mbp][1]) - int(self.mv[mbq][1]))
            if dx >= 4 or dy >= 4:
                return 1
        return 0

    def _filter_line(self, s, bs, qpav, a_off, b_off, chroma):
        """8.7.2: one line of samples across an edge; s = [p3, p2, p1, p0, q0, q1, q2, q3] (chroma: p1 .. q1 used); returns the new list"""
        ia, ib = max(0, min(51, qpav + a_off)), max(0, min(51, qpav + b_off))
        alpha, beta = _ALPHA[ia], _BETA[ib]
        p3, p2, p1, p0, q0, q1, q2, q3 = s
        if not (bs and abs(p0 - q0) < alpha and abs(p1 - p0) < beta and abs(q1 - q0) < beta):
            return s
        ap, aq = abs(p2 - p0), abs(q2 - q0)
        if bs < 4:
            tc0 = _TC0[ia][bs - 1]
            tc = tc0 + 1 if chroma else tc0 + (1 if ap < beta else 0) + (1 if aq < beta else 0)
            d = max(-tc, min(tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3))
            np0, nq0 = _clip1(p0 + d), _clip1(q0 - d)
            np1, nq1 = p1, q1
            if not chroma and ap < beta:
                np1 = p1 + max(-tc0, min(tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1))
            if not chroma and aq < beta:
                nq1 = q1 + max(-tc0, min(tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1))
            return [p3, p2, np1, np0, nq0, nq1, q2, q3]
        np2, np1, nq1, nq2 = p2, p1, q1, q2
        small = abs(p0 - q0) < ((alpha >> 2) + 2)
        if not chroma and ap < beta and small:
            np0 = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3
            np1 = (p2 + p1 + p0 + q0 + 2) >> 2
            np2 = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3
        else:
            np0 = (2 * p1 + p0 + q1 + 2) >> 2
        if not chroma and aq < beta and small:
            nq0 = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3
            nq1 = (p0 + q0 + q1 + q2 + 2) >> 2
            nq2 = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3
        else:
            nq0 = (2 * q1 + q0 + p1 + 2) >> 2
        return [p3, np2, np1, np0, nq0, nq1, nq2, q3]

    def _deblock(self):
        self.fy, self.fu, self.fv = self.y.copy(), self.cu.copy(), self.cv.copy()
        off = self.pps["chroma_qp_offset"]
        qpc = lambda q: _QPC[max(0, min(51, int(q) + off))]
        for mb in range(self.mbw * self.mbh):
            idc = int(self.dbk[mb])
            if idc == 1:
                continue
            mx, my = mb % self.mbw, mb // self.mbw
            a_off, b_off = int(self.dbk_ab[mb][0]), int(self.dbk_ab[mb][1])
            left_ok = mx > 0 and (idc == 0 or self.slice_of[mb - 1] == self.slice_of[mb])
            top_ok = my > 0 and (idc == 0 or self.slice_of[mb - self.mbw] == self.slice_of[mb])
            for vertical in (True, False):
                for e in range(4):
                    if e == 0 and not (left_ok if vertical else top_ok):
                        continue
                    mbp = mb if e else (mb - 1 if vertical else mb - self.mbw)
                    for k in range(16):  # line across the edge (luma)
                        seg = k >> 2
                        if vertical:
                            bqx, bqy = mx * 4 + e, my * 4 + seg
                            bpx, bpy = bqx - 1, bqy
                        else:
                            bqx, bqy = mx * 4 + seg, my * 4 + e
                            bpx, bpy = bqx, bqy - 1
                        bs = self._bs(mbp, mb, bpx, bpy, bqx, bqy, e == 0)
                        if not bs:
                            continue
                        qpav = (int(self.qp[mbp]) + int(self.qp[mb]) + 1) >> 1
                        if vertical:
                            x, y = mx * 16 + 4 * e, my * 16 + k
                            s = [int(v) for v in self.fy[y, x - 4:x + 4]]
                            self.fy[y, x - 4:x + 4] = self._filter_line(s, bs, qpav, a_off, b_off, False)
                        else:
                            x, y = mx * 16 + k, my * 16 + 4 * e
                            s = [int(v) for v in self.fy[y - 4:y + 4, x]]
                            self.fy[y - 4:y + 4, x] = self._filter_line(s, bs, qpav, a_off, b_off, False)
                        # chroma: edges 0 and 2 of luma are the chroma edges 0 and 4; chroma line k / 2 takes the strength of luma line k
                        if e % 2 == 0 and k % 2 == 0:
                            for plane in (self.fu, self.fv):
                                qc = (qpc(self.qp[mbp]) + qpc(self.qp[mb]) + 1) >> 1
                                if vertical:
                                    x, y = mx * 8 + 2 * e, my * 8 + (k >> 1)
                                    v = [int(t) for t in plane[y, x - 2:x + 2]]
                                    r = self._filter_line([0, 0, v[0], v[1], v[2], v[3], 0, 0], bs, qc, a_off, b_off, True)
                                    plane[y, x - 2:x + 2] = r[2:6]
                                else:
                                    x, y = mx * 8 + (k >> 1), my * 8 + 2 * e
                                    v = [int(t) for t in plane[y - 2:y + 2, x]]
                                    r = self._filter_line([0, 0, v[0], v[1], v[2], v[3], 0, 0], bs, qc, a_off, b_off, True)
                                    plane[y - 2:y + 2, x] = r[2:6]
