/*
 * orc_deblock.h -- in-loop deblocking filter (TEST INFRASTRUCTURE ONLY; see orc.h).
 *
 * PARITY UNPINNED (orc.h), and conformance unpinned too: the reference holds no fixture for this path and no other
 * H.264 decoder is available here, so the tables below (Table 8-16 alpha / beta, Table 8-17 tC0) are restated from the
 * standard as published and checked only for their structure (tests/test_oracle_deblock.py) and against a second,
 * independently written restatement of the process in Python.
 *
 * A restatement of H.264 8.7 (deblocking filter process) for the streams this codec produces: frame macroblocks, one
 * reference picture, 4x4 transform, 4:2:0, one QP per picture, FilterOffsetA = FilterOffsetB = 0 and
 * disable_deblocking_filter_idc = 2 (edges that coincide with slice boundaries are not filtered; picture edges never
 * are).  Macroblocks are filtered in raster order: the four vertical luma edges left to right (each 16 samples), the two
 * vertical chroma edges, then the horizontal ones; an edge sees the samples the earlier edges and macroblocks left
 * behind.  Shared by the oracle encoder (the filtered picture is the reference of the next) and the oracle decoder.
 * Behind the reference's x264_encoder_encode (av_encode.c:970) this is libx264's loop filter, on in every preset the
 * reference can select (av_encode.c:365).
 */
#ifndef ORC_DEBLOCK_H
#define ORC_DEBLOCK_H
#include <stdint.h>
#include <stdlib.h>

static const uint8_t orc_dbk_alpha[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13,
                                          15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255};
static const uint8_t orc_dbk_beta[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
                                         6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18};
/* tC0 for bS = 1, 2, 3 */
static const uint8_t orc_dbk_tc0[52][3] = {
    {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0},  {0, 0, 0},  {0, 0, 0},
    {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 1, 1}, {0, 1, 1}, {1, 1, 1},  {1, 1, 1},  {1, 1, 1},
    {1, 1, 1}, {1, 1, 2}, {1, 1, 2}, {1, 1, 2}, {1, 1, 2}, {1, 2, 3}, {1, 2, 3}, {2, 2, 3}, {2, 2, 4}, {2, 3, 4}, {2, 3, 4},  {3, 3, 5},  {3, 4, 6},
    {3, 4, 6}, {4, 5, 7}, {4, 5, 8}, {4, 6, 9}, {5, 7, 10}, {6, 8, 11}, {6, 8, 13}, {7, 10, 14}, {8, 11, 16}, {9, 12, 18}, {10, 13, 20}, {11, 15, 23}, {13, 17, 25}};

typedef struct {
    uint8_t *plane[3]; /* the decoded picture, filtered in place */
    int stride[3];
    int mbw, mbh;
    const uint8_t *intra;  /* [mb] non-zero: intra macroblock */
    const uint8_t *nnz_l;  /* [mb][16] coefficients per luma 4x4 block, index 4 * by + bx */
    const int8_t *mv;      /* [mb][2] quarter-sample vector (P macroblocks, the inferred one for P_Skip) */
    const int32_t *slice;  /* [mb] slice the macroblock belongs to */
    const uint8_t *qp;     /* [mb] QP_Y */
    const uint8_t *enable; /* [mb] non-zero: disable_deblocking_filter_idc of the macroblock's slice is 2 (0: idc 1, not filtered) */
    const uint8_t *qpc_of; /* [52] QP_C by QP_Y (table 8-15, chroma_qp_index_offset 0) */
} orc_dbk;

static inline int orc_dbk_clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }

/* one edge sample position: p3 p2 p1 p0 | q0 q1 q2 q3 along `step` (1: across a vertical edge, stride: across a horizontal one) */
static inline void orc_dbk_filter(uint8_t *q0p, int step, int bs, int qpav, int chroma) {
    const int alpha = orc_dbk_alpha[qpav], beta = orc_dbk_beta[qpav];
    const int p0 = q0p[-step], p1 = q0p[-2 * step], q0 = q0p[0], q1 = q0p[step];
    if (!(abs(p0 - q0) < alpha && abs(p1 - p0) < beta && abs(q1 - q0) < beta)) return;
    if (bs < 4) {
        const int tc0 = orc_dbk_tc0[qpav][bs - 1];
        int tc;
        if (chroma)
            tc = tc0 + 1;
        else {
            const int p2 = q0p[-3 * step], q2 = q0p[2 * step];
            const int ap = abs(p2 - p0) < beta, aq = abs(q2 - q0) < beta;
            tc = tc0 + ap + aq;
            if (ap) q0p[-2 * step] = (uint8_t)(p1 + orc_dbk_clip3(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1));
            if (aq) q0p[step] = (uint8_t)(q1 + orc_dbk_clip3(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1));
        }
        const int delta = orc_dbk_clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
        q0p[-step] = (uint8_t)orc_dbk_clip3(0, 255, p0 + delta);
        q0p[0] = (uint8_t)orc_dbk_clip3(0, 255, q0 - delta);
    } else if (chroma) {
        q0p[-step] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
        q0p[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
    } else {
        const int p2 = q0p[-3 * step], p3 = q0p[-4 * step], q2 = q0p[2 * step], q3 = q0p[3 * step];
        const int strong = abs(p0 - q0) < ((alpha >> 2) + 2);
        if (strong && abs(p2 - p0) < beta) {
            q0p[-step] = (uint8_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
            q0p[-2 * step] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
            q0p[-3 * step] = (uint8_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
        } else
            q0p[-step] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
        if (strong && abs(q2 - q0) < beta) {
            q0p[0] = (uint8_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
            q0p[step] = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
            q0p[2 * step] = (uint8_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
        } else
            q0p[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
    }
}

/* boundary strength of the edge segment between luma 4x4 blocks (mbp, blkp) | (mbq, blkq); mb_edge: the segment lies on a macroblock edge */
static inline int orc_dbk_bs(const orc_dbk *d, int mbp, int blkp, int mbq, int blkq, int mb_edge) {
    if (d->intra[mbp] || d->intra[mbq]) return mb_edge ? 4 : 3;
    if (d->nnz_l[mbp * 16 + blkp] || d->nnz_l[mbq * 16 + blkq]) return 2;
    if (mbp != mbq && (abs(d->mv[2 * mbp] - d->mv[2 * mbq]) >= 4 || abs(d->mv[2 * mbp + 1] - d->mv[2 * mbq + 1]) >= 4)) return 1;
    return 0; /* one reference picture, one vector per macroblock */
}

static inline void orc_deblock_picture(const orc_dbk *d) {
    for (int my = 0; my < d->mbh; my++)
        for (int mx = 0; mx < d->mbw; mx++) {
            const int mb = my * d->mbw + mx;
            if (!d->enable[mb]) continue;
            const int left = mx > 0 && d->slice[mb - 1] == d->slice[mb], top = my > 0 && d->slice[mb - d->mbw] == d->slice[mb];
            for (int dir = 0; dir < 2; dir++) { /* 0: vertical edges (filtered across columns), 1: horizontal edges */
                for (int e = 0; e < 4; e++) {
                    if (e == 0 && !(dir ? top : left)) continue;
                    const int mbp = e ? mb : (dir ? mb - d->mbw : mb - 1);
                    const int qpav_l = (d->qp[mbp] + d->qp[mb] + 1) >> 1;
                    const int qpav_c = (d->qpc_of[d->qp[mbp]] + d->qpc_of[d->qp[mb]] + 1) >> 1;
                    for (int k = 0; k < 16; k++) { /* sample position along the edge */
                        const int seg = k >> 2; /* 4x4 block along the edge */
                        const int blkq = dir ? 4 * e + seg : 4 * seg + e;
                        const int blkp = e ? (dir ? 4 * (e - 1) + seg : 4 * seg + e - 1) : (dir ? 12 + seg : 4 * seg + 3);
                        const int bs = orc_dbk_bs(d, mbp, blkp, mb, blkq, e == 0);
                        if (!bs) continue;
                        uint8_t *y = d->plane[0] + (size_t)(my * 16 + (dir ? 4 * e : k)) * d->stride[0] + mx * 16 + (dir ? k : 4 * e);
                        orc_dbk_filter(y, dir ? d->stride[0] : 1, bs, qpav_l, 0);
                        /* chroma: the edges at luma 0 and 8 = chroma 0 and 4; chroma sample k / 2 takes the strength of luma
                         * position k (8.7.2: the bS of the corresponding luma edge sample) */
                        if (!(e & 1) && !(k & 1))
                            for (int p = 1; p <= 2; p++) {
                                uint8_t *c = d->plane[p] + (size_t)(my * 8 + (dir ? 2 * e : k / 2)) * d->stride[p] + mx * 8 + (dir ? k / 2 : 2 * e);
                                orc_dbk_filter(c, dir ? d->stride[p] : 1, bs, qpav_c, 1);
                            }
                    }
                }
            }
        }
}
#endif
