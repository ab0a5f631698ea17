/*
 * orc_intra4.h -- Intra4x4 prediction of the CPU oracle (TEST INFRASTRUCTURE ONLY; PARITY UNPINNED, see orc.h), shared by
 * its encoder and its decoder.  H.264 8.3.1: neighbour availability and the nine predictors (8.3.1.2.1 - 8.3.1.2.9),
 * restated from the standard's text; no reference file holds this arithmetic (the reference delegates it to libx264,
 * av_encode.c:970).  tests/test_oracle_intra4.py holds a second, independent restatement in Python.
 */
#ifndef ORC_INTRA4_H
#define ORC_INTRA4_H
#include <stddef.h>
#include <stdint.h>

/* luma4x4BlkIdx of the block at (bx, by) in units of blocks (6.4.3: the inverse 4x4 luma block scan) */
static inline int orc_i4_idx(int bx, int by) { return (by >> 1) * 8 + (bx >> 1) * 4 + (by & 1) * 2 + (bx & 1); }

typedef struct {
    int top, left, corner, topright; /* availability of p[0..3, -1], p[-1, 0..3], p[-1, -1], p[4..7, -1] */
    int t[8];                        /* p[x, -1], x = 0 .. 7 (4 .. 7: p[3, -1] repeated when only they are missing, 8.3.1.2) */
    int l[4];                        /* p[-1, y] */
    int c;                           /* p[-1, -1] */
} orc_i4_nb;

/* Neighbours of block (bx, by) of the macroblock whose top-left luma sample is `mb` in the plane `rec` (stride `stride`):
 * samples of this macroblock come from blocks decoded earlier, others from the neighbouring macroblocks, available iff
 * that macroblock is (6.4.11.4): aL left, aT above, aTL above-left, aTR above-right.  The upper-right block of blocks 3,
 * 11 (decoded later) and of the blocks in the macroblock's right column below its first row (in the macroblock to the
 * right) is never available. */
static inline void orc_i4_neighbours(const uint8_t *mb, int stride, int bx, int by, int aL, int aT, int aTL, int aTR, orc_i4_nb *n) {
    const uint8_t *p = mb + (size_t)(4 * by) * stride + 4 * bx;
    n->top = by > 0 || aT;
    n->left = bx > 0 || aL;
    n->corner = (bx > 0 && by > 0) || (bx == 0 && by > 0 && aL) || (bx > 0 && by == 0 && aT) || (bx == 0 && by == 0 && aTL);
    if (by == 0)
        n->topright = bx < 3 ? aT : aTR;
    else
        n->topright = bx < 3 && orc_i4_idx(bx + 1, by - 1) < orc_i4_idx(bx, by);
    for (int x = 0; x < 8; x++) n->t[x] = 128;
    for (int y = 0; y < 4; y++) n->l[y] = 128;
    n->c = 128;
    if (n->top) {
        for (int x = 0; x < 4; x++) n->t[x] = p[-(ptrdiff_t)stride + x];
        for (int x = 4; x < 8; x++) n->t[x] = n->topright ? p[-(ptrdiff_t)stride + x] : n->t[3];
    }
    if (n->left)
        for (int y = 0; y < 4; y++) n->l[y] = p[(ptrdiff_t)y * stride - 1];
    if (n->corner) n->c = p[-(ptrdiff_t)stride - 1];
}

/* mode usable with these neighbours (8.3.1.2.1 - 9: each mode names the samples it needs) */
static inline int orc_i4_mode_ok(const orc_i4_nb *n, int mode) {
    switch (mode) {
        case 0: return n->top;                          /* vertical */
        case 1: return n->left;                         /* horizontal */
        case 2: return 1;                               /* DC */
        case 3: return n->top;                          /* diagonal down left (upper-right samples substituted when missing) */
        case 4: case 5: case 6: return n->top && n->left && n->corner; /* diagonal down right, vertical right, horizontal down */
        case 7: return n->top;                          /* vertical left */
        case 8: return n->left;                         /* horizontal up */
    }
    return 0;
}

/* pred[4 * y + x] of mode `mode` */
static inline void orc_i4_predict(const orc_i4_nb *n, int mode, uint8_t pred[16]) {
    /* P(x, -1) = T(x), P(-1, y) = L(y), with index -1 of either meaning the corner */
#define T(x) ((x) < 0 ? n->c : n->t[x])
#define L(y) ((y) < 0 ? n->c : n->l[y])
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) {
            int v = 0;
            switch (mode) {
                case 0: v = T(x); break;
                case 1: v = L(y); break;
                case 2:
                    if (n->top && n->left) v = (T(0) + T(1) + T(2) + T(3) + L(0) + L(1) + L(2) + L(3) + 4) >> 3;
                    else if (n->left) v = (L(0) + L(1) + L(2) + L(3) + 2) >> 2;
                    else if (n->top) v = (T(0) + T(1) + T(2) + T(3) + 2) >> 2;
                    else v = 128;
                    break;
                case 3:
                    v = (x == 3 && y == 3) ? (T(6) + 3 * T(7) + 2) >> 2 : (T(x + y) + 2 * T(x + y + 1) + T(x + y + 2) + 2) >> 2;
                    break;
                case 4:
                    if (x > y) v = (T(x - y - 2) + 2 * T(x - y - 1) + T(x - y) + 2) >> 2;
                    else if (x < y) v = (L(y - x - 2) + 2 * L(y - x - 1) + L(y - x) + 2) >> 2;
                    else v = (T(0) + 2 * n->c + L(0) + 2) >> 2;
                    break;
                case 5: {
                    const int z = 2 * x - y;
                    if (z >= 0 && !(z & 1)) v = (T(x - (y >> 1) - 1) + T(x - (y >> 1)) + 1) >> 1;
                    else if (z >= 0) v = (T(x - (y >> 1) - 2) + 2 * T(x - (y >> 1) - 1) + T(x - (y >> 1)) + 2) >> 2;
                    else if (z == -1) v = (L(0) + 2 * n->c + T(0) + 2) >> 2;
                    else v = (L(y - 1) + 2 * L(y - 2) + L(y - 3) + 2) >> 2;
                    break;
                }
                case 6: {
                    const int z = 2 * y - x;
                    if (z >= 0 && !(z & 1)) v = (L(y - (x >> 1) - 1) + L(y - (x >> 1)) + 1) >> 1;
                    else if (z >= 0) v = (L(y - (x >> 1) - 2) + 2 * L(y - (x >> 1) - 1) + L(y - (x >> 1)) + 2) >> 2;
                    else if (z == -1) v = (L(0) + 2 * n->c + T(0) + 2) >> 2;
                    else v = (T(x - 1) + 2 * T(x - 2) + T(x - 3) + 2) >> 2;
                    break;
                }
                case 7:
                    v = !(y & 1) ? (T(x + (y >> 1)) + T(x + (y >> 1) + 1) + 1) >> 1
                                 : (T(x + (y >> 1)) + 2 * T(x + (y >> 1) + 1) + T(x + (y >> 1) + 2) + 2) >> 2;
                    break;
                case 8: {
                    const int z = x + 2 * y;
                    if (z > 5) v = L(3);
                    else if (z == 5) v = (L(2) + 3 * L(3) + 2) >> 2;
                    else if (!(z & 1)) v = (L(y + (x >> 1)) + L(y + (x >> 1) + 1) + 1) >> 1;
                    else v = (L(y + (x >> 1)) + 2 * L(y + (x >> 1) + 1) + L(y + (x >> 1) + 2) + 2) >> 2;
                    break;
                }
            }
            pred[4 * y + x] = (uint8_t)v;
        }
#undef T
#undef L
}

/* predIntra4x4PredMode (8.3.1.1) from the modes of the blocks to the left (A) and above (B): -1 = that block's macroblock is
 * not available (then the prediction is DC whatever the other says); a macroblock that is not Intra4x4 counts as DC */
static inline int orc_i4_pred_mode(int modeA, int modeB) {
    if (modeA < 0 || modeB < 0) return 2;
    return modeA < modeB ? modeA : modeB;
}
#endif
