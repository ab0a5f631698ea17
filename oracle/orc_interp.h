/*
 * orc_interp.h -- luma sample interpolation at quarter-sample positions (TEST INFRASTRUCTURE ONLY; see orc.h).
 *
 * PARITY UNPINNED (orc.h).  A restatement of H.264 8.4.2.2.1 (luma sample interpolation process): the six-tap filter
 * (1, -5, 20, 20, -5, 1) for the half-sample positions b, h and -- from the UNCLIPPED intermediate sums -- the centre j;
 * quarter-sample positions as rounded averages of the two nearest integer / half samples (Table 8-12 of the standard,
 * Figure 8-4 naming: G H / M N the integer samples of a cell, b between G and H, h between G and M, m between H and N,
 * s between M and N, j the centre).  Reference coordinates are clamped to the picture (the standard's Clip3 on
 * xInt / yInt), so a plane needs no border.
 * Shared by the oracle encoder (sub-sample refinement, spec/CODEC.md section 2a, and the prediction it codes against) and
 * the oracle decoder.  Behind the reference's x264_encoder_encode (av_encode.c:970) this is libx264's subpixel motion
 * estimation (the presets of av_encode.c:365 switch it on), which the reference does not ship.
 */
#ifndef ORC_INTERP_H
#define ORC_INTERP_H
#include <stdint.h>

typedef struct {
    const uint8_t *p;
    int stride, w, h;
} orc_plane;

static inline int orc_px(const orc_plane *r, int x, int y) {
    x = x < 0 ? 0 : x >= r->w ? r->w - 1 : x;
    y = y < 0 ? 0 : y >= r->h ? r->h - 1 : y;
    return r->p[(size_t)y * r->stride + x];
}
static inline int orc_tap6(int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; }
static inline int orc_clip255(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
/* unclipped half-sample sums: b1 between (x, y) and (x + 1, y); h1 between (x, y) and (x, y + 1) */
static inline int orc_b1(const orc_plane *r, int x, int y) {
    return orc_tap6(orc_px(r, x - 2, y), orc_px(r, x - 1, y), orc_px(r, x, y), orc_px(r, x + 1, y), orc_px(r, x + 2, y), orc_px(r, x + 3, y));
}
static inline int orc_h1(const orc_plane *r, int x, int y) {
    return orc_tap6(orc_px(r, x, y - 2), orc_px(r, x, y - 1), orc_px(r, x, y), orc_px(r, x, y + 1), orc_px(r, x, y + 2), orc_px(r, x, y + 3));
}
/* sample of the doubled grid: (hx, hy) in HALF-sample units; even / even = integer sample, odd / even = b, even / odd = h,
 * odd / odd = j */
static inline int orc_half_sample(const orc_plane *r, int hx, int hy) {
    const int x = hx >> 1, y = hy >> 1; /* floor: hx, hy may be negative */
    if (!(hx & 1) && !(hy & 1)) return orc_px(r, x, y);
    if ((hx & 1) && !(hy & 1)) return orc_clip255((orc_b1(r, x, y) + 16) >> 5);
    if (!(hx & 1) && (hy & 1)) return orc_clip255((orc_h1(r, x, y) + 16) >> 5);
    const int j1 = orc_tap6(orc_b1(r, x, y - 2), orc_b1(r, x, y - 1), orc_b1(r, x, y), orc_b1(r, x, y + 1), orc_b1(r, x, y + 2), orc_b1(r, x, y + 3));
    return orc_clip255((j1 + 512) >> 10);
}
/* luma sample at (xq, yq) in QUARTER-sample units (8.4.2.2.1).  Both even: a sample of the doubled grid.  One odd: the
 * rounded average of the two grid samples on either side along that axis (a, c, d, n, f, i, k, q).  Both odd: the rounded
 * average of the two corners of the surrounding grid cell that are half samples of type b / h (e, g, p, r) -- the corners
 * whose half-unit coordinates have an odd sum. */
static inline int orc_luma_qpel(const orc_plane *r, int xq, int yq) {
    const int ox = xq & 1, oy = yq & 1;
    const int hx0 = (xq - ox) >> 1, hy0 = (yq - oy) >> 1; /* lower corner of the cell on the doubled grid (floor) */
    if (!ox && !oy) return orc_half_sample(r, hx0, hy0);
    if (ox && !oy) return (orc_half_sample(r, hx0, hy0) + orc_half_sample(r, hx0 + 1, hy0) + 1) >> 1;
    if (!ox && oy) return (orc_half_sample(r, hx0, hy0) + orc_half_sample(r, hx0, hy0 + 1) + 1) >> 1;
    if ((hx0 + hy0) & 1) /* the lower corner is a b / h sample: it and the opposite corner */
        return (orc_half_sample(r, hx0, hy0) + orc_half_sample(r, hx0 + 1, hy0 + 1) + 1) >> 1;
    return (orc_half_sample(r, hx0 + 1, hy0) + orc_half_sample(r, hx0, hy0 + 1) + 1) >> 1;
}
/* bw x bh prediction block whose top-left sample sits at (xq, yq) quarter units */
static inline void orc_luma_pred_block(const orc_plane *r, int xq, int yq, int bw, int bh, uint8_t *dst, int dstride) {
    for (int y = 0; y < bh; y++)
        for (int x = 0; x < bw; x++) dst[y * dstride + x] = (uint8_t)orc_luma_qpel(r, xq + 4 * x, yq + 4 * y);
}
#endif
