/*
 * venc_lines.h -- per-lane logic of round 3's kernels that can be checked without a GPU (compiled by hipcc into the
 * kernels and by g++ into the host-check library, like venc_device.h):
 *   venc_dbk_line         one line across an edge, H.264 8.7.2.3 / 8.7.2.4 (k_deblock, and k_deblock_ls where a strong edge is in the wave)
 *   venc_dbk_line_normal  the normal filter (bS 1 .. 3) without a branch (k_deblock_ls's fast path)
 *   venc_nz_locate        entry k of an item's concatenated lists of 4x4 blocks -> (segment, index) (k_me_fix's follow-up blocks)
 */
#ifndef VENC_LINES_H
#define VENC_LINES_H
#include "venc_device.h"

#if defined(__HIPCC__)
#define VENC_ABSDIFF_U8(a, b) ((int)__builtin_amdgcn_sad_u8((uint32_t)(a), (uint32_t)(b), 0u)) /* operands < 256 */
#else
#define VENC_ABSDIFF_U8(a, b) ((a) > (b) ? (a) - (b) : (b) - (a))
#endif
VENC_DEV int venc_imin(int a, int b) { return a < b ? a : b; }
VENC_DEV int venc_imax(int a, int b) { return a > b ? a : b; }

/* (A branch-free restatement with selects, the strong filter behind a wave-uniform test, was measured SLOWER in the strip kernel:
 * 376 -> 403 us per launch, 6.5 M -> 8.3 M vector instructions -- most lines fail the alpha / beta test and the branches skip the rest.) */
/* one line across an edge: s[0..7] = p3 p2 p1 p0 q0 q1 q2 q3 (chroma: p1 p0 q0 q1 in s[2..5]); 8.7.2.3 / 8.7.2.4 */
VENC_DEV void venc_dbk_line(int s[8], int bs, int alpha, int beta, int tc0, bool chroma) {
    const int p3 = s[0], p2 = s[1], p1 = s[2], p0 = s[3], q0 = s[4], q1 = s[5], q2 = s[6], q3 = s[7];
    if (!(bs && VENC_ABSDIFF_U8(p0, q0) < alpha && VENC_ABSDIFF_U8(p1, p0) < beta && VENC_ABSDIFF_U8(q1, q0) < beta)) return;
    const bool ap = !chroma && VENC_ABSDIFF_U8(p2, p0) < beta, aq = !chroma && VENC_ABSDIFF_U8(q2, q0) < beta;
    if (bs < 4) {
        const int tc = chroma ? tc0 + 1 : tc0 + (ap ? 1 : 0) + (aq ? 1 : 0);
        const int delta = venc_imin(venc_imax((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc), tc);
        s[3] = venc_clip_u8(p0 + delta);
        s[4] = venc_clip_u8(q0 - delta);
        const int avg = (p0 + q0 + 1) >> 1;
        if (ap) s[2] = p1 + venc_imin(venc_imax((p2 + avg - (p1 << 1)) >> 1, -tc0), tc0);
        if (aq) s[5] = q1 + venc_imin(venc_imax((q2 + avg - (q1 << 1)) >> 1, -tc0), tc0);
    } else {
        const bool small = !chroma && VENC_ABSDIFF_U8(p0, q0) < ((alpha >> 2) + 2);
        if (small && ap) {
            s[3] = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3;
            s[2] = (p2 + p1 + p0 + q0 + 2) >> 2;
            s[1] = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3;
        } else
            s[3] = (2 * p1 + p0 + q1 + 2) >> 2;
        if (small && aq) {
            s[4] = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3;
            s[5] = (p0 + q0 + q1 + q2 + 2) >> 2;
            s[6] = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3;
        } else
            s[4] = (2 * q1 + q0 + p1 + 2) >> 2;
    }
}

/* The normal filter (bS 1 .. 3, 8.7.2.3) of one line without a branch: p2 .. q2 in, p1 p0 q0 q1 out (unchanged where the line is not
 * filtered).  What a step of k_deblock_ls costs is the LENGTH of the filter wave's instruction stream, not the work of the lanes it
 * skips: venc_dbk_line's nested tests (most lines fail the alpha / beta test) save a throughput-bound kernel a third of its
 * instructions and cost this one a branch each.  Same arithmetic as venc_dbk_line, term by term (tests/test_device_functions_cpu.py). */
VENC_DEV void venc_dbk_line_normal(bool chroma, int p2, int *p1, int *p0, int *q0, int *q1, int q2, int bs, int alpha, int beta, int tc0) {
    const int P1 = *p1, P0 = *p0, Q0 = *q0, Q1 = *q1;
    const bool f = bs != 0 && VENC_ABSDIFF_U8(P0, Q0) < alpha && VENC_ABSDIFF_U8(P1, P0) < beta && VENC_ABSDIFF_U8(Q1, Q0) < beta;
    const bool ap = !chroma && VENC_ABSDIFF_U8(p2, P0) < beta;
    const bool aq = !chroma && VENC_ABSDIFF_U8(q2, Q0) < beta;
    const int tc = chroma ? tc0 + 1 : tc0 + (ap ? 1 : 0) + (aq ? 1 : 0);
    const int delta = venc_imin(venc_imax((((Q0 - P0) << 2) + (P1 - Q1) + 4) >> 3, -tc), tc);
    const int np0 = venc_clip_u8(P0 + delta), nq0 = venc_clip_u8(Q0 - delta);
    if (!chroma) {
        const int avg = (P0 + Q0 + 1) >> 1;
        const int np1 = P1 + venc_imin(venc_imax((p2 + avg - (P1 << 1)) >> 1, -tc0), tc0);
        const int nq1 = Q1 + venc_imin(venc_imax((q2 + avg - (Q1 << 1)) >> 1, -tc0), tc0);
        *p1 = f && ap ? np1 : P1;
        *q1 = f && aq ? nq1 : Q1;
    }
    *p0 = f ? np0 : P0;
    *q0 = f ? nq0 : Q0;
}

/* The follow-up blocks of the k_me_fix launch (venc_me2.hip luma_nz_redo) see an item's listed 4x4 blocks as ONE concatenated list:
 * 256 thread-segments of rpt (tile, wave) segments each.  part[t] = entries before thread-segment t (part[256] = all), cnt32 = the
 * segments' counts as bytes (four per word, rpt / 4 words per thread-segment).  Entry k (< part[256]) lies in the last thread-segment
 * that starts at or before it (never an empty one: the next one starts behind k); inside, the counts are walked.
 * Returns the (tile, wave) segment; *index = the entry's place inside it. */
VENC_DEV int venc_nz_locate(const uint32_t *part, const uint32_t *cnt32, int rpt, uint32_t k, uint32_t *index) {
    int t = 0;
    for (int step = 128; step; step >>= 1)
        if (part[t + step] <= k) t += step;
    uint32_t rem = k - part[t];
    int r = t * rpt;
    const int rpt4 = rpt >> 2;
    for (int q = 0; q < rpt4; q++) { /* rem < the sum of this thread-segment's counts */
        const uint32_t wd = cnt32[t * rpt4 + q];
        const uint32_t c0 = wd & 255u, c1 = (wd >> 8) & 255u, c2 = (wd >> 16) & 255u, c3 = wd >> 24;
        if (rem < c0 + c1 + c2 + c3) {
            if (rem >= c0) {
                rem -= c0, r++;
                if (rem >= c1) {
                    rem -= c1, r++;
                    if (rem >= c2) rem -= c2, r++;
                }
            }
            break;
        }
        rem -= c0 + c1 + c2 + c3, r += 4;
    }
    *index = rem;
    return r;
}
#endif
